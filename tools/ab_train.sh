#!/bin/bash
# A/B of prebuilt library variants (tools/ab/lib*.so) on the training step: tools/ab_train.sh "V0 V1" [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
cp sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so
for r in $(seq 1 ${2:-1}); do for v in $1; do
  cp tools/ab/lib$v.so sinddm_amd/libsinddm_hip.so
  echo "$v $(python tools/train_bench.py ${SCALE:-4} 5 2>&1 | tail -1 | cut -c1-100)"
done; done | tee gpurun_out/ab_train.log
cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so
