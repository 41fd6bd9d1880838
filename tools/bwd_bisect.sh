#!/bin/bash
# tools/bwd_bisect.sh "<variants>": run tools/bwd_bisect.py once per prebuilt library variant (tools/ab/lib<V>.so)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
cp sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so
for v in $1; do
  cp tools/ab/lib$v.so sinddm_amd/libsinddm_hip.so
  python tools/bwd_bisect.py $v 2>&1 | tail -1
done | tee gpurun_out/bwd_bisect.log
cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so
