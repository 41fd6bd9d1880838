#!/bin/bash
# per-scale times of a full sample for prebuilt variants: tools/scale_ab.sh "<variants>" "<cfg:batch ...>"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
cp sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so
for v in $1; do for cb in ${2:-"C2:16"}; do
  cp tools/ab/lib$v.so sinddm_amd/libsinddm_hip.so
  echo "== $v $cb"; python tools/scale_times.py ${cb%%:*} ${cb##*:} 2>&1 | grep -v amdgpu.ids
done; done | tee gpurun_out/scale_ab.log
cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so
