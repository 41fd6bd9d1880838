#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sampler_fast.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -4
cp sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so
for v in H5 H8 H12; do cp tools/ab/lib$v.so sinddm_amd/libsinddm_hip.so; echo "== $v";
 for a in "C2 1 16" "C2 2 16" "C2 3 16" "C4 2 16" "C4 3 16" "C3 0 64" "C3 1 64" "C5 0 4" "C5 1 4" "C5 2 4"; do python tools/scale_chain_profile.py $a 2>&1 | grep -v amdgpu.ids; done; done
cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so
