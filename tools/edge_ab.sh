#!/bin/bash
# tools/edge_ab.sh "<variants>": tools/edge_cost.py per prebuilt library variant
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
cp sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so
for v in $1; do cp tools/ab/lib$v.so sinddm_amd/libsinddm_hip.so; echo "== $v"; python tools/edge_cost.py 2>&1 | grep -v amdgpu | head -3; done | tee gpurun_out/edge_ab.log
cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so
