#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
cp sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so
for v in $1; do cp tools/ab/lib$v.so sinddm_amd/libsinddm_hip.so; echo "== $v"; python tools/w4_debug.py 2>&1 | grep -v amdgpu.ids | cut -c1-150; done
cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so
