#!/bin/bash
# A/B of prebuilt variants + optional item timing probe:  tools/r3_ab.sh "<variants>" [probe variant]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out; rm -f gpurun_out/ab.log
timeout 1500 bash tools/ab2.sh "$1" ${3:-1} "C2 C3"
[ -n "$2" ] && python tools/w4_seg.py $2 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r3_w4_seg.txt
