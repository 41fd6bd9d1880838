#!/bin/bash
# per-segment item timing of -DW4_TIMING variants ($1), then A/B of variants ($2)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
for v in $1; do echo "== $v"; timeout 300 python tools/w4_seg.py $v 2>&1 | grep -v "amdgpu.ids"; done | tee gpurun_out/r4h_seg.txt
bash tools/ab.sh "$2" ${3:-2}
