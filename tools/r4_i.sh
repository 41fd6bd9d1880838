#!/bin/bash
# full GPU suite of the current library, then A/B of prebuilt variants: tools/r4_i.sh "BASE GELU" [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/r4i_tests.txt
bash tools/ab.sh "$1" ${2:-2}
