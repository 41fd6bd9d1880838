#!/bin/bash
# parity of the current library on the conv tests, then A/B of prebuilt variants: tools/r4_g.sh "BASE SKEW" [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_wino4.py tests/test_gpu_forward.py tests/test_gpu_sampler_fast.py tests/test_build_isa.py -x -q 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/r4g_tests.txt
bash tools/ab.sh "$1" ${2:-2}
