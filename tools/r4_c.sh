#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out; rm -f gpurun_out/ab.log
timeout 900 python -m pytest tests/test_gpu_wino4.py tests/test_gpu_forward.py tests/test_gpu_parity_band.py tests/test_gpu_sampler_fast.py tests/test_gpu_sampler_shapes.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -25
timeout 900 bash tools/ab2.sh "$1" ${2:-1} "C3 C2"
cp gpurun_out/ab.log gpurun_out/${3:-r4_ab}.log
