#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_forward.py tests/test_gpu_parity_band.py tests/test_gpu_sampler_fast.py tests/test_gpu_sampler_shapes.py tests/test_gpu_chain_pin.py tests/test_gpu_wino4.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -25
for s in 0 1 3 4; do python tools/scale_chain_profile.py C2 $s 16 2>&1 | grep -v amdgpu.ids; done
for s in 0 1 2 5; do python tools/scale_chain_profile.py C3 $s 64 2>&1 | grep -v amdgpu.ids; done
