#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
timeout 120 tools/ab/acc_drain 2>&1 | tee gpurun_out/r4_ubench_acc_drain.txt
timeout 300 python -m pytest tests/test_gpu_chain_pin.py -x -q -m gpu -k "g4" 2>&1 | grep -v amdgpu.ids | tail -15
