#!/bin/bash
# round 4, first GPU call: new parity pins + RCCL one-rank test + A/B of the conv_wino4 epilogue + item timing
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out; rm -f gpurun_out/ab.log
timeout 900 python -m pytest tests/test_gpu_wino4.py tests/test_gpu_forward.py tests/test_gpu_chain_pin.py tests/test_gpu_rccl.py -x -q -m gpu -s 2>&1 | grep -v amdgpu.ids | tail -40 > gpurun_out/r4_first_tests.log
tail -15 gpurun_out/r4_first_tests.log
timeout 600 bash tools/ab2.sh "OLD NEW" 1 "C2 C3"
timeout 300 python tools/w4_seg.py T 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r4_w4_seg.txt
