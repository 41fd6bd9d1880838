#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out; rm -f gpurun_out/ab.log
timeout 600 python -m pytest tests/test_gpu_wino4.py tests/test_gpu_forward.py tests/test_gpu_parity_band.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -5
timeout 900 bash tools/ab2.sh "OLD NEW V3" 2 "C2 C3"
timeout 300 python tools/w4_seg.py T 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r4_w4_seg_v3.txt | head -24
