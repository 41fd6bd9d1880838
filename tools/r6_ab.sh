#!/bin/bash
# round 6: parity of the in-tree library on the conv_wh paths + same-box A/B of library variants: tools/r6_ab.sh "OLD T1" [rounds] [tag]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
TAG=${3:-r06b}
timeout 1200 python -m pytest tests/test_gpu_h2.py tests/test_gpu_forward.py tests/test_gpu_sampler_shapes.py -q -m gpu -p no:cacheprovider -x 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/${TAG}_tests.txt
tail -5 gpurun_out/${TAG}_tests.txt
bash tools/ab_libs.sh "$1" ${2:-2}
cp gpurun_out/ab_libs.log gpurun_out/${TAG}_ab.txt
