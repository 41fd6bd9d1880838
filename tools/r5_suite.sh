#!/bin/bash
# full GPU suite + smoke on one box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
TAG=${1:-r5}
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | grep -v amdgpu.ids | tail -25 > gpurun_out/tests_$TAG.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1
tail -8 gpurun_out/tests_$TAG.log; tail -2 gpurun_out/smoke_$TAG.log
