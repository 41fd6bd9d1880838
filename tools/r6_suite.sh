#!/bin/bash
# round 6: whole GPU suite + smoke (+ optional per-segment stamps of conv_wh from a -DWH_TIMING variant): tools/r6_suite.sh <tag> [WT]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
TAG=${1:-r06c}
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -40 > gpurun_out/${TAG}_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1
tail -12 gpurun_out/${TAG}_gpu_tests.txt; tail -2 gpurun_out/${TAG}_smoke.txt
if [ -n "$2" ]; then timeout 600 python tools/wh_seg.py $2 > gpurun_out/${TAG}_wh_seg.txt 2>&1; tail -30 gpurun_out/${TAG}_wh_seg.txt; fi
