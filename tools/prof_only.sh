#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-x}
mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_$TAG -o bench -- python $ROOT/bench.py --steps 10 --warmup 2 --no-full --no-cpu > $ROOT/gpurun_out/prof_$TAG.log 2>&1
tail -1 $ROOT/gpurun_out/prof_$TAG.log | cut -c1-300
