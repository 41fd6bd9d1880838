#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-x}
mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_train_$TAG -o train -- python $ROOT/tools/train_bench.py 4 3 > $ROOT/gpurun_out/train_$TAG.log 2>&1
tail -2 $ROOT/gpurun_out/train_$TAG.log
