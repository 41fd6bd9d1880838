#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-x}
cd $ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_train.py -q -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/traintests_$TAG.log
cat gpurun_out/traintests_$TAG.log
for s in 0 2 4; do python tools/train_bench.py $s 5 2>&1 | tail -1; done | tee gpurun_out/trainbench_$TAG.log
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_train_$TAG -o train -- python $ROOT/tools/train_bench.py 4 3 > $ROOT/gpurun_out/train_$TAG.log 2>&1
