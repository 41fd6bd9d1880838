#!/bin/bash
# One GPU-box round: parity tests, smoke, benchmark, rocprofv3 kernel stats.  Run through
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh <tag>'
# Outputs land in gpurun_out/ (merged back); copy what should be judged into profiles/.
TAG=${1:-r}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd $ROOT
python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/tests_$TAG.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1
python bench.py > gpurun_out/bench_$TAG.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_$TAG -o bench -- python $ROOT/bench.py --steps 10 --warmup 2 --no-full --no-cpu --no-strong > $ROOT/gpurun_out/prof_$TAG.log 2>&1
cd $ROOT
tail -4 gpurun_out/tests_$TAG.log; tail -2 gpurun_out/smoke_$TAG.log; tail -1 gpurun_out/bench_$TAG.log; tail -2 gpurun_out/prof_$TAG.log
find gpurun_out/prof_$TAG -type f | head
