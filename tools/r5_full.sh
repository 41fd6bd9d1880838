#!/bin/bash
# round 5: full GPU suite, smoke, default bench, kernel-trace stats of a short bench
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
TAG=${1:-r5}
timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/tests_$TAG.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_$TAG.log 2>&1
tail -1 gpurun_out/bench_$TAG.log > gpurun_out/bench_$TAG.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -o bench -- python $ROOT/bench.py --steps 10 --warmup 2 --no-full --no-cpu --no-strong --no-train --no-ab > $ROOT/gpurun_out/prof_$TAG.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${TAG}_c3 -o bench -- python $ROOT/bench.py --config C3 --steps 10 --warmup 2 --no-full --no-cpu --no-strong --no-train --no-ab --no-c2 > $ROOT/gpurun_out/prof_${TAG}_c3.log 2>&1
cd $ROOT
tail -6 gpurun_out/tests_$TAG.log; tail -2 gpurun_out/smoke_$TAG.log; tail -c 600 gpurun_out/bench_$TAG.json; echo; find gpurun_out/prof_$TAG -name "*stats*" | head
