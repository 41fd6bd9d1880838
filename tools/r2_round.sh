#!/bin/bash
# round-2 GPU box script: parity-band tests first (new), then the rest, then baseline benches.
TAG=${1:-r02a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd $ROOT
python -m pytest tests/test_gpu_parity_band.py -q -m gpu -p no:cacheprovider -s 2>&1 | tail -40 > gpurun_out/band_$TAG.log
python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_parity_band.py 2>&1 | tail -15 > gpurun_out/tests_$TAG.log
python bench.py --no-cpu > gpurun_out/bench_c2_$TAG.log 2>&1
python bench.py --config C3 --steps 5 --warmup 1 --no-cpu --no-full > gpurun_out/bench_c3_$TAG.log 2>&1
tail -30 gpurun_out/band_$TAG.log; tail -4 gpurun_out/tests_$TAG.log; tail -1 gpurun_out/bench_c2_$TAG.log; tail -1 gpurun_out/bench_c3_$TAG.log
