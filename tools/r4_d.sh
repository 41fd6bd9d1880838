#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out; rm -f gpurun_out/ab.log
export SINDDM_BENCH_NOFINITE=1
timeout 1200 bash tools/ab2.sh "$1" ${2:-1} "${3:-C3}"
cp gpurun_out/ab.log gpurun_out/${4:-r4_ab}.log
