#!/bin/bash
# memory-side PMC passes of the training step (the weight-gradient kernels): HBM-side bytes, L2 hit rate, L1->L2 latency
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r05w}
mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
run() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $ROOT/gpurun_out/pmc_${TAG}_$n -o pmc --output-format csv -- python $ROOT/tools/train_bench.py 4 2 > $ROOT/gpurun_out/pmc_${TAG}_$n.log 2>&1
  echo "pass $n rc=$?"; }
run fetch FETCH_SIZE WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
cd $ROOT
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_ fetch tcc tcp2 2>&1 | grep -E "==|wgrad_w|conv_wh" | cut -c1-300 | tee gpurun_out/${TAG}_wgrad_mem_summary.txt
