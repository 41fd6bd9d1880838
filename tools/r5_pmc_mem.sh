#!/bin/bash
# memory-side PMC passes of the C3 step (L2 hit rate, L1->L2 request latency, TA busy)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r05m}
mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
run() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $ROOT/gpurun_out/pmc_${TAG}_$n -o pmc --output-format csv -- python $ROOT/bench.py --config C3 --steps 2 --warmup 1 --no-full --no-cpu --no-c2 --no-train --no-strong --no-ab > $ROOT/gpurun_out/pmc_${TAG}_$n.log 2>&1
  echo "pass $n rc=$?"; }
run tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum
cd $ROOT
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_ tcp2 tcc ta 2>&1 | grep -E "==|conv_wh|dwconv" | cut -c1-400 | tee gpurun_out/${TAG}_mem_summary.txt
