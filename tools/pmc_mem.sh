#!/bin/bash
# PMC passes on the memory side of the conv kernel (TLB, L1 stalls, L2 -> fabric): tools/pmc_mem.sh <tag> [config] [variant lib]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}; CFG=${2:-C3}
mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
if [ -n "$3" ]; then cp $ROOT/sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so; cp $ROOT/tools/ab/lib$3.so $ROOT/sinddm_amd/libsinddm_hip.so; export SINDDM_BENCH_NOFINITE=1; fi
run() {  # name, counters...
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $ROOT/gpurun_out/pmc_${TAG}_$n -o pmc --output-format csv -- python $ROOT/bench.py --config $CFG --steps 2 --warmup 1 --no-full --no-cpu --no-c2 --no-train --no-strong > $ROOT/gpurun_out/pmc_${TAG}_$n.log 2>&1
  echo "pass $n rc=$?"
}
run tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
run tlb2 TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum
run tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
run tcc2 TCC_EA_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_EA_RDREQ_LEVEL_sum
run grbm GRBM_GUI_ACTIVE
[ -n "$3" ] && cp /tmp/lib_keep.so $ROOT/sinddm_amd/libsinddm_hip.so
cd $ROOT
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_ tlb tlb2 tcp tcp2 ta tcc tcc2 grbm 2>&1 | grep -E "==|conv_wino4|dwconv|conv1x1" | cut -c1-400 | tee gpurun_out/pmc_${TAG}_mem_summary.txt
