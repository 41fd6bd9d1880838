#!/bin/bash
# PMC passes for the conv kernel (separate runs, --kernel-trace only; no sys/hip traces with --pmc).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}
mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $ROOT/gpurun_out/counters_list.txt 2>&1
run() {  # name, counters...
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $ROOT/gpurun_out/pmc_${TAG}_$n -o pmc --output-format csv -- python $ROOT/bench.py --steps 2 --warmup 1 --no-full --no-cpu > $ROOT/gpurun_out/pmc_${TAG}_$n.log 2>&1
  echo "pass $n rc=$?"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVES
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $ROOT; ls gpurun_out/pmc_${TAG}_*; grep -c . gpurun_out/counters_list.txt
