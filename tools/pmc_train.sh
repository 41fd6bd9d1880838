#!/bin/bash
# PMC passes over the training step (separate runs, --kernel-trace only; no sys/hip traces with --pmc).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}
mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
run() {  # name, counters...
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $ROOT/gpurun_out/pmct_${TAG}_$n -o pmc --output-format csv -- python $ROOT/tools/train_bench.py 4 1 > $ROOT/gpurun_out/pmct_${TAG}_$n.log 2>&1
  echo "pass $n rc=$?"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVES
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd $ROOT; python tools/pmc_summary.py gpurun_out/pmct_${TAG}_ sq1 sq2 fetch write grbm > gpurun_out/pmct_${TAG}_summary.txt; cat gpurun_out/pmct_${TAG}_summary.txt | cut -c1-400
