#!/bin/bash
# instruction-count PMC pass of the training step (binary16 kernels): MFMA / VALU instruction counts, busy cycles
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r05t}
mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
run() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $ROOT/gpurun_out/pmc_${TAG}_$n -o pmc --output-format csv -- python $ROOT/tools/train_bench.py 4 2 > $ROOT/gpurun_out/pmc_${TAG}_$n.log 2>&1
  echo "pass $n rc=$?"; }
run sq1 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
cd $ROOT
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_ sq1 sq2 2>&1 | grep -E "==|wgrad_wh|conv_wh|dwconv5_wgrad" | cut -c1-300 | tee gpurun_out/${TAG}_train_pmc_summary.txt
