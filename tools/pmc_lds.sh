#!/bin/bash
# LDS / MFMA counters of the Winograd kernels at C3 (two passes):  tools/pmc_lds.sh <tag>
TAG=${1:-x}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
CMD="python $ROOT/bench.py --config C3 --steps 2 --warmup 1 --no-full --no-cpu --no-c2 --no-train --no-strong"
for ps in "sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES" "sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "grbm GRBM_GUI_ACTIVE GRBM_COUNT"; do
  set -- $ps; n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $ROOT/gpurun_out/pmc_${TAG}_c3_$n -o pmc --output-format csv -- $CMD > $ROOT/gpurun_out/pmc_${TAG}_c3_$n.log 2>&1
done
cd $ROOT; python tools/pmc_summary.py gpurun_out/pmc_${TAG}_c3_ sq2 sq1 grbm | grep -E "pass|wino" | tee gpurun_out/${TAG}_pmc_lds.txt
