#!/bin/bash
# round 5: sustained matrix-pipe rates under the power limit (ubench) + PMC passes of conv_h2 at C3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/h2_power tools/ubench/h2_power.hip -lpthread 2>/dev/null
timeout 120 /tmp/h2_power 2.5 | tee gpurun_out/r5b_h2_power.txt
cd /tmp; export TMPDIR=/tmp
run() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $ROOT/gpurun_out/pmc_r5b_$n -o pmc --output-format csv -- python $ROOT/bench.py --steps 2 --warmup 1 --no-full --no-cpu --no-c2 --no-train --no-strong > $ROOT/gpurun_out/pmc_r5b_$n.log 2>&1
  echo "pass $n rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVES
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd $ROOT
python tools/pmc_summary.py gpurun_out/pmc_r5b_ sq1 sq2 grbm 2>&1 | grep -E "pass|conv_h2|dwconv|conv1x1" | tee gpurun_out/r5b_pmc_summary.txt
