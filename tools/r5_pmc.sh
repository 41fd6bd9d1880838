#!/bin/bash
# round 5 PMC passes (separate runs, --kernel-trace only): tools/r5_pmc.sh <tag>   -> gpurun_out/pmc_<tag>_{c3,c2}_{sq1,grbm,fetch,write}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r05}
mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
run() { cfg=$1; n=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $ROOT/gpurun_out/pmc_${TAG}_${cfg}_$n -o pmc --output-format csv -- python $ROOT/bench.py --config ${cfg^^} --steps 2 --warmup 1 --no-full --no-cpu --no-c2 --no-train --no-strong --no-ab > $ROOT/gpurun_out/pmc_${TAG}_${cfg}_$n.log 2>&1
  echo "pass $cfg $n rc=$?"; }
for cfg in c3 c2; do
  run $cfg sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
  run $cfg sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVES
  run $cfg grbm GRBM_GUI_ACTIVE GRBM_COUNT
  run $cfg fetch FETCH_SIZE
  run $cfg write WRITE_SIZE
  cd $ROOT
  python tools/pmc_summary.py gpurun_out/pmc_${TAG}_${cfg}_ sq1 sq2 grbm fetch write 2>&1 | cut -c1-600 > gpurun_out/${TAG}_pmc_${cfg}_summary.txt
  cd /tmp
done
cd $ROOT; python tools/traffic_from_pmc.py gpurun_out/pmc_${TAG}_ $TAG > gpurun_out/${TAG}_traffic.json; tail -5 gpurun_out/${TAG}_pmc_c3_summary.txt | cut -c1-300
