#!/bin/bash
# round 6 evidence for profiles/: rocprofv3 kernel-trace stats of the bench legs (C3 headline, C2, training step) and the PMC
# passes of the C3 step (separate runs, --kernel-trace only with --pmc), traffic.json rebuilt from the FETCH / WRITE passes.
#   gpurun --timeout 2400 -- 'bash tools/r6_prof.sh r06p'
TAG=${1:-r06p}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
B3="python $ROOT/bench.py --config C3 --steps 5 --warmup 1 --no-full --no-cpu --no-c2 --no-train --no-strong --no-ab"
B2="python $ROOT/bench.py --config C2 --steps 10 --warmup 2 --no-full --no-cpu --no-train --no-strong --no-ab"
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_${TAG}_c3 -o bench -- $B3 > $ROOT/gpurun_out/prof_${TAG}_c3.log 2>&1
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_${TAG}_c2 -o bench -- $B2 > $ROOT/gpurun_out/prof_${TAG}_c2.log 2>&1
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_${TAG}_train -o train -- python $ROOT/tools/train_bench.py 4 3 > $ROOT/gpurun_out/prof_${TAG}_train.log 2>&1
run() {  # cfg name counters...
  cfg=$1; n=$2; shift; shift
  cmd="$B2"; [ $cfg = c3 ] && cmd="python $ROOT/bench.py --config C3 --steps 2 --warmup 1 --no-full --no-cpu --no-c2 --no-train --no-strong --no-ab"
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $ROOT/gpurun_out/pmc_${TAG}_${cfg}_$n -o pmc --output-format csv -- $cmd > $ROOT/gpurun_out/pmc_${TAG}_${cfg}_$n.log 2>&1
  echo "pass $cfg $n rc=$?"
}
for cfg in c3 c2; do
  run $cfg fetch FETCH_SIZE
  run $cfg write WRITE_SIZE
done
run c3 sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
run c3 sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVES
run c3 grbm GRBM_GUI_ACTIVE GRBM_COUNT
run c3 l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run c3 tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
cd $ROOT
for c in c3 c2; do python tools/rocprof_summary.py $(find gpurun_out/prof_${TAG}_$c -name "*results.db" | head -1) > gpurun_out/${TAG}_bench_${c}_kernel_stats.txt 2>&1; grep "^{\"metric\"" gpurun_out/prof_${TAG}_$c.log | tail -1 > gpurun_out/${TAG}_bench_${c}.json; done
python tools/rocprof_summary.py $(find gpurun_out/prof_${TAG}_train -name "*results.db" | head -1) > gpurun_out/${TAG}_train_kernel_stats.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_c3_ fetch write sq1 sq2 grbm l2 tcp2 > gpurun_out/${TAG}_pmc_c3_summary.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_c2_ fetch write > gpurun_out/${TAG}_pmc_c2_summary.txt 2>&1
python tools/traffic_from_pmc.py gpurun_out/pmc_${TAG}_ $TAG > gpurun_out/${TAG}_traffic.json 2> gpurun_out/${TAG}_traffic.err
# the raw rocprof outputs are big: keep the summaries only
rm -rf gpurun_out/prof_${TAG}_c3 gpurun_out/prof_${TAG}_c2 gpurun_out/prof_${TAG}_train gpurun_out/pmc_${TAG}_*/
head -24 gpurun_out/${TAG}_bench_c3_kernel_stats.txt; grep "conv_wh" gpurun_out/${TAG}_pmc_c3_summary.txt | cut -c1-400; head -c 600 gpurun_out/${TAG}_traffic.json
