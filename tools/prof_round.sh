#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats of the bench command (C3 headline + C2), a training step, and the
# PMC passes for the dominant kernel (separate runs, --kernel-trace only: no sys/hip traces together with --pmc).
#   gpurun --timeout 2400 -- 'bash tools/prof_round.sh r02a'
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
B3="python $ROOT/bench.py --config C3 --steps 5 --warmup 1 --no-full --no-cpu --no-c2 --no-train --no-strong"
B2="python $ROOT/bench.py --config C2 --steps 10 --warmup 2 --no-full --no-cpu --no-train --no-strong"
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_${TAG}_c3 -o bench -- $B3 > $ROOT/gpurun_out/prof_${TAG}_c3.log 2>&1
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_${TAG}_c2 -o bench -- $B2 > $ROOT/gpurun_out/prof_${TAG}_c2.log 2>&1
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_${TAG}_train -o train -- python $ROOT/tools/train_bench.py 4 3 > $ROOT/gpurun_out/prof_${TAG}_train.log 2>&1
run() {  # cfg name counters...
  cfg=$1; n=$2; shift; shift
  cmd="$B2"; [ $cfg = c3 ] && cmd="python $ROOT/bench.py --config C3 --steps 2 --warmup 1 --no-full --no-cpu --no-c2 --no-train --no-strong"
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $ROOT/gpurun_out/pmc_${TAG}_${cfg}_$n -o pmc --output-format csv -- $cmd > $ROOT/gpurun_out/pmc_${TAG}_${cfg}_$n.log 2>&1
  echo "pass $cfg $n rc=$?"
}
for cfg in c2 c3; do
  run $cfg sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
  run $cfg grbm GRBM_GUI_ACTIVE GRBM_COUNT
  run $cfg fetch FETCH_SIZE
  run $cfg write WRITE_SIZE
done
for cfg in c2 c3; do
  run $cfg sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVES
  run $cfg l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
done
cd $ROOT
for c in c3 c2; do python tools/rocprof_summary.py $(find gpurun_out/prof_${TAG}_$c -name "*results.db" | head -1) > gpurun_out/${TAG}_bench_${c}_kernel_stats.txt 2>&1; grep "^{\"metric\"" gpurun_out/prof_${TAG}_$c.log | tail -1 > gpurun_out/${TAG}_bench_${c}.json; done
python tools/rocprof_summary.py $(find gpurun_out/prof_${TAG}_train -name "*results.db" | head -1) > gpurun_out/${TAG}_train_kernel_stats.txt 2>&1
for c in c2 c3; do python tools/pmc_summary.py gpurun_out/pmc_${TAG}_${c}_ sq1 grbm fetch write sq2 l2 > gpurun_out/${TAG}_pmc_${c}_summary.txt 2>&1; done
head -30 gpurun_out/${TAG}_bench_c3_kernel_stats.txt; head -12 gpurun_out/${TAG}_pmc_c3_summary.txt
