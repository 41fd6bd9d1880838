#!/bin/bash
# kernel stats of 55 reverse steps at pyramid scales of C2 (batch 16): tools/scale_kstats.sh "<scales>" <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${2:-x}; mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
for s in $1; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/sk_${TAG}_s$s -o k -- python $ROOT/tools/scale_step_profile.py $s > $ROOT/gpurun_out/sk_${TAG}_s$s.log 2>&1
  echo "== scale $s"; python $ROOT/tools/rocprof_summary.py $ROOT/gpurun_out/sk_${TAG}_s$s/k_results.db 2>&1 | head -16 | cut -c1-150
done | tee $ROOT/gpurun_out/sk_${TAG}.txt
