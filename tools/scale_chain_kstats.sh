#!/bin/bash
# kernel stats of 220 production reverse steps per pyramid scale: tools/scale_chain_kstats.sh <config> "<scales>" <batch> <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; CFG=${1:-C2}; B=${3:-16}; TAG=${4:-x}; mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
for s in $2; do
  python $ROOT/tools/scale_chain_profile.py $CFG $s $B 2>&1 | grep -v amdgpu.ids
  timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/sck_${TAG}_s$s -o k -- python $ROOT/tools/scale_chain_profile.py $CFG $s $B > $ROOT/gpurun_out/sck_${TAG}_s$s.log 2>&1
  python $ROOT/tools/rocprof_summary.py $ROOT/gpurun_out/sck_${TAG}_s$s/k_results.db 2>&1 | head -14 | cut -c1-150
done | tee $ROOT/gpurun_out/sck_${TAG}.txt
