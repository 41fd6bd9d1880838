#!/bin/bash
# kernel stats of 40 reverse steps at given image sizes: tools/shape_kstats.sh "133x177 133x176" <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${2:-x}; mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
for hw in $1; do
  h=${hw%%x*}; w=${hw##*x}
  timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/shk_${TAG}_$hw -o k -- python $ROOT/tools/shape_step_profile.py $h $w > $ROOT/gpurun_out/shk_${TAG}_$hw.log 2>&1
  echo "== $hw"; python $ROOT/tools/rocprof_summary.py $ROOT/gpurun_out/shk_${TAG}_$hw/k_results.db 2>&1 | sed -n 3,14p | cut -c1-150
done | tee $ROOT/gpurun_out/shk_${TAG}.txt
