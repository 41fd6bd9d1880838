#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-x}
cd $ROOT; mkdir -p gpurun_out
OUT=gpurun_out/winoabl_$TAG.log; : > $OUT
for ABL in ${2:-0 1 2 3}; do
  echo "== ABL=$ABL" >> $OUT
  SINDDM_WINO_ABL=$ABL python bench.py --steps 10 --warmup 2 --no-cpu --no-full 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('ms/step', d['ms_per_step'], 'conv TF/s', r['achieved'], 'avg_launch_ms', r['avg_launch_ms'], 'launches', r['launches'])" >> $OUT
done
cat $OUT
