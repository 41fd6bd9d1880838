#!/bin/bash
# A/B of the direct conv kernel's workgroup shape on the bench workload (Winograd off), one process per variant.
#   bash tools/conv_variants.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT; mkdir -p gpurun_out
OUT=gpurun_out/variants_${1:-x}.log; : > $OUT
for VAR in 4 8; do
  echo "== direct kernel, SINDDM_CONV_VAR=$VAR" >> $OUT
  SINDDM_CONV_WINO=0 SINDDM_CONV_VAR=$VAR python bench.py --steps 10 --warmup 2 --no-full --no-cpu 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'conv TF/s', r['achieved'], 'frac', r['frac'], 'avg_launch_ms', r['avg_launch_ms'])" >> $OUT
done
cat $OUT
