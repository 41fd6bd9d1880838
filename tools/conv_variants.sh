#!/bin/bash
# A/B of the conv kernel tuning knobs on the bench workload (one process per variant).
#   bash tools/conv_variants.sh <tag> "<NT:VAR> <NT:VAR> ..." [test_var]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT; mkdir -p gpurun_out
OUT=gpurun_out/variants_${1:-x}.log; : > $OUT
for cfg in ${2:-"0:4"}; do
  NT=${cfg%%:*}; VAR=${cfg##*:}
  echo "== NT=$NT VAR=$VAR" >> $OUT
  SINDDM_CONV_NT=$NT SINDDM_CONV_VAR=$VAR python bench.py --steps 10 --warmup 2 --no-full --no-cpu 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'conv TF/s', r['achieved'], 'frac', r['frac'], 'avg_launch_ms', r['avg_launch_ms'])" >> $OUT
done
if [ -n "$3" ]; then
  SINDDM_CONV_VAR=$3 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3 >> $OUT
fi
cat $OUT
