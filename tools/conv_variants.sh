#!/bin/bash
# A/B of the conv kernel tuning knobs on the bench workload (one process per variant).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT; mkdir -p gpurun_out
OUT=gpurun_out/variants_${1:-x}.log; : > $OUT
for NT in 4 2; do for VAR in 4 5; do
  echo "== NT=$NT VAR=$VAR" >> $OUT
  SINDDM_CONV_NT=$NT SINDDM_CONV_VAR=$VAR python bench.py --steps 10 --warmup 2 --no-full --no-cpu 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'conv TF/s', r['achieved'], 'frac', r['frac'], 'avg_launch_ms', r['avg_launch_ms'])" >> $OUT
done; done
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3 >> $OUT
python bench.py --no-cpu 2>&1 | tail -1 >> $OUT
cat $OUT
