#!/bin/bash
# forward/train parity tests + sampler bench (+ training step) after a Winograd kernel change
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-x}
cd $ROOT; mkdir -p gpurun_out
OUT=gpurun_out/wino_$TAG.log; : > $OUT
python -m pytest tests/test_gpu_forward.py tests/test_gpu_train.py tests/test_gpu_e2e.py -q -p no:cacheprovider 2>&1 | tail -4 >> $OUT
for i in 1 2; do
python bench.py --steps 10 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'conv TF/s', r['achieved'], 'frac', r['frac'], 'exec', r.get('executed_tflops'), 'launches', r['launches'], 'full', d['full_sample'])" >> $OUT
done
python tools/train_bench.py 4 5 2>&1 | tail -1 >> $OUT
cat $OUT
