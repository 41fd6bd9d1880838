#!/bin/bash
# round 5: binary16 gate tests + the bench's training leg (A/B of the switch) after a kernel edit
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
TAG=${1:-r5h}
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_dp_train.py tests/test_gpu_chain_pin.py -q -x -s -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -16 > gpurun_out/h2chk_$TAG.log
rm -f gpurun_out/train_ab_$TAG.txt
for m in 3 0 3 0; do
  timeout 600 python bench.py --config C2 --steps 5 --warmup 2 --no-full --no-cpu --no-strong --no-ab --no-c2 --h2 $m 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['train']
print('h2=$m train ms', t['ms_per_step'], 'path', t.get('train_path'), 'wgrad', t['wgrad_roofline']['ms_per_step'], {k: (v['launches'], v['ms_per_step'], v['frac']) for k, v in t['wgrad_roofline']['kernel_mix'].items()}, 'convs', t['conv_roofline']['ms_per_step'], {k: (v['launches'], v['ms_per_step'], v['frac']) for k, v in t['conv_roofline']['kernel_mix'].items()}, '| C2 step', d['ms_per_step'])" >> gpurun_out/train_ab_$TAG.txt
done
cat gpurun_out/h2chk_$TAG.log; cat gpurun_out/train_ab_$TAG.txt
