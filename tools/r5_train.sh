#!/bin/bash
# round 5: the training path on the binary16 hi/lo Winograd kernel -- parity tests, then the bench's training leg A/B (switch 3 vs 0)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
TAG=${1:-r5t}
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_dp_train.py tests/test_gpu_h2.py tests/test_gpu_chain_pin.py tests/test_gpu_sampler_shapes.py -q -x -s -p no:cacheprovider --durations=5 2>&1 | grep -v amdgpu.ids | tail -30 > gpurun_out/train_tests_$TAG.log
for m in 3 0 3 0; do
  timeout 600 python bench.py --config C2 --steps 5 --warmup 2 --no-full --no-cpu --no-strong --no-ab --no-c2 --h2 $m 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['train']
print('h2=$m train ms', t['ms_per_step'], 'wgrad', t['wgrad_roofline']['ms_per_step'], 'convs', t['conv_roofline']['ms_per_step'], t['conv_roofline'].get('kernel'), '| C2 step', d['ms_per_step'])" >> gpurun_out/train_ab_$TAG.txt
done
cat gpurun_out/train_tests_$TAG.log | tail -25; cat gpurun_out/train_ab_$TAG.txt
