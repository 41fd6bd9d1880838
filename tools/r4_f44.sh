#!/bin/bash
# F(4x4) kernel: parity tests, then A/B of the run-time switch on one box (same library): tools/r4_f44.sh [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_wino6.py -x -q 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r4_f44_tests.txt
for r in $(seq 1 ${1:-2}); do for v in 0 1; do
  timeout 600 python bench.py --f44 $v --steps 10 --warmup 2 --no-cpu --no-full --no-train --no-strong 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('F44=$v', 'C3 ms/step', d['ms_per_step'], 'exec TF/s', r['achieved'], 'alg TF/s', r['algorithmic_tflops'], r.get('power'), '| C2 ms/step', d['c2']['ms_per_step'])"
done; done | tee gpurun_out/r4_f44_ab.log
