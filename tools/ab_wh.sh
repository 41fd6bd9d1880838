#!/bin/bash
# A/B of conv_wh ablation builds (results wrong: finiteness check off): tools/ab_wh.sh "W0 W1 ..." 
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
cp sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so
for v in $1; do
  cp tools/ab/lib$v.so sinddm_amd/libsinddm_hip.so
  SINDDM_BENCH_NOFINITE=1 timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu --no-full --no-train --no-strong --no-c2 --no-ab 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('$v', 'C3 ms/step', d['ms_per_step'], 'conv launch ms', r['avg_launch_ms'], 'W', r['power']['socket_w'], 'MHz', r['power']['sclk_mhz'])"
done | tee gpurun_out/ab_wh.log
cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so
