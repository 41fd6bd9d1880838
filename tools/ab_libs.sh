#!/bin/bash
# A/B of prebuilt library variants on one box (C3 + C2 steps): tools/ab_libs.sh "BASE X Y" [rounds]   (BASE = the in-tree library)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
cp sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so
for r in $(seq 1 ${2:-1}); do for v in $1; do
  if [ $v = BASE ]; then cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so; else cp tools/ab/lib$v.so sinddm_amd/libsinddm_hip.so; fi
  timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu --no-full --no-train --no-strong --no-ab 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('$v', 'C3 ms/step', d['ms_per_step'], 'conv launch ms', r['avg_launch_ms'], 'W', r['power']['socket_w'], 'MHz', r['power']['sclk_mhz'], '| C2 ms/step', d['c2']['ms_per_step'])"
done; done | tee gpurun_out/ab_libs.log
cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so
