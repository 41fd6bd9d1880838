#!/bin/bash
# A/B of prebuilt library variants with the F(4x4) switch on: tools/ab_f44.sh "W6 W6A1 ..." [rounds]   (name BASE = switch off)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
cp sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so
for r in $(seq 1 ${2:-1}); do for v in $1; do
  f=1; lib=$v; [ $v = BASE ] && { f=0; lib=W6; }
  cp tools/ab/lib$lib.so sinddm_amd/libsinddm_hip.so
  SINDDM_BENCH_NOFINITE=1 python bench.py --f44 $f --steps 10 --warmup 2 --no-cpu --no-full --no-train --no-strong --no-c2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('$v', 'ms/step', d['ms_per_step'], 'conv launch ms', r['avg_launch_ms'], 'exec TF/s', r['achieved'], 'W', r['power']['socket_w'] if r.get('power') else None, 'MHz', r['power']['sclk_mhz'] if r.get('power') else None)"
done; done | tee gpurun_out/ab_f44.log
cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so
