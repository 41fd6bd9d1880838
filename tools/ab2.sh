#!/bin/bash
# A/B runs of prebuilt library variants (tools/ab/lib*.so) in one box: tools/ab2.sh "A B C" [rounds] [configs]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
cp sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so
CFGS=${3:-"C2 C3"}
for r in $(seq 1 ${2:-2}); do for v in $1; do for c in $CFGS; do
  cp tools/ab/lib$v.so sinddm_amd/libsinddm_hip.so
  st=20; [ $c = C3 ] && st=4
  python bench.py --config $c --steps $st --warmup 2 --no-cpu --no-full --no-c2 --no-train --no-strong 2>&1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline()); r = d['roofline']
    print('$v', '$c', 'ms/step', d['ms_per_step'], 'wino frac', r['frac'], 'avg_launch_ms', r['avg_launch_ms'], 'share', r['share_of_step'])
except Exception as e:
    print('$v', '$c', 'FAILED', e)"
done; done; done | tee -a gpurun_out/ab.log
cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so
