#!/bin/bash
# A/B runs of prebuilt library variants (tools/ab/lib*.so) in one box: tools/ab.sh "A B C" [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
cp sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so
for r in $(seq 1 ${2:-2}); do for v in $1; do
  cp tools/ab/lib$v.so sinddm_amd/libsinddm_hip.so
  python bench.py --steps 10 --warmup 2 --no-cpu --no-full 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('$v', 'ms/step', d['ms_per_step'], 'conv TF/s', r['achieved'])"
done; done | tee gpurun_out/ab.log
cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so
