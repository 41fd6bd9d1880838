#!/bin/bash
# per-scale rates of full samples (C3 batch 64, C2 batch 16) for library variants on one box: tools/r6_scales.sh "BASE P7 ..." [tag]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
TAG=${2:-r06s}
cp sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so
for v in $1; do
  if [ $v = BASE ]; then cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so; else cp tools/ab/lib$v.so sinddm_amd/libsinddm_hip.so; fi
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu --no-train --no-strong --no-ab 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
f3 = d['full_sample']; f2 = d['c2']['full_sample']
print('$v', 'C3 img/s', f3['imgs_per_sec'], [p['mpx_steps_per_sec'] for p in f3['per_scale_this_rank']], '| C2 img/s', f2['imgs_per_sec'], [p['mpx_steps_per_sec'] for p in f2['per_scale_this_rank']])"
done | tee gpurun_out/${TAG}_scales.txt
cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so
