#!/bin/bash
# full-sample per-scale rates of prebuilt library variants on one box: tools/ab_full.sh "BASE X Y"   (BASE = the in-tree library)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
cp sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so
for v in $1; do
  if [ $v = BASE ]; then cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so; else cp tools/ab/lib$v.so sinddm_amd/libsinddm_hip.so; fi
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-train --no-strong --no-ab 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$v C3 full', d['full_sample']['imgs_per_sec'], [x['mpx_steps_per_sec'] for x in d['full_sample']['per_scale_this_rank']], '| C2 full', d['c2']['full_sample']['imgs_per_sec'], [x['mpx_steps_per_sec'] for x in d['c2']['full_sample']['per_scale_this_rank']])"
done | tee gpurun_out/ab_full.log
cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so
