#!/bin/bash
# kernel stats of 55 reverse steps at C2 scale <s> for prebuilt variants: tools/kstat_ab.sh "<variants>" <scale> <kernel grep>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
cp $ROOT/sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so
for v in $1; do
  cp $ROOT/tools/ab/lib$v.so $ROOT/sinddm_amd/libsinddm_hip.so
  rm -rf $ROOT/gpurun_out/ksab_$v
  timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/ksab_$v -o k -- python $ROOT/tools/scale_step_profile.py ${2:-4} 64 > $ROOT/gpurun_out/ksab_$v.log 2>&1
  echo "== $v"; python $ROOT/tools/rocprof_summary.py $ROOT/gpurun_out/ksab_$v/k_results.db 2>&1 | grep -E "${3:-c3_gelu}" | head -3 | cut -c1-140
done
cp /tmp/lib_keep.so $ROOT/sinddm_amd/libsinddm_hip.so
