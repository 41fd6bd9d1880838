#!/bin/bash
# HBM traffic of the Winograd kernels at C3 for prebuilt library variants: tools/pmc_traffic_ab.sh "<variants>"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
cp $ROOT/sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so
CMD="python $ROOT/bench.py --config ${CFG:-C3} --steps 2 --warmup 1 --no-full --no-cpu --no-c2 --no-train --no-strong"
for v in $1; do
  cp $ROOT/tools/ab/lib$v.so $ROOT/sinddm_amd/libsinddm_hip.so
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $ROOT/gpurun_out/pmctr_${v}_$c
    timeout 600 rocprofv3 --kernel-trace --pmc $c -d $ROOT/gpurun_out/pmctr_${v}_$c -o pmc --output-format csv -- $CMD > $ROOT/gpurun_out/pmctr_${v}_$c.log 2>&1
    echo "== $v $c"; (cd $ROOT; python tools/pmc_summary.py gpurun_out/pmctr_${v}_ $c | grep wino4 | cut -c1-200)
  done
done | tee $ROOT/gpurun_out/pmc_traffic_ab.txt
cp /tmp/lib_keep.so $ROOT/sinddm_amd/libsinddm_hip.so
