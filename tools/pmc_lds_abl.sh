#!/bin/bash
# where do conv_wino4's LDS bank conflicts come from?  SQ LDS counters of prebuilt ablation variants (tools/ab/lib<V>.so):
#   tools/pmc_lds_abl.sh "<variants>"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
cp $ROOT/sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so
CMD="python $ROOT/bench.py --config C3 --steps 2 --warmup 1 --no-full --no-cpu --no-c2 --no-train --no-strong --no-ab"
for v in $1; do
  cp $ROOT/tools/ab/lib$v.so $ROOT/sinddm_amd/libsinddm_hip.so
  rm -rf $ROOT/gpurun_out/pmcabl_${v}_sq2
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d $ROOT/gpurun_out/pmcabl_${v}_sq2 -o pmc --output-format csv -- $CMD > $ROOT/gpurun_out/pmcabl_${v}.log 2>&1
  echo "== $v"; (cd $ROOT; python tools/pmc_summary.py gpurun_out/pmcabl_${v}_ sq2 | grep -E "conv_wh|wino4" | cut -c1-400)
done | tee $ROOT/gpurun_out/pmc_lds_abl.txt
cp /tmp/lib_keep.so $ROOT/sinddm_amd/libsinddm_hip.so
