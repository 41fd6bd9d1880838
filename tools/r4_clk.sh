#!/bin/bash
# sustained clock per kernel (GRBM_GUI_ACTIVE / 8 XCDs / duration) of library variants: tools/r4_clk.sh "BASE SKEW"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
cp sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so
for v in $1; do
  cp tools/ab/lib$v.so sinddm_amd/libsinddm_hip.so
  (cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $ROOT/gpurun_out/pmc_clk_$v -o pmc --output-format csv -- python $ROOT/bench.py --steps 3 --warmup 1 --no-full --no-cpu > $ROOT/gpurun_out/pmc_clk_$v.log 2>&1)
  echo "== $v"; python tools/pmc_summary.py gpurun_out/pmc_clk_ $v | grep -i "wino4\|dwconv5_rows\|conv1x1" | awk '{for(i=1;i<=NF;i++){if($i~/^avg_ns=/){split($i,a,"=");ns=a[2]} if($i~/^GRBM_GUI_ACTIVE=/){split($i,b,"=");g=b[2]}} printf "%s  clock %.3f GHz (per XCD)\n", $0, g/8/ns}'
done | tee gpurun_out/r4_clk.txt
cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so
