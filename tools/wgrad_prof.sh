#!/bin/bash
# per-call durations of the weight-gradient kernels: tools/wgrad_prof.sh "<abl list>"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
for ab in $1; do
  SINDDM_WGRAD_ABL=$ab rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/wprof_$ab -o w -- python $ROOT/tools/train_bench.py 4 2 > $ROOT/gpurun_out/wprof_$ab.log 2>&1
  python - $ROOT/gpurun_out/wprof_$ab <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "wgrad" in r["Kernel_Name"]]
# last step only: the final 11+ wgrad calls
seq = [(r["Kernel_Name"].split("(")[0].replace("sinddm::", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "")) for r in rows]
n = len(seq) // 4
print(sys.argv[1].split("_")[-1], [(a[:12], round(b), c) for a, b, c in seq[-n:]])
PY
done
