#!/bin/bash
# per-scale rates of the full C2 / C3 samples under the three 3x3 paths (threshold calibration)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
for v in 0 1 3; do
  timeout 900 python bench.py --h2 $v --steps 5 --warmup 1 --no-cpu --no-train --no-strong --no-ab 2>&1 | tail -1 > gpurun_out/r5s_$v.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r5s_$v.json"))
print("H2=$v C3", d["ms_per_step"], d["full_sample"]["imgs_per_sec"], [(x["size"], x["mpx_steps_per_sec"]) for x in d["full_sample"]["per_scale_this_rank"]])
print("H2=$v C2", d["c2"]["ms_per_step"], d["c2"]["full_sample"]["imgs_per_sec"], [(x["size"], x["mpx_steps_per_sec"]) for x in d["c2"]["full_sample"]["per_scale_this_rank"]])
PY
done | tee gpurun_out/r5_scales.txt
