#!/bin/bash
# round 5: parity tests of the binary16 kernels, then C3/C2 A/B of the run-time switch on one box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_h2.py -x -q -s -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -25 | tee gpurun_out/r5a_h2_tests.txt
for v in 3 1 0; do
  timeout 600 python bench.py --h2 $v --steps 10 --warmup 2 --no-cpu --no-full --no-train --no-strong --no-ab 2>&1 | tail -1 > gpurun_out/r5a_bench_h2_$v.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r5a_bench_h2_$v.json")); r=d["roofline"]
print("H2=$v", "C3 ms/step", d["ms_per_step"], "conv launch ms", r["avg_launch_ms"], "mix", {k: (x["launches"], x["avg_launch_ms"]) for k, x in r["kernel_mix"].items()}, r.get("power"), "| C2 ms/step", d.get("c2", {}).get("ms_per_step"))
PY
done | tee gpurun_out/r5a_ab.txt
