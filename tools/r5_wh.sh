#!/bin/bash
# conv_wh iteration: its parity tests (mode 3 only) + C3 step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_h2.py -x -q -s -p no:cacheprovider -k "wh_winograd" 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r5w_tests.txt
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu --no-full --no-train --no-strong --no-ab 2>&1 | tail -1 > gpurun_out/r5w_bench.json
python - <<PY
import json
d=json.load(open("gpurun_out/r5w_bench.json")); r=d["roofline"]
print("C3 ms/step", d["ms_per_step"], "conv launch ms", r["avg_launch_ms"], "mix", {k: (x["launches"], x["avg_launch_ms"]) for k, x in r["kernel_mix"].items()}, r["power"]["socket_w"], "W", r["power"]["sclk_mhz"], "MHz | C2 ms/step", d.get("c2", {}).get("ms_per_step"))
PY
