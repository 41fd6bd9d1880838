#!/bin/bash
# socket power / clocks sampled while the C3 bench loop runs: is the dominant kernel power-limited?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
(python bench.py --steps 150 --warmup 2 --no-cpu --no-full > gpurun_out/power_bench.log 2>&1 &)
for i in $(seq 1 70); do rocm-smi --showpower --showclocks 2>&1 | grep -i "power (W)\|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.3; done | uniq -c
wait
tail -n 1 gpurun_out/power_bench.log | cut -c1-200
