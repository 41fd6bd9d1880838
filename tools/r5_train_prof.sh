#!/bin/bash
# kernel-trace stats of the training step (tools/train_bench.py: C2 finest scale, batch 32, 2 warm-up + 5 timed steps)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r5}
mkdir -p $ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_train_$TAG -o train -- python $ROOT/tools/train_bench.py 4 5 > $ROOT/gpurun_out/prof_train_$TAG.log 2>&1
cd $ROOT
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_train_$TAG/**/train_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = open("gpurun_out/train_kernel_stats_$TAG.txt", "w")
for r in rows[:32]:
    line = f'{r["Name"][:72]:72s} {int(r["Calls"]):5d} {float(r["TotalDurationNs"]) / 7e6:8.3f} ms/step {100 * float(r["TotalDurationNs"]) / tot:5.1f}%'
    print(line); out.write(line + "\n")
PY
tail -1 gpurun_out/prof_train_$TAG.log
