cd $GRAFT_REPO_ROOT
S=$(date +%s)
python bench.py 2> gpurun_out/bench_default.err > gpurun_out/bench_default.out
echo "rc=$? wall=$(( $(date +%s) - S )) s"
tail -1 gpurun_out/bench_default.out
tail -3 gpurun_out/bench_default.err
