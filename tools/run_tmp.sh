cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -3
python tools/train_bench.py 4 5 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_tr -o tr -- python $R/tools/train_bench.py 4 3 > $R/gpurun_out/prof_tr.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_tr -name "*results.db" | head -1) 2>&1 | head -8 | tail -5
