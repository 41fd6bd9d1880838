cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_forward.py tests/test_gpu_parity_band.py tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -3
python tools/scale_times.py C2 16 2>&1 | tail -6
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_s4 -o s4 -- python $R/tools/scale0_only.py 4 16 > $R/gpurun_out/prof_s4.log 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/prof_s4 -name "*results.db" | head -1) 2>&1 | head -10 | tail -7
