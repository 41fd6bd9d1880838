cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_forward.py tests/test_gpu_parity_band.py tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -3
python tools/scale_times.py C2 16 2>&1 | tail -6
