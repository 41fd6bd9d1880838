cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_forward.py tests/test_gpu_train.py tests/test_gpu_parity_band.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v Warning | tail -6
python tools/step_overhead.py 16 2>&1 | tail -5
