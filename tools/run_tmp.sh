cd $GRAFT_REPO_ROOT
rm -f gpurun_out/ab.log
python -m pytest tests/test_gpu_forward.py tests/test_gpu_parity_band.py tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -3
bash tools/ab2.sh "n2 n4" 2 "C2 C3"
python tools/w2_phase.py P 2>&1 | grep -A20 "epilogue segments"
