cd $GRAFT_REPO_ROOT
rm -f gpurun_out/ab.log
python -m pytest tests/test_gpu_forward.py tests/test_gpu_parity_band.py -m gpu -x -q 2>&1 | tail -3
bash tools/ab2.sh "w0 w1" 2 "C2 C3"
