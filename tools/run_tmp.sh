cd $GRAFT_REPO_ROOT
rm -f gpurun_out/ab.log
bash tools/ab2.sh "b1 h1" 2 "C2 C3"
cp tools/ab/libh1.so sinddm_amd/libsinddm_hip.so
python -m pytest tests/test_gpu_forward.py -m gpu -x -q 2>&1 | tail -2
