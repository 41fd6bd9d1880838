cd $GRAFT_REPO_ROOT
rm -f gpurun_out/ab.log
python -m pytest tests/test_gpu_forward.py tests/test_gpu_parity_band.py tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -3
bash tools/ab2.sh "p0 n1" 2 "C2 C3"
export SINDDM_BENCH_NOFINITE=1
bash tools/ab2.sh "n1w" 1 "C2"
python bench.py --config C2 --steps 10 --warmup 2 --no-cpu --no-full --no-c2 2>&1 | tail -1
