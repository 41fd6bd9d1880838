cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_sampler_fast.py tests/test_gpu_e2e.py tests/test_gpu_forward.py -q -m gpu -p no:cacheprovider -x 2>&1 | grep -v Warning | tail -12 > gpurun_out/fast_tests.log
tail -12 gpurun_out/fast_tests.log
python bench.py --config C2 --steps 20 --warmup 3 --no-cpu --no-train > gpurun_out/bench_c2_chain.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench_c2_chain.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('C2 value',d['value'],'ms',d['ms_per_step'],'full',d['full_sample'])
else:
    print(open('gpurun_out/bench_c2_chain.log').read()[-2000:])
PY
