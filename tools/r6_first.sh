#!/bin/bash
# round 6, first GPU call: the new parity tests (G18 chain at batch 64, non-benign weights, dim 80 / 240) + the default bench line
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
TAG=${1:-r06a}
timeout 1500 python -m pytest tests/test_gpu_h2.py tests/test_gpu_chain_pin.py -q -m gpu -p no:cacheprovider -s -k "gate or behind or c3" 2>&1 | grep -v amdgpu.ids | tail -60 > gpurun_out/${TAG}_new_tests.txt
timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench.err
tail -40 gpurun_out/${TAG}_new_tests.txt; python - <<'P'
import json,sys
try:
    r=json.loads(open('gpurun_out/%s_bench_default.json'%sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/r06a_bench_default.json').read().strip().splitlines()[-1])
    print({k:r[k] for k in ('value','ms_per_step')}, r['roofline']['frac'], r['roofline']['avg_launch_ms'], r.get('train',{}).get('ms_per_step'))
except Exception as e: print('bench parse', e)
P
