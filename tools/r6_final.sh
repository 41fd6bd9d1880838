#!/bin/bash
# round 6, full GPU round of the final tree: whole -m gpu suite (the parity figures the tests print are kept), smoke, the default
# bench line (what the driver runs).   gpurun --timeout 3300 -- 'bash tools/r6_final.sh r06f'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
TAG=${1:-r06f}
timeout 2700 python -m pytest tests -q -m gpu -p no:cacheprovider -rP > /tmp/${TAG}_full.txt 2>&1
grep -v amdgpu.ids /tmp/${TAG}_full.txt | tail -12 > gpurun_out/${TAG}_gpu_tests.txt
grep -a "rel-L2\|vs float64\|fused steps\|worst chain" /tmp/${TAG}_full.txt | cut -c1-300 > gpurun_out/${TAG}_parity_figures.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1
timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench.err
tail -6 gpurun_out/${TAG}_gpu_tests.txt; tail -2 gpurun_out/${TAG}_smoke.txt; wc -l gpurun_out/${TAG}_parity_figures.txt
python - <<P
import json
r=json.loads(open('gpurun_out/${TAG}_bench_default.json').read().strip().splitlines()[-1])
rf=r['roofline']
print({k:r[k] for k in ('value','ms_per_step')}, 'frac', rf['frac'], 'launch ms', rf['avg_launch_ms'], 'mfma_busy', rf.get('mfma_busy',{}).get('frac_of_peak'), 'traffic_stale', rf.get('traffic_stale'))
print('full', r['full_sample'].get('images_per_sec'), 'c2', r['c2']['ms_per_step'], r['c2'].get('full_sample',{}).get('images_per_sec'), 'train', r['train']['ms_per_step'], r['train'].get('train_loop'))
print('fp32 path', r['fp32_mfma_path'])
P
