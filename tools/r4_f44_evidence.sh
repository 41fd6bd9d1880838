#!/bin/bash
# Evidence round for the experimental F(4x4) kernel (tools/ab/libW6F.so = build_variant.sh W6F -DSINDDM_WINO_F44_BUILD=1):
# its parity tests, the same-box A/B against conv_wino4 (same library, run-time switch), and one PMC pass (MFMA count).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
cp sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so
cp tools/ab/libW6F.so sinddm_amd/libsinddm_hip.so
timeout 900 python -m pytest tests/test_gpu_wino6.py tests/test_gpu_wino4.py tests/test_gpu_forward.py -q 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/r04e_f44_tests.txt
for r in 1 2; do for f in 0 1; do
  python bench.py --f44 $f --steps 10 --warmup 2 --no-cpu --no-full --no-train --no-strong 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('F44=$f', 'C3 ms/step', d['ms_per_step'], 'conv launch ms', r['avg_launch_ms'], 'executed TF/s', r['achieved'], 'mix', {k: v['launches'] for k, v in r['kernel_mix'].items()}, 'W', r['power']['socket_w'], 'MHz', r['power']['sclk_mhz'], '| C2 ms/step', d['c2']['ms_per_step'])"
done; done | tee gpurun_out/r04e_f44_ab.txt
(cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $ROOT/gpurun_out/pmc_f44 -o pmc --output-format csv -- python $ROOT/bench.py --f44 1 --config C3 --steps 2 --warmup 1 --no-full --no-cpu --no-c2 --no-train --no-strong > $ROOT/gpurun_out/pmc_f44.log 2>&1)
python tools/pmc_summary.py gpurun_out/pmc_f 44 | grep -i "wino" | tee gpurun_out/r04e_f44_pmc.txt
cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so
