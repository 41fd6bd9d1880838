#!/bin/bash
# round 6: parity on the conv_wh / wgrad_wh paths, same-box A/B of library variants (inference: $1, training: $2), stamps ($3)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
TAG=${4:-r06d}
timeout 1500 python -m pytest tests/test_gpu_h2.py tests/test_gpu_forward.py tests/test_gpu_sampler_shapes.py tests/test_gpu_sampler_fast.py "tests/test_gpu_train.py::test_net_backward_binary16_convs_vs_float64" "tests/test_gpu_train.py::test_weight_gradients_reproducible_run_to_run" -q -m gpu -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/${TAG}_tests.txt
tail -6 gpurun_out/${TAG}_tests.txt
[ -n "$1" ] && { bash tools/ab_libs.sh "$1" 2; cp gpurun_out/ab_libs.log gpurun_out/${TAG}_ab.txt; }
if [ -n "$2" ]; then
  cp sinddm_amd/libsinddm_hip.so /tmp/lib_keep.so
  for r in 1 2; do for v in $2; do
    if [ $v = BASE ]; then cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so; else cp tools/ab/lib$v.so sinddm_amd/libsinddm_hip.so; fi
    timeout 400 python bench.py --config C2 --steps 5 --warmup 2 --no-cpu --no-full --no-strong --no-ab --no-c2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); t = d['train']
print('$v', 'train ms/step', t['ms_per_step'], 'wgrad', t.get('wgrad_roofline', {}).get('kernel_mix'), 'loop', t.get('train_loop', {}).get('ms_per_optimizer_step'), t.get('train_loop', {}).get('scale_picks'))"
  done; done | tee gpurun_out/${TAG}_ab_train.txt
  cp /tmp/lib_keep.so sinddm_amd/libsinddm_hip.so
fi
if [ -n "$3" ]; then timeout 600 python tools/wh_seg.py $3 > gpurun_out/${TAG}_wh_seg.txt 2>&1; grep -A3 "launch 2\|launch 6" gpurun_out/${TAG}_wh_seg.txt | cut -c1-330; fi
