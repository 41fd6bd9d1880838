#!/bin/bash
# parity of a variant library on the conv_wh paths, then same-box A/B, then stamps: tools/r6_ab3.sh <variant under test> "<A/B list>" [stamps variant] [tag]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
TAG=${4:-r06g}
cp sinddm_amd/libsinddm_hip.so /tmp/lib_keep0.so
cp tools/ab/lib$1.so sinddm_amd/libsinddm_hip.so
timeout 1500 python -m pytest tests/test_gpu_h2.py tests/test_gpu_forward.py tests/test_gpu_sampler_shapes.py "tests/test_gpu_train.py::test_net_backward_binary16_convs_vs_float64" "tests/test_gpu_chain_pin.py::test_full_chain_c2_t1000_batch16_golden" -q -m gpu -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/${TAG}_tests.txt
cp /tmp/lib_keep0.so sinddm_amd/libsinddm_hip.so
tail -4 gpurun_out/${TAG}_tests.txt
bash tools/ab_libs.sh "$2" 2; cp gpurun_out/ab_libs.log gpurun_out/${TAG}_ab.txt
if [ -n "$3" ]; then timeout 600 python tools/wh_seg.py $3 > gpurun_out/${TAG}_wh_seg.txt 2>&1; grep -A3 "launch 2\|launch 6" gpurun_out/${TAG}_wh_seg.txt | cut -c1-330; fi
