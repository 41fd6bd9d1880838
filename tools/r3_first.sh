#!/bin/bash
# round 3: parity tests that reach conv_wino4, then A/B of prebuilt variants:  tools/r3_first.sh "<variants>" [test files]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
TESTS=${2:-"tests/test_gpu_wino4.py tests/test_gpu_parity_band.py tests/test_gpu_forward.py"}
timeout 1200 python -m pytest $TESTS -x -q -m gpu -p no:cacheprovider -k "not full_chain_c2" 2>&1 | tail -15 > gpurun_out/r3_first_tests.log
tail -5 gpurun_out/r3_first_tests.log
rm -f gpurun_out/ab.log
timeout 1200 bash tools/ab2.sh "${1:-v3 v4}" 1 "C2 C3"
