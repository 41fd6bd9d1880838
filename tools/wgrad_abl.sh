#!/bin/bash
# wgrad tuning sweeps: tools/wgrad_abl.sh "<w3 list>" "<abl list>" "<stage list>"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out
for w3 in $1; do for ab in $2; do for sg in ${3:-1}; do
  echo "w3=$w3 abl=$ab stage=$sg"
  SINDDM_WGRAD_W3=$w3 SINDDM_WGRAD_ABL=$ab SINDDM_WGRAD_STAGE=$sg python tools/train_bench.py 4 5 2>&1 | tail -1
done; done; done | tee gpurun_out/wgrad_abl.log
