#!/bin/bash
# build a library variant for A/B runs: tools/build_variant.sh <name> [extra hipcc flags...]  -> tools/ab/lib<name>.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/tools/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-spill-vgpr-to-agpr=0 -fPIC -shared -I $ROOT/include "$@" \
  $ROOT/sinddm_amd/csrc/sinddm_fwd.hip $ROOT/sinddm_amd/csrc/sinddm_bwd.hip -o $ROOT/tools/ab/lib$NAME.so
echo built tools/ab/lib$NAME.so
