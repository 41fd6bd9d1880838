#!/bin/bash
# forward-only variant build (the backward object is reused): tools/build_variant_fwd.sh <name> [extra hipcc flags...] -> tools/ab/lib<name>.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/tools/ab
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-spill-vgpr-to-agpr=0 -fPIC -I $ROOT/include"
[ -f $ROOT/tools/ab/bwd.o ] || /opt/rocm/bin/hipcc $F -c $ROOT/sinddm_amd/csrc/sinddm_bwd.hip -o $ROOT/tools/ab/bwd.o
/opt/rocm/bin/hipcc $F "$@" -c $ROOT/sinddm_amd/csrc/sinddm_fwd.hip -o $ROOT/tools/ab/fwd_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $ROOT/tools/ab/fwd_$NAME.o $ROOT/tools/ab/bwd.o -o $ROOT/tools/ab/lib$NAME.so
echo built tools/ab/lib$NAME.so
