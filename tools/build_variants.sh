#!/bin/bash
# Build A/B variants of libtokenflow_hip.so into build/variants/ (travels to the GPU box; not tracked).
# usage: tools/build_variants.sh name "-DTF_TUNE_X=1 ..." [name2 "flags2" ...]
# The variant-independent objects come from the regular in-tree build (made up to date first).  RELINK=1: keep a variant's
# existing objects and only link again (after a change to one of the variant-independent files).
set -e
cd "$(dirname "$0")/../tokenflow_amd/csrc"
make -j8 > /dev/null
OUT=../../build/variants
mkdir -p $OUT
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  if [ -z "$RELINK" ] || [ ! -f $OUT/ext_attn_$name.o ]; then
    /opt/rocm/bin/hipcc $BASE -fno-honor-nans $flags -c ext_attn.hip -o $OUT/ext_attn_$name.o &
    /opt/rocm/bin/hipcc $BASE -fno-honor-nans $flags -c ext_attn_fused.hip -o $OUT/ext_attn_fused_$name.o &
    /opt/rocm/bin/hipcc $BASE $flags -c nn_search.hip -o $OUT/nn_search_$name.o &
    wait
  fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC tf_abi.o gather_blend.o layer_norm.o head_exchange.o ddim_step.o comm.o rank_exec.o $OUT/nn_search_$name.o $OUT/ext_attn_fused_$name.o $OUT/ext_attn_$name.o -ldl -o $OUT/lib_$name.so
  echo built $OUT/lib_$name.so
done
