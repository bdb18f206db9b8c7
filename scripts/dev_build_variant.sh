#!/bin/bash
# Developer: an A/B build of the library with extra -D flags on ONE source (the other objects are the shipped build's):
#   bash scripts/dev_build_variant.sh <tag> <source.hip> [-DFLAG ...]   ->  ab_libs/libigcn_hip_<tag>.so   (load it with IGCN_LIB_PATH)
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
mkdir -p ab_libs/obj
extra=""; [ "$src" = score_topk.hip ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -Iinclude -Iigcn_cf_amd/csrc $extra "$@" \
  -c igcn_cf_amd/csrc/$src -o ab_libs/obj/${src%.hip}_$tag.o
objs=""
for f in spmm bpr score_topk topk_order sampler csr_util; do
  if [ "$f.hip" = "$src" ]; then objs="$objs ab_libs/obj/${f}_$tag.o"; else objs="$objs igcn_cf_amd/csrc/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_libs/libigcn_hip_$tag.so $objs
echo ab_libs/libigcn_hip_$tag.so
