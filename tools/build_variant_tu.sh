#!/bin/bash
# build_variant_tu.sh <tu name without extension, e.g. elementwise> <out.so> [extra hipcc flags]:
# link a variant library that differs from the product only in ONE translation unit compiled with extra flags
set -e
TU=$1; OUT=$2; shift 2
cd "$(dirname "$0")/.."
mkdir -p build/variants
OBJ=build/variants/$(basename $OUT .so).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I pivotcvae_amd/csrc "$@" -x hip -c pivotcvae_amd/csrc/$TU.hip -o $OBJ
OBJS=""
for o in error elementwise gemm_f32 catalog_f32 catalog_bf16 catalog_sparse candidate_ce catalog_sample catalog_api; do
  if [ $o = $TU ]; then OBJS="$OBJS $OBJ"; else OBJS="$OBJS pivotcvae_amd/lib/obj/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
echo built $OUT
