#!/bin/bash
# build_variant.sh <catalog_bf16 source> <out.so> [extra hipcc flags]: link a variant library that differs only in the bf16 catalog TU
set -e
SRC=$1; OUT=$2; shift 2
cd "$(dirname "$0")/.."
mkdir -p build/variants
OBJ=build/variants/$(basename $OUT .so).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I pivotcvae_amd/csrc "$@" -x hip -c $SRC -o $OBJ
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT pivotcvae_amd/lib/obj/error.o pivotcvae_amd/lib/obj/elementwise.o pivotcvae_amd/lib/obj/gemm_f32.o pivotcvae_amd/lib/obj/catalog_f32.o pivotcvae_amd/lib/obj/catalog_sparse.o pivotcvae_amd/lib/obj/catalog_api.o $OBJ
echo built $OUT
