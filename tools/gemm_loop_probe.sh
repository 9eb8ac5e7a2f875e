#!/bin/bash
# gemm_loop_probe.sh: what the split-bf16 K loop of gemm_tile_dma is made of.  Probe builds of gemm_f32.hip without the operand
# split's vector work (-DGEMM_PROBE_NO_SPLIT), without the MFMAs (-DGEMM_PROBE_NO_MFMA) and without both (results are garbage: the
# loop's data movement - LDS-DMA fill, barrier, LDS reads - stays), each through tools/gemm_launch_floor.py (fixed cost + us per
# 32-deep chunk of a forward launch inside a replayed hipGraph).  -> gpurun_out/gemm_loop_probe.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
OUT=$ROOT/gpurun_out/gemm_loop_probe.txt
mkdir -p build/variants
bash tools/build_variant_tu.sh gemm_f32 build/variants/gemm_NO_SPLIT.so -DGEMM_PROBE_NO_SPLIT > /dev/null
bash tools/build_variant_tu.sh gemm_f32 build/variants/gemm_NO_MFMA.so -DGEMM_PROBE_NO_MFMA > /dev/null
bash tools/build_variant_tu.sh gemm_f32 build/variants/gemm_NO_BOTH.so -DGEMM_PROBE_NO_SPLIT -DGEMM_PROBE_NO_MFMA > /dev/null
echo "# [8192 x K] . [256 x K]^T forward launches in a replayed hipGraph, us per launch; fit = fixed + per 32-deep chunk" > $OUT
for v in "" build/variants/gemm_NO_SPLIT.so build/variants/gemm_NO_MFMA.so build/variants/gemm_NO_BOTH.so; do
  echo "== ${v:-product build}" >> $OUT
  PCVAE_LIB=$v timeout -k 10 120 python3 tools/gemm_launch_floor.py 2>/dev/null | tail -2 >> $OUT
done
cat $OUT
