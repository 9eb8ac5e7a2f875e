#!/bin/bash
# screen_loop_probe.sh: what a steady-state slot of the pipelined screening kernel (config 3: D = 64, pass B over 10^5 items) is made of.
# Probe builds of catalog_bf16.hip (ids are garbage, every id stays a valid row): no item ever passes (-DSCREEN_PROBE_NO_CAND), and on top
# of that the seam without wait + barrier / without barrier, the steps without their LDS waits, the slot without its branch; each through
# tools/gen3_trace.sh (rocprofv3 kernel trace of eight batches of recommend()).  Beside it PCVAE_PLAN_NSPLIT = 4 / 16 on the product
# build: kernel time = rounds x (fixed + slots x per-slot) separates a workgroup's fixed cost from its slots.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
OUT=$ROOT/gpurun_out/screen_loop_probe.txt
mkdir -p build/variants gpurun_out
B="bash tools/build_variant_tu.sh catalog_bf16"
$B build/variants/screen_NO_CAND.so -DSCREEN_PROBE_NO_CAND > /dev/null
$B build/variants/screen_NO_SEAMWAIT.so -DSCREEN_PROBE_NO_CAND -DSCREEN_PROBE_NO_SEAMWAIT > /dev/null
$B build/variants/screen_NO_BARRIER.so -DSCREEN_PROBE_NO_CAND -DSCREEN_PROBE_NO_BARRIER > /dev/null
$B build/variants/screen_NO_LDSWAIT.so -DSCREEN_PROBE_NO_CAND -DSCREEN_PROBE_NO_LDSWAIT > /dev/null
$B build/variants/screen_NO_BRANCH.so -DSCREEN_PROBE_NO_CAND -DSCREEN_PROBE_NO_BRANCH > /dev/null
echo "# catalog_screen_pipe_kernel<64, 4, 1> (pass B) / <64, 4, 0> (pass A over N / 4), config 3, us per launch (max = the slate pass, R = 40 960)" > $OUT
echo "== product" >> $OUT; bash tools/gen3_trace.sh | head -2 >> $OUT
for ns in 4 16; do echo "== product, PCVAE_PLAN_NSPLIT=$ns" >> $OUT; PCVAE_PLAN_NSPLIT=$ns bash tools/gen3_trace.sh | head -2 >> $OUT; done
for v in NO_CAND NO_SEAMWAIT NO_BARRIER NO_LDSWAIT NO_BRANCH; do
  echo "== $v" >> $OUT; PCVAE_LIB=$ROOT/build/variants/screen_$v.so bash tools/gen3_trace.sh | head -2 >> $OUT
done
cat $OUT
