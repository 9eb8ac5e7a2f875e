#!/bin/bash
# screen_prefix_sweep.sh [config]: greedy generation (recommend(return_item=True): bf16 screening + exact fp32 rescoring) against the
# size of the catalog prefix that seeds the screening threshold (PCVAE_SCREEN_PREFIX_DIV: the prefix is N / div items; 1 = round 5's
# behaviour below 262144 items, pass A over the whole catalog).  -> gpurun_out/screen_prefix_sweep.txt
CFG=${1:-3}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
echo "# config $CFG: generate (screened argmax) per prefix divisor; ids_identical = the same ids as the exact f32 kernel" > $OUT/screen_prefix_sweep.txt
for div in 1 2 3 4 6 8 16; do
  if [ "$div" = "1" ]; then export PCVAE_SCREEN_PREFIX_MIN_ITEMS=262144; else unset PCVAE_SCREEN_PREFIX_MIN_ITEMS; fi
  PCVAE_SCREEN_PREFIX_DIV=$div PCVAE_BENCH_EXTRAS=$OUT/sweep_extras.json python3 $ROOT/bench.py --config $CFG --steps 3 --warmup 2 --no-variants --no-cpu-baseline > $OUT/sweep.log 2>&1
  python3 - "$div" "$OUT/sweep_extras.json" >> $OUT/screen_prefix_sweep.txt <<'PY'
import json, sys
g = json.load(open(sys.argv[2]))["generate"]
print(f"prefix N / {sys.argv[1]:>2s}: {g['ms_per_batch']:.3f} ms per batch  {g['value'] / 1e6:.3f} M slates/s  frac of bf16 peak {g['frac']:.3f}  "
      f"ids_identical_to_f32_kernel {g['ids_identical_to_f32_kernel']}")
PY
done
cat $OUT/screen_prefix_sweep.txt
