#!/bin/bash
# nsplit_sweep.sh [config] [dtype]: the catalog kernel's time as a function of the number of catalog ranges per row block
# (PCVAE_PLAN_NSPLIT, experiments only) - the measurement behind catalog_plan's per-round overhead term (PLAN_ROUND_TILES).
CFG=${1:-3}; DT=${2:-bf16}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/nsplit_sweep_config${CFG}_${DT}.txt
: > $OUT
for ns in default 1 2 3 4 5 6 8 10 12 16 24; do
  if [ "$ns" = "default" ]; then unset PCVAE_PLAN_NSPLIT; else export PCVAE_PLAN_NSPLIT=$ns; fi
  for rep in 1 2; do
    timeout -k 10 200 python3 $ROOT/bench.py --config $CFG --dtype $DT --steps 20 --warmup 5 --no-graph --no-cpu-baseline --no-extras --no-variants 2>/dev/null | \
      python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('nsplit=$ns rep=$rep kernel_ms=%.4f frac=%.4f step_ms=%.4f' % (r['ms_per_launch'], r['frac'], d['ms_per_step']))" | tee -a $OUT
  done
done
