#!/bin/bash
# profile_r04.sh [part]: the evidence behind bench.py's roofline blocks, round 4 (run on the GPU box; parts keep one gpurun call short).
#   part A: config 4 - the default line (headline bf16x6 + variants), kernel stats of the four arithmetics, PMC passes of bf16x6
#   part B: PMC passes (SQ1, SQ2, FETCH_SIZE, WRITE_SIZE) + kernel stats of the kernels at the widths that are not 128:
#           config 3 bf16 (catalog_ce_bf16_pipe_kernel<64, 4>), config 5 bf16 (<256, 2>), config 5 bf16x3 (x3_pipe_kernel<256, 1, 2>)
#   part C: config 5 end to end WITH extras (generate, eval) on one GPU
PART=${1:-A}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {  # stats <name> <bench args...>
  local name=$1; shift
  rm -rf $OUT/trace_$name
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -- python3 $ROOT/bench.py "$@" > $OUT/trace_$name.log 2>&1
  grep '^{"metric"' $OUT/trace_$name.log | tail -1 > $OUT/${name}_bench_under_rocprof.json
  find $OUT/trace_$name -name "*kernel_stats.csv" -exec cp {} $OUT/${name}_kernel_stats.csv \;
  rm -rf $OUT/trace_$name
  echo "[stats] $name done $(date +%T)" | tee -a $OUT/progress.log
}
pmc() {  # pmc <name> <set name> "<counters>" <bench args...>
  local name=$1 set=$2 ctr=$3; shift 3
  rm -rf $OUT/pmc_tmp
  timeout -k 10 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_tmp -- python3 $ROOT/bench.py "$@" > $OUT/pmc_${name}_$set.log 2>&1
  python3 $ROOT/tools/summarize_pmc.py $OUT/pmc_tmp > $OUT/${name}_pmc_$set.csv
  rm -rf $OUT/pmc_tmp
  echo "[pmc] $name $set done $(date +%T)" | tee -a $OUT/progress.log
}
pmc4() {  # the four separate passes for one workload
  local name=$1; shift
  pmc $name SQ1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "$@"
  pmc $name SQ2 "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT" "$@"
  pmc $name FETCH_SIZE "FETCH_SIZE" "$@"
  pmc $name WRITE_SIZE "WRITE_SIZE" "$@"
}
LEAN="--no-cpu-baseline --no-extras --no-variants"
if [ "$PART" = "A" ]; then
  python3 $ROOT/bench.py > $OUT/bench.log 2>&1 && grep '^{"metric"' $OUT/bench.log | tail -1 > $OUT/config4_bench.json
  echo "[A] plain bench done $(date +%T)" | tee -a $OUT/progress.log
  stats x6_config4 --steps 5 --warmup 2 $LEAN
  stats f32_config4 --dtype f32 --steps 3 --warmup 1 $LEAN
  stats x3_config4 --dtype bf16x3 --steps 5 --warmup 2 $LEAN
  stats bf16_config4 --dtype bf16 --steps 5 --warmup 2 $LEAN
  pmc4 x6_config4 --steps 2 --warmup 1 $LEAN
  PCVAE_BENCH_FORCE_DIST=1 python3 $ROOT/bench.py --global-batch 1024 --steps 20 --warmup 5 $LEAN > $OUT/shard1024.log 2>&1
  grep '^{"metric"' $OUT/shard1024.log | tail -1 > $OUT/x6_config4_B1024_rccl1_bench.json
  echo "[A] shard done $(date +%T)" | tee -a $OUT/progress.log
elif [ "$PART" = "B" ]; then
  stats bf16_config3 --config 3 --steps 5 --warmup 2 --no-variants --no-cpu-baseline --no-extras
  pmc4 bf16_config3 --config 3 --steps 3 --warmup 1 --no-graph $LEAN
  stats bf16_config5 --config 5 --steps 2 --warmup 1 $LEAN
  pmc4 bf16_config5 --config 5 --steps 2 --warmup 1 $LEAN
  stats x3_config5 --config 5 --dtype bf16x3 --steps 2 --warmup 1 $LEAN
  pmc4 x3_config5 --config 5 --dtype bf16x3 --steps 1 --warmup 1 $LEAN
elif [ "$PART" = "C" ]; then
  python3 $ROOT/bench.py --config 5 --steps 2 --warmup 1 --no-variants > $OUT/bench5.log 2>&1
  grep '^{"metric"' $OUT/bench5.log | tail -1 > $OUT/config5_bench.json
  echo "[C] config 5 with extras done $(date +%T)" | tee -a $OUT/progress.log
fi
ls -la $OUT | tail -40
