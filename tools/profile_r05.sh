#!/bin/bash
# profile_r05.sh [part]: the evidence behind round 5's bench blocks (run on the GPU box; parts keep one gpurun call short).
#   part A: config 4 - the default line (headline + variants incl. candidates_nneg1000 / _nneg50 + pivot_rules); the reference's
#           DEFAULT training mode as its own line (--n_candidate 1000 / 50): kernel stats + FETCH_SIZE / WRITE_SIZE passes of
#           candidate_ce_kernel<128, true>; the sampled-rule step (pivotcvae_spt_pi) kernel stats
#   part E: config 3's kernel on the 3-range plan, four PMC passes
#   part D: SQ counters of the candidate kernel
#   part C: config 3 on the round-5 plan; candidate mode on bf16 rows at configs 3 / 5; config 3's full line
#   part B: the gather evidence on this tree (tools/profile_gather.sh); bf16 / bf16x3 FETCH / WRITE passes at config 4 (traffic.json's
#           two stale entries)
PART=${1:-A}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {  # stats <name> <bench args...>
  local name=$1; shift
  rm -rf $OUT/trace_$name
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -- python3 $ROOT/bench.py "$@" > $OUT/trace_$name.log 2>&1
  grep '^{"metric"' $OUT/trace_$name.log | tail -1 > $OUT/${name}_bench_under_rocprof.json
  find $OUT/trace_$name -name "*kernel_stats.csv" -exec cp {} $OUT/${name}_kernel_stats.csv \;
  rm -rf $OUT/trace_$name
  echo "[stats] $name done $(date +%T)" | tee -a $OUT/progress.log
}
pmc() {  # pmc <name> <set name> "<counters>" <bench args...>
  local name=$1 set=$2 ctr=$3; shift 3
  rm -rf $OUT/pmc_tmp
  timeout -k 10 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_tmp -- python3 $ROOT/bench.py "$@" > $OUT/pmc_${name}_$set.log 2>&1
  python3 $ROOT/tools/summarize_pmc.py $OUT/pmc_tmp > $OUT/${name}_pmc_$set.csv
  rm -rf $OUT/pmc_tmp
  echo "[pmc] $name $set done $(date +%T)" | tee -a $OUT/progress.log
}
LEAN="--no-cpu-baseline --no-extras --no-variants"
if [ "$PART" = "A" ]; then
  python3 $ROOT/bench.py > $OUT/bench.log 2>&1 && grep '^{"metric"' $OUT/bench.log | tail -1 > $OUT/config4_bench.json
  echo "[A] plain bench done $(date +%T)" | tee -a $OUT/progress.log
  for cn in 1000 50; do
    python3 $ROOT/bench.py --n_candidate $cn --steps 20 --warmup 5 $LEAN > $OUT/cand$cn.log 2>&1
    grep '^{"metric"' $OUT/cand$cn.log | tail -1 > $OUT/cand${cn}_config4_bench.json
    stats cand${cn}_config4 --n_candidate $cn --steps 10 --warmup 2 $LEAN
    pmc cand${cn}_config4 FETCH_SIZE "FETCH_SIZE" --n_candidate $cn --steps 3 --warmup 1 $LEAN
    pmc cand${cn}_config4 WRITE_SIZE "WRITE_SIZE" --n_candidate $cn --steps 3 --warmup 1 $LEAN
  done
  stats spt_config4 --model pivotcvae_spt_pi --steps 3 --warmup 1 $LEAN
elif [ "$PART" = "E" ]; then
  # config 3 on the 3-range plan: the four PMC passes of catalog_ce_bf16_pipe_kernel<64, 4>
  stats bf16_config3 --config 3 --steps 20 --warmup 5 --no-graph $LEAN
  pmc bf16_config3 SQ1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --config 3 --steps 3 --warmup 1 --no-graph $LEAN
  pmc bf16_config3 SQ2 "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT" --config 3 --steps 3 --warmup 1 --no-graph $LEAN
  pmc bf16_config3 FETCH_SIZE "FETCH_SIZE" --config 3 --steps 3 --warmup 1 --no-graph $LEAN
  pmc bf16_config3 WRITE_SIZE "WRITE_SIZE" --config 3 --steps 3 --warmup 1 --no-graph $LEAN
elif [ "$PART" = "D" ]; then
  # what the candidate kernel's waves do: SQ counters (issuing / issue-stalled / parked at waits), two passes
  pmc cand1000_config4 SQ1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --n_candidate 1000 --steps 3 --warmup 1 --no-graph $LEAN
  pmc cand1000_config4 SQ2 "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU" --n_candidate 1000 --steps 3 --warmup 1 --no-graph $LEAN
elif [ "$PART" = "C" ]; then
  # config 3 on the round-5 plan (3 ranges), candidate mode on bf16 rows at configs 3 / 5 (their stated arithmetic)
  stats bf16_config3 --config 3 --steps 20 --warmup 5 --no-graph $LEAN
  for c in 3 5; do
    python3 $ROOT/bench.py --config $c --n_candidate 1000 --steps 10 --warmup 3 $LEAN > $OUT/cand1000_config$c.log 2>&1
    grep '^{"metric"' $OUT/cand1000_config$c.log | tail -1 > $OUT/cand1000_config${c}_bench.json
    stats cand1000_config$c --config $c --n_candidate 1000 --steps 5 --warmup 2 $LEAN
  done
  python3 $ROOT/bench.py --config 3 --steps 20 --warmup 5 > $OUT/bench3.log 2>&1
  grep '^{"metric"' $OUT/bench3.log | tail -1 > $OUT/config3_bench.json
elif [ "$PART" = "B" ]; then
  bash $ROOT/tools/profile_gather.sh > $OUT/gather.log 2>&1
  cp $ROOT/gpurun_out/prof_gather/gather_kernel_stats.csv $OUT/gather_kernel_stats.csv
  cp $ROOT/gpurun_out/prof_gather/gather_pmc_FETCH_SIZE.csv $ROOT/gpurun_out/prof_gather/gather_pmc_WRITE_SIZE.csv $OUT/
  cp $ROOT/gpurun_out/prof_gather/plain.txt $OUT/gather_timer_plain.txt
  cp $ROOT/gpurun_out/prof_gather/under_rocprof.txt $OUT/gather_timer_under_rocprof.txt
  echo "[B] gather done $(date +%T)" | tee -a $OUT/progress.log
  for dt in bf16 bf16x3; do
    stats ${dt}_config4 --dtype $dt --steps 3 --warmup 1 $LEAN
    pmc ${dt}_config4 FETCH_SIZE "FETCH_SIZE" --dtype $dt --steps 2 --warmup 1 $LEAN
    pmc ${dt}_config4 WRITE_SIZE "WRITE_SIZE" --dtype $dt --steps 2 --warmup 1 $LEAN
  done
fi
ls -la $OUT | tail -40
