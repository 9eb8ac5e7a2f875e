#!/bin/bash
# profile_r06.sh [part]: the evidence behind round 6's numbers (run on the GPU box; parts keep one gpurun call short).
#   part A: the default line as the driver runs it (--steps 20 --warmup 5) + its extras file; one rank's shard of the 8-GPU run
#           (B = 1024, graph replay, a REAL 1-rank RCCL all-reduce: the line's `dist` block) in full-softmax and candidate mode
#   part H: the headline kernel on this tree: kernel stats + four PMC passes (SQ1, SQ2, FETCH_SIZE, WRITE_SIZE)
#   part C: candidate mode (Cn = 1000 / 50) lines + kernel stats per MLP arithmetic (what the MLP GEMMs cost in the light steps)
#   part G: config 3: the full line (generate after the prefix-screening change) + the generate chain's kernel stats
#   part M: the launches of ONE candidate-mode step (Cn = 50) in order, per MLP arithmetic: which GEMM launch costs what
#   part O: the other BASELINE configs on this tree: config 1 / 2 lines (as specified), config 5 (N = 10M, K = 20, D = 256, bf16) line and
#           its candidate mode (bf16 rows: the stated arithmetic)
#   part S: the train step's gather kernel: alignment probe + same-box reference points + write-request counters
PART=${1:-A}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {  # stats <name> <program> <args...>
  local name=$1; shift
  rm -rf $OUT/trace_$name
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -- python3 "$@" > $OUT/trace_$name.log 2>&1
  grep '^{"metric"' $OUT/trace_$name.log | tail -1 > $OUT/${name}_bench_under_rocprof.json
  [ -s $OUT/${name}_bench_under_rocprof.json ] || rm -f $OUT/${name}_bench_under_rocprof.json
  find $OUT/trace_$name -name "*kernel_stats.csv" -exec cp {} $OUT/${name}_kernel_stats.csv \;
  rm -rf $OUT/trace_$name
  echo "[stats] $name done $(date +%T)" | tee -a $OUT/progress.log
}
pmc() {  # pmc <name> <set name> "<counters>" <program> <args...>
  local name=$1 set=$2 ctr=$3; shift 3
  rm -rf $OUT/pmc_tmp
  timeout -k 10 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_tmp -- python3 "$@" > $OUT/pmc_${name}_$set.log 2>&1
  python3 $ROOT/tools/summarize_pmc.py $OUT/pmc_tmp > $OUT/${name}_pmc_$set.csv
  rm -rf $OUT/pmc_tmp
  echo "[pmc] $name $set done $(date +%T)" | tee -a $OUT/progress.log
}
line() {  # line <name> <bench args...>: one bench run, its line and its extras file
  local name=$1; shift
  PCVAE_BENCH_EXTRAS=$OUT/${name}_bench_extras.json python3 $ROOT/bench.py "$@" > $OUT/$name.log 2> $OUT/$name.err
  tail -1 $OUT/$name.log > $OUT/${name}_bench.json
  echo "[line] $name: $(wc -c < $OUT/${name}_bench.json) bytes $(date +%T)" | tee -a $OUT/progress.log
}
B=$ROOT/bench.py
LEAN="--no-cpu-baseline --no-extras --no-variants"
if [ "$PART" = "A" ]; then
  line config4 --steps 20 --warmup 5
  PCVAE_BENCH_FORCE_DIST=1 line x6_config4_B1024_rccl1 --global-batch 1024 --steps 50 --warmup 10 $LEAN
  PCVAE_BENCH_FORCE_DIST=1 line cand1000_config4_B1024_rccl1 --global-batch 1024 --n_candidate 1000 --steps 200 --warmup 20 $LEAN
elif [ "$PART" = "H" ]; then
  stats x6_config4 $B --steps 6 --warmup 2 $LEAN
  pmc x6_config4 SQ1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" $B --steps 2 --warmup 1 $LEAN
  pmc x6_config4 SQ2 "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT" $B --steps 2 --warmup 1 $LEAN
  pmc x6_config4 FETCH_SIZE "FETCH_SIZE" $B --steps 2 --warmup 1 $LEAN
  pmc x6_config4 WRITE_SIZE "WRITE_SIZE" $B --steps 2 --warmup 1 $LEAN
elif [ "$PART" = "C" ]; then
  line cand1000_config4 --n_candidate 1000 --steps 50 --warmup 10 --no-extras --no-variants
  line cand50_config4 --n_candidate 50 --steps 200 --warmup 20 --no-extras --no-variants
  for mlp in f32 bf16x3 bf16x6; do
    stats cand50_config4_mlp_$mlp $B --n_candidate 50 --mlp $mlp --steps 20 --warmup 5 --no-graph $LEAN
  done
  stats cand1000_config4 $B --n_candidate 1000 --steps 10 --warmup 2 $LEAN
elif [ "$PART" = "G" ]; then
  line config3 --config 3 --steps 20 --warmup 5
  stats config3_generate $ROOT/tools/gen_trace_run.py 3
elif [ "$PART" = "O" ]; then
  line config1 --config 1 --steps 50 --warmup 10
  line config2 --config 2 --steps 50 --warmup 10
  line config5 --config 5 --steps 5 --warmup 2
  line cand1000_config5 --config 5 --n_candidate 1000 --steps 20 --warmup 5 --no-extras --no-variants
  line cand1000_config3 --config 3 --n_candidate 1000 --steps 50 --warmup 10 --no-extras --no-variants
elif [ "$PART" = "M" ]; then
  for mlp in f32 bf16x3 bf16x6; do
    rm -rf $OUT/tr_m
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_m -- python3 $ROOT/tools/step_trace_run.py 4 8192 bf16x6 8 $mlp 50 > $OUT/tr_m_$mlp.log 2>&1 &&
      python3 $ROOT/tools/step_trace_list.py $(find $OUT/tr_m -name "*kernel_trace.csv") > $OUT/cand50_config4_mlp_${mlp}_step_launches.txt
    rm -rf $OUT/tr_m
    echo "[M] $mlp done $(date +%T)" | tee -a $OUT/progress.log
  done
elif [ "$PART" = "S" ]; then
  python3 $ROOT/tools/assemble_align_probe.py > $OUT/assemble_row_align_probe.txt 2>&1
  (rocprofv3-avail list 2>/dev/null || rocprofv3 --list-avail 2>/dev/null) | grep -io "TCC_[A-Z0-9_]*WR[A-Z0-9_]*\|TCC_[A-Z0-9_]*STALL[A-Z0-9_]*" | sort -u > $OUT/avail_tcc_write_counters.txt
  pmc assemble WR "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" $ROOT/tools/assemble_align_probe.py
  pmc assemble WR2 "TCC_EA0_WR_UNCACHED_32B_sum TCC_REQ_sum TCC_WRITE_sum" $ROOT/tools/assemble_align_probe.py
  pmc assemble SZ "FETCH_SIZE" $ROOT/tools/assemble_align_probe.py
  pmc assemble SZW "WRITE_SIZE" $ROOT/tools/assemble_align_probe.py
fi
ls -la $OUT | tail -40
