#!/bin/bash
# profile_round.sh <tag>: the evidence behind bench.py's roofline block, for one round (run on the GPU box).
#   1. bench.py (config 4, headline arithmetic = bf16x3, variants included)  -> gpurun_out/prof_<tag>/bench.json
#   2. the headline alone under rocprofv3 --kernel-trace --stats              -> x3_config4_kernel_stats.csv (+ the bench line)
#      and the other arithmetics / modes the same way                         -> f32_, bf16_, nneg_ _config4_kernel_stats.csv
#   3. separate --pmc passes (kernel-trace only) for HBM traffic and SQ        -> x3_config4_pmc_<set>.csv (per-kernel means)
#   4. kernel stats of config 3 and config 5 (their stated arithmetic: bf16)
TAG=${1:-r03}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $OUT/bench.log 2>&1 && grep '^{"metric"' $OUT/bench.log | tail -1 > $OUT/bench.json
echo "[1] plain bench done" | tee -a $OUT/progress.log
stats() {  # stats <name> <bench args...>
  local name=$1; shift
  rm -rf $OUT/trace_$name
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -- python3 $ROOT/bench.py "$@" > $OUT/trace_$name.log 2>&1
  grep '^{"metric"' $OUT/trace_$name.log | tail -1 > $OUT/${name}_bench_under_rocprof.json
  find $OUT/trace_$name -name "*kernel_stats.csv" -exec cp {} $OUT/${name}_kernel_stats.csv \;
  rm -rf $OUT/trace_$name
  echo "[stats] $name done" | tee -a $OUT/progress.log
}
stats x3_config4 --steps 5 --warmup 2 --no-variants --no-extras --no-cpu-baseline
stats f32_config4 --dtype f32 --steps 5 --warmup 2 --no-variants --no-extras --no-cpu-baseline
stats bf16_config4 --dtype bf16 --steps 5 --warmup 2 --no-variants --no-extras --no-cpu-baseline
stats nneg_config4 --n_neg 1000 --steps 5 --warmup 2 --no-variants --no-extras --no-cpu-baseline
pmc() {  # pmc <name> "<counters>"
  local name=$1
  rm -rf $OUT/pmc_$name
  timeout -k 10 600 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-variants > $OUT/pmc_$name.log 2>&1
  python3 $ROOT/tools/summarize_pmc.py $OUT/pmc_$name > $OUT/x3_config4_pmc_$name.csv
  rm -rf $OUT/pmc_$name
  echo "[pmc] $name done" | tee -a $OUT/progress.log
}
pmc FETCH_SIZE "FETCH_SIZE"
pmc WRITE_SIZE "WRITE_SIZE"
pmc SQ1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
pmc SQ2 "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT"
# round 3: memory-side counters of the SPARSE kernel (the reference's default n_neg = 1000 mode)
pmcn() {
  rm -rf $OUT/pmcn_$1
  timeout -k 10 600 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $OUT/pmcn_$1 -- python3 $ROOT/bench.py --n_neg 1000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-variants > $OUT/pmcn_$1.log 2>&1
  python3 $ROOT/tools/summarize_pmc.py $OUT/pmcn_$1 > $OUT/nneg_config4_pmc_$1.csv
  rm -rf $OUT/pmcn_$1
  echo "[pmc nneg] $1 done" | tee -a $OUT/progress.log
}
pmcn FETCH_SIZE
pmcn WRITE_SIZE
stats f32_config2 --config 2 --steps 20 --warmup 5 --no-variants --no-graph
stats f32_config1 --config 1 --steps 20 --warmup 5 --no-variants --no-graph --no-extras
stats bf16_config3 --config 3 --steps 5 --warmup 2 --no-variants
stats bf16_config5 --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-variants
stats x3_config5 --config 5 --dtype bf16x3 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-variants
stats x3_config3 --config 3 --dtype bf16x3 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-variants
# the 8-GPU run's per-rank load on one GPU, through a 1-rank RCCL group (the collective path of Trainer.step)
PCVAE_BENCH_FORCE_DIST=1 python3 $ROOT/bench.py --global-batch 1024 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-variants > $OUT/shard1024.log 2>&1
grep '^{"metric"' $OUT/shard1024.log | tail -1 > $OUT/x3_config4_B1024_rccl1_bench.json
# one traced EAGER step per config: the launch listing (tools/step_trace_list.py)
for spec in "2 1024 f32" "3 4096 bf16" "4 8192 bf16x3"; do
  set -- $spec
  rm -rf $OUT/t
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $ROOT/tools/step_trace_run.py $1 $2 $3 8 > $OUT/t$1.log 2>&1 &&
  python3 $ROOT/tools/step_trace_list.py $(find $OUT/t -name "*kernel_trace.csv") > $OUT/$3_config$1_step_launches.txt
  rm -rf $OUT/t
  echo "[trace] config $1 done" | tee -a $OUT/progress.log
done
bash $ROOT/tools/profile_gather.sh > $OUT/gather_profile.log 2>&1
cp $ROOT/gpurun_out/prof_gather/plain.txt $OUT/gather_kernel_timer.txt
cp $ROOT/gpurun_out/prof_gather/gather_kernel_stats.csv $ROOT/gpurun_out/prof_gather/gather_pmc_FETCH_SIZE.csv $ROOT/gpurun_out/prof_gather/gather_pmc_WRITE_SIZE.csv $OUT/
ls -la $OUT
