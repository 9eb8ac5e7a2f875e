# configs 1-3 on the final tree: kernel stats (eager), step listings, and the hipGraph-replay bench lines -> gpurun_out/prof_small/
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/prof_small; rm -rf $OUT; mkdir -p $OUT
stats() {
  local name=$1; shift
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -- python3 $ROOT/bench.py "$@" > $OUT/trace_$name.log 2>&1
  grep '^{"metric"' $OUT/trace_$name.log | tail -1 > $OUT/${name}_bench_under_rocprof.json
  find $OUT/trace_$name -name "*kernel_stats.csv" -exec cp {} $OUT/${name}_kernel_stats.csv \;
  rm -rf $OUT/trace_$name
}
stats f32_config2 --config 2 --steps 20 --warmup 5 --no-variants --no-graph
stats f32_config1 --config 1 --steps 20 --warmup 5 --no-variants --no-graph --no-extras
for spec in "2 1024 f32" "3 4096 bf16"; do
  set -- $spec
  rm -rf $OUT/t
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $ROOT/tools/step_trace_run.py $1 $2 $3 8 > $OUT/t$1.log 2>&1 &&
  python3 $ROOT/tools/step_trace_list.py $(find $OUT/t -name "*kernel_trace.csv") > $OUT/$3_config$1_step_launches.txt
  rm -rf $OUT/t
done
# hipGraph-replay step times of the small configs (the numbers the tables quote)
for c in 1 2 3; do python3 $ROOT/bench.py --config $c --steps 200 --warmup 20 --no-variants --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/graph_config$c.json; done
ls $OUT
