# one traced EAGER train step of configs 2 and 3 (and the kernel stats of the same runs): gpurun_out/tr3/
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/tr3; rm -rf $O; mkdir -p $O
for spec in "2 1024 f32" "3 4096 bf16" "4 8192 bf16x3"; do
  set -- $spec
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $GRAFT_REPO_ROOT/tools/step_trace_run.py $1 $2 $3 8 > $O/t$1.log 2>&1 &&
  python3 $GRAFT_REPO_ROOT/tools/step_trace_list.py $(find $O/t -name "*kernel_trace.csv") > $O/step$1_$3.txt &&
  find $O/t -name "*kernel_stats.csv" -exec cp {} $O/step$1_$3_kernel_stats.csv \; ;
  rm -rf $O/t
done
