# one traced EAGER train step (kernel durations) of a small config: tools/trace_step_small.sh <config> <B> <arith>  -> gpurun_out/tr_small/
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/tr_small; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $GRAFT_REPO_ROOT/tools/step_trace_run.py $1 $2 $3 8 > $O/log.txt 2>&1 &&
python3 $GRAFT_REPO_ROOT/tools/step_trace_list.py $(find $O/t -name "*kernel_trace.csv") > $O/step$1_$3.txt
rm -rf $O/t
