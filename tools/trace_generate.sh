# kernel stats of a few greedy generate batches at config 4 -> gpurun_out/gen/
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/gen; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $GRAFT_REPO_ROOT/tools/gen_trace_run.py 4 > $O/log.txt 2>&1
find $O/t -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/t
