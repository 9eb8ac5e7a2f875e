cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/tr2; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t4 -- python3 $GRAFT_REPO_ROOT/tools/step_trace_run.py 4 8192 bf16x3 5 > $O/t4.log 2>&1 &&
python3 $GRAFT_REPO_ROOT/tools/step_trace_list.py $(find $O/t4 -name "*kernel_trace.csv") > $O/step4_x3.txt &&
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t1 -- python3 $GRAFT_REPO_ROOT/tools/step_trace_run.py 4 1024 bf16x3 6 > $O/t1.log 2>&1 &&
python3 $GRAFT_REPO_ROOT/tools/step_trace_list.py $(find $O/t1 -name "*kernel_trace.csv") > $O/step4_x3_b1024.txt &&
rm -rf $O/t4 $O/t1
