cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/tr1; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t4 -- python3 $GRAFT_REPO_ROOT/tools/step_trace_run.py 4 8192 bf16 6 > $O/t4.log 2>&1 &&
python3 $GRAFT_REPO_ROOT/tools/step_trace_list.py $(find $O/t4 -name "*kernel_trace.csv") > $O/step4.txt &&
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t3 -- python3 $GRAFT_REPO_ROOT/tools/step_trace_run.py 3 4096 bf16 6 > $O/t3.log 2>&1 &&
python3 $GRAFT_REPO_ROOT/tools/step_trace_list.py $(find $O/t3 -name "*kernel_trace.csv") > $O/step3.txt &&
rm -rf $O/t4 $O/t3; tail -n 3 $O/step4.txt $O/step3.txt
