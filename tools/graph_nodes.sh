# kernels of ONE hipGraph-replayed step (adam to adam) of a mid-size pt_pi / gt_pi model -> gpurun_out/gn/step_<variant>.txt : any
# __amd_rocclr_fillBuffer* / copyBuffer* in the listing is a memset / memcpy NODE of the captured graph (see pcvae_zero)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/gn; rm -rf $O; mkdir -p $O
for v in pivotcvae_pt_pi pivotcvae_gt_pi; do
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t_$v -- python3 $GRAFT_REPO_ROOT/tools/graph_nodes_run.py $v > $O/log_$v.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/step_trace_list.py $(find $O/t_$v -name "*kernel_trace.csv") 4 > $O/step_$v.txt
rm -rf $O/t_$v
done
