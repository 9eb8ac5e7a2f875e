# evidence for the gather rooflines: kernel-attached events (plain run) vs rocprofv3's kernel trace, + FETCH / WRITE counters
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_gather; rm -rf $O; mkdir -p $O
python3 $GRAFT_REPO_ROOT/tools/gather_timer_run.py 20 > $O/plain.txt 2>$O/plain.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $GRAFT_REPO_ROOT/tools/gather_timer_run.py 20 > $O/under_rocprof.txt 2>&1
find $O/t -name "*kernel_stats.csv" -exec cp {} $O/gather_kernel_stats.csv \; ; rm -rf $O/t
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p -- python3 $GRAFT_REPO_ROOT/tools/gather_timer_run.py 6 > $O/pmc_$c.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/summarize_pmc.py $O/p | grep -i "kernel,\|gather_rows\|assemble" > $O/gather_pmc_$c.csv; rm -rf $O/p
done
cat $O/plain.txt; grep -i "gather_rows\|assemble\|Name" $O/gather_kernel_stats.csv | cut -c1-200; cat $O/gather_pmc_*.csv
