#!/bin/bash
# gen3_trace.sh [config]: rocprofv3 kernel trace of eight batches of greedy generation (tools/screen_pmc_run.py) - the screening kernels' averages;
# PCVAE_LIB selects a variant library (tools/screen_loop_probe.sh)
CFG=${1:-3}
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/gen3_trace
timeout -k 10 90 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gen3_trace -- python3 $R/tools/screen_pmc_run.py $CFG > $R/gpurun_out/gen3_trace.log 2>&1
find $R/gpurun_out/gen3_trace -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/gen3_kernel_stats.csv \;
rm -rf $R/gpurun_out/gen3_trace
grep "catalog_screen" $R/gpurun_out/gen3_kernel_stats.csv | awk -F'",' '{n=split($1,a,"::"); split($2,b,","); printf "%-70s calls %s avg %.1f us min %.1f max %.1f\n", substr($1,1,90), b[1], b[3]/1e3, b[5]/1e3, b[6]/1e3}'
