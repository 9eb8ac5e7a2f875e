#!/bin/bash
# screen_pmc.sh [config]: rocprofv3 kernel trace + counter passes over greedy generation (tools/screen_pmc_run.py), the screening kernels
# only -> gpurun_out/screen_pmc/{kernel_stats.csv, all_counters.csv}
CFG=${1:-3}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/screen_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/screen_pmc_run.py $CFG > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/trace
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_WAIT_INST_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_SALU SQ_INSTS_SMEM"; do
  i=$((i + 1))
  rm -rf $OUT/pmc_tmp
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_tmp -- python3 $ROOT/tools/screen_pmc_run.py $CFG > $OUT/pass$i.log 2>&1
  echo "pass $i ($set): rc $?" | tee -a $OUT/progress.log
  python3 $ROOT/tools/summarize_pmc.py $OUT/pmc_tmp | grep "kernel,counter\|catalog_screen" > $OUT/pass$i.csv
  rm -rf $OUT/pmc_tmp
done
cat $OUT/pass*.csv | grep -v "^kernel,counter" > $OUT/all_counters.csv
grep "catalog_screen" $OUT/kernel_stats.csv | cut -c1-200
cat $OUT/all_counters.csv
