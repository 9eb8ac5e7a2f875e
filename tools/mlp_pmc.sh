#!/bin/bash
# mlp_pmc.sh: rocprofv3 counter passes over the MLP GEMM kernel (tools/mlp_pmc_run.py: enc_1 forward per arithmetic), one small counter set per
# pass -> gpurun_out/mlp_pmc/*.csv + a summary (tools/mlp_pmc_summary.py): MFMA-pipe utilisation, vector / LDS issue cycles, instruction counts.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/mlp_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/avail_sq_counters.txt
rm -rf $OUT/trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/mlp_pmc_run.py > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/trace
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" "SQ_BUSY_CYCLES SQ_INSTS_SALU" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  i=$((i + 1))
  rm -rf $OUT/pmc_tmp
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_tmp -- python3 $ROOT/tools/mlp_pmc_run.py > $OUT/pass$i.log 2>&1
  echo "pass $i ($set): rc $?" | tee -a $OUT/progress.log
  python3 $ROOT/tools/summarize_pmc.py $OUT/pmc_tmp | grep "kernel,counter\|gemm_group" > $OUT/pass$i.csv
  rm -rf $OUT/pmc_tmp
done
cat $OUT/pass*.csv | grep -v "^kernel,counter" > $OUT/all_counters.csv
python3 $ROOT/tools/mlp_pmc_summary.py $OUT/all_counters.csv $OUT/kernel_stats.csv | tee $OUT/summary.txt
