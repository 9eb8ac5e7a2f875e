#!/bin/bash
# range_sweep.sh: config 5 (bf16 and bf16x3) at several limits on the table bytes a workgroup streams per catalog range
# (PCVAE_RANGE_MB): launch time of the catalog kernel + FETCH_SIZE (L2 memory-side reads).  Do the 32 workgroups of an XCD keep
# sharing their stream in L2 when a range is shorter?
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/range_sweep
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "bf16 0" "bf16 1024" "bf16 256" "bf16 96" "bf16x3 0" "bf16x3 256" "bf16x3 160" ; do
  set -- $spec
  export PCVAE_RANGE_MB=$2
  rm -rf $OUT/pmc_tmp
  timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_tmp -- python3 $ROOT/bench.py --config 5 --dtype $1 --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-variants > $OUT/run_$1_$2.log 2>&1
  python3 $ROOT/tools/summarize_pmc.py $OUT/pmc_tmp | grep "pipe_kernel" > $OUT/fetch_$1_$2.csv
  grep '^{"metric"' $OUT/run_$1_$2.log | tail -1 | python3 -c "import json,sys; d=json.load(sys.stdin); print('$1 range_mb=$2', 'ms_per_launch', round(d['roofline']['ms_per_launch'],1), 'frac', round(d['roofline']['frac'],3), 'rec', d['elbo']['recLoss'])" | tee -a $OUT/summary.txt
  cat $OUT/fetch_$1_$2.csv | tee -a $OUT/summary.txt
  rm -rf $OUT/pmc_tmp
done
