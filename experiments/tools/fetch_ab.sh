#!/bin/bash
# fetch_ab.sh <dtype> <lib.so>...: memory-side reads (FETCH_SIZE, one rocprofv3 --pmc pass each) of the catalog CE kernel at config 4's shape
# for several builds of the library, on one box (CTRS="<counters of one pass>" overrides FETCH_SIZE).  Output: gpurun_out/fetch_ab.txt
DT=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  rm -rf /tmp/fetch_ab_tmp
  export PCVAE_LIB=$ROOT/$L
  timeout -k 10 300 rocprofv3 --pmc ${CTRS:-FETCH_SIZE} --kernel-trace --output-format csv -d /tmp/fetch_ab_tmp -- python3 $ROOT/tools/bench_catalog.py --dtype $DT --iters 3 > /tmp/fetch_ab.log 2>&1
  echo "## $L [${CTRS:-FETCH_SIZE}] $(grep '^{' /tmp/fetch_ab.log | tail -1)" >> $OUT/fetch_ab.txt
  python3 $ROOT/tools/summarize_pmc.py /tmp/fetch_ab_tmp | grep -i "pipe_kernel\|f32_kernel" >> $OUT/fetch_ab.txt
done
cat $OUT/fetch_ab.txt
