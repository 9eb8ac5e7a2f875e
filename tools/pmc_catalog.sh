#!/bin/bash
# pmc_catalog.sh <tag> <counter set ...>   (one rocprofv3 --pmc pass per quoted set; kernel-trace only, as gpurun requires)
# Collects counters for the stand-alone catalog CE call (tools/bench_catalog.py) into gpurun_out/pmc_<tag>_<i>/ and
# prints per-kernel means.
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "$@"; do
  OUT=$ROOT/gpurun_out/pmc_${TAG}_$i
  rm -rf $OUT; mkdir -p $OUT
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/bench_catalog.py --iters 2 ${BENCH_ARGS} > $OUT/log.txt 2>&1
  python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    if 'catalog' in k:
        print(k, {c: round(sum(x) / len(x), 1) for c, x in v.items()}, 'n=', len(next(iter(v.values()))))
PY
  i=$((i+1))
done
