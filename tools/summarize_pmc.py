#!/usr/bin/env python3
"""Per-kernel means of the counters of one rocprofv3 --pmc pass:  summarize_pmc.py <rocprof output dir>  -> CSV on stdout"""
import collections, csv, glob, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
for (k, c), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([k, c, len(v), round(sum(v) / len(v), 1)])
