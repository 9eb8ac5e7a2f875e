#!/usr/bin/env python3
"""count_launches.py <kernel_stats.csv> <steps>: launches per train step from a rocprofv3 --kernel-trace --stats summary"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
tot = 0
for r in rows:
    c = int(r["Calls"])
    tot += c
    if c >= steps:
        print("%6.2f/step %9.1f us  %s" % (c / steps, float(r["AverageNs"]) / 1e3, r["Name"][:100]))
print("total launches", tot, "per step ~ %.1f" % (tot / steps))
