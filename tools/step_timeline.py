#!/usr/bin/env python3
"""step_timeline.py <kernel_trace.csv> [occurrence]: kernels of ONE train step (between two launches of the fused
catalog kernel) from a rocprofv3 --kernel-trace csv, grouped by kernel name, with the GPU idle gaps."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'fast_kernel' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[k], idx[k + 1]
t0 = prev = int(rows[a]['End_Timestamp'])
by = collections.OrderedDict()
gap = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:60]
    c = by.setdefault(name, [0, 0.0])
    c[0] += 1
    c[1] += (e - s) / 1e3
    gap += max(0, s - prev)
    prev = e
for n, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"{d:9.1f} us  x{c:<3d} {n}")
print(f"step span {(prev - t0) / 1e3:.1f} us, idle gaps {gap / 1e3:.1f} us, launches {b - a}")
