#!/usr/bin/env python3
"""step_trace_list.py <kernel_trace.csv> [occurrence]: the launches of ONE train step (from one adam_kernel launch to the next)
in order: duration, idle gap before it, grid, workgroup, name."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) - 2
a, b = idx[k], idx[k + 1]
prev = int(rows[a]['End_Timestamp'])
busy = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').replace('at::native::', '')
    g = "x".join(str(int(r[f'Grid_Size_{d}']) // max(1, int(r[f'Workgroup_Size_{d}']))) for d in 'XYZ')
    print(f"{(e - s) / 1e3:8.1f} us  gap {(s - prev) / 1e3:6.1f}  wgs {g:>12s}  {name[:100]}")
    busy += e - s
    prev = e
print(f"launches {b - a}, busy {busy / 1e3:.1f} us, span {(prev - int(rows[a]['End_Timestamp'])) / 1e3:.1f} us")
