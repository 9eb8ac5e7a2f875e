#!/usr/bin/env python3
"""step_sequence.py <kernel_trace.csv> [occurrence]: the kernels of ONE train step in launch order (name, microseconds),
between two launches of the fused catalog kernel, from a rocprofv3 --kernel-trace csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'catalog_ce_bf16_fast_kernel' in r['Kernel_Name'] or 'catalog_ce_f32_kernel' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
for r in rows[idx[k] + 1:idx[k + 1] + 1]:
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').replace('at::native::', '')
    print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  {name[:110]}")
