#!/usr/bin/env python3
"""mlp_pmc_summary.py <all_counters.csv> <kernel_stats.csv>: per arithmetic of the MLP GEMM kernel (template argument 0 f32, 1 bf16x3, 2 bf16x6)
what the counter passes of tools/mlp_pmc.sh say.  Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs."""
import csv
import re
import sys

ctr = {}
for row in csv.reader(open(sys.argv[1])):
    m = re.search(r"gemm_group_kernel<\w+, \w+, (\d)>", row[0])
    if m:
        ctr.setdefault(int(m.group(1)), {})[row[1]] = float(row[3])
avg = {}
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        m = re.search(r"gemm_group_kernel<\w+, \w+, (\d)>", r["Name"])
        if m:
            avg[int(m.group(1))] = float(r["AverageNs"]) / 1e3
M, N, K = 8192, 256, 1419
for mode, name, mult, peak in ((0, "f32", 1, 157.3e12), (1, "bf16x3", 3, 2500e12), (2, "bf16x6", 6, 2500e12)):
    c = ctr.get(mode)
    if not c:
        continue
    print(f"== {name}: gemm_group_kernel<false, false, {mode}>, [8192 x 1419] . [256 x 1419]^T forward")
    if mode in avg:
        us = avg[mode]
        print(f"   kernel trace: {us:.1f} us per launch = {2.0 * M * N * K / us / 1e6:.1f} TFLOP/s algorithmic, {mult * 2.0 * M * N * K / us * 1e6 / peak:.3f} of the "
              f"{'f32' if mode == 0 else 'bf16'} matrix pipe's peak as issued")
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        print(f"   {cyc:.0f} GPU cycles per launch" + (f" ({cyc / avg[mode] / 1e3:.2f} GHz)" if mode in avg else ""))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            print(f"   MFMA pipe busy: {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.3f} of the SIMD-cycles")
        if "SQ_BUSY_CYCLES" in c:
            print(f"   SQ_BUSY_CYCLES {c['SQ_BUSY_CYCLES']:.0f}")
        for k, label in (("SQ_ACTIVE_INST_VALU", "vector ALU issue"), ("SQ_ACTIVE_INST_LDS", "LDS issue"), ("SQ_ACTIVE_INST_SCA", "scalar issue"),
                         ("SQ_ACTIVE_INST_MISC", "misc issue"), ("SQ_INST_CYCLES_VMEM", "vector-memory issue")):
            if k in c:   # quad-cycles summed over waves -> cycles per SIMD: x 4 / 1024
                print(f"   {label}: {c[k]:.0f} quad-cycles = {4 * c[k] / (1024 * cyc):.3f} of the SIMD-cycles")
        if "SQ_LDS_BANK_CONFLICT" in c:
            print(f"   LDS bank-conflict cycles: {c['SQ_LDS_BANK_CONFLICT']:.0f} = {c['SQ_LDS_BANK_CONFLICT'] / (256 * cyc):.3f} of the CU-cycles")
    if "SQ_WAVE_CYCLES" in c:
        wc = c["SQ_WAVE_CYCLES"]
        print("   waves: issuing {:.3f}, issue-stalled {:.3f}, at waits (waitcnt / barrier) {:.3f} of their cycles".format(
            c.get("SQ_ACTIVE_INST_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_WAIT_ANY", 0) / wc))
    waves = 512 * 4
    chunks = (K + 31) // 32
    for k, label in (("SQ_INSTS_VALU", "vector ALU"), ("SQ_INSTS_MFMA", "MFMA"), ("SQ_INSTS_LDS", "LDS"), ("SQ_INSTS_VMEM", "vector memory"),
                     ("SQ_INSTS_SALU", "scalar ALU")):
        if k in c:
            print(f"   {label} instructions: {c[k]:.0f} per launch = {c[k] / waves / chunks:.1f} per wave and 32-deep chunk")
    extra = sorted(k for k in c if k.startswith("SQ_INSTS_VALU_MFMA"))
    for k in extra:
        print(f"   {k}: {c[k]:.0f}")
