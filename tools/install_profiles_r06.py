#!/usr/bin/env python3
"""install_profiles_r06.py: copy the summaries tools/profile_r06.sh left under gpurun_out/prof_r06/ into profiles/ (the tracked,
judged copies, r06_*) and refresh profiles/traffic.json from THIS tree's PMC passes of the headline kernel
(catalog_ce_x3_pipe_kernel<128, 2, 3>: key config4_bf16x6_gpus1 + the _r06_x6_config4 record with the SQ-derived figures)."""
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", "prof_r06"), os.path.join(ROOT, "profiles")
for f in sorted(glob.glob(os.path.join(src, "*.csv")) + glob.glob(os.path.join(src, "*_bench*.json")) + glob.glob(os.path.join(src, "*.txt"))):
    if os.path.getsize(f) == 0 or os.path.basename(f).startswith("assemble_pmc"):
        continue
    shutil.copy(f, os.path.join(dst, "r06_" + os.path.basename(f)))
tpath = os.path.join(dst, "traffic.json")
t = json.load(open(tpath))
R4 = 81920
FETCH_NOTE = "x2 (gfx950: 128-B requests tallied at 64 B for 16 B/lane reads; MI355X_MICROARCH.md HBM section)"


def pmc(name, set_name, kern):
    out = {}
    p = os.path.join(src, f"{name}_pmc_{set_name}.csv")
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            if kern in r["kernel"]:
                out[r["counter"]] = float(r["mean_per_dispatch"])
    return out


def avg_ms(name, kern):
    p = os.path.join(src, f"{name}_kernel_stats.csv")
    if not os.path.exists(p):
        return None, None
    for r in csv.DictReader(open(p)):
        if kern in r["Name"]:
            return float(r["AverageNs"]) / 1e6, int(r["Calls"])
    return None, None


name, kern = "x6_config4", "catalog_ce_x3_pipe_kernel<128, 2, 3>"
short = kern.split("<")[0]
ms, calls = avg_ms(name, kern)
c = {**pmc(name, "FETCH_SIZE", short), **pmc(name, "WRITE_SIZE", short)}
sq = {**pmc(name, "SQ1", short), **pmc(name, "SQ2", short)}
if ms is not None and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    fetch, write = c["FETCH_SIZE"] * 1024 * 2, c["WRITE_SIZE"] * 1024
    alg = 768e6 + R4 * 128 * 4 + 2 * 2 * R4 * 130 * 4
    flops = 4.0 * R4 * 1e6 * 128
    rec = {"round": "r06", "kernel": kern, "kernel_trace_avg_ms": ms, "kernel_trace_calls": calls, "FETCH_SIZE_KB_mean": c["FETCH_SIZE"],
           "WRITE_SIZE_KB_mean": c["WRITE_SIZE"], "fetch_correction": FETCH_NOTE, "memory_side_bytes_per_launch": fetch + write,
           "memory_side_GBps": (fetch + write) / ms / 1e6, "algorithmic_bytes_per_launch": alg,
           "algorithmic_bytes_are": "table image [N, 384] bf16 (c0 | c1 | c2) once + rx + the ranges' partials written and read",
           "traffic_over_algorithmic": (fetch + write) / alg, "algorithmic_TFLOPs": flops / (ms * 1e-3) / 1e12,
           "frac_of_dense_bf16_peak_algorithmic": flops / (ms * 1e-3) / 2500e12, "mfma_issue_frac": 6 * flops / (ms * 1e-3) / 2500e12,
           "note": "FETCH_SIZE / WRITE_SIZE count the L2s' memory-side requests: Infinity-Cache hits are included; one table image per 32 "
                   "row blocks by design (the XCDs' L2s stream the same range)",
           "collected_with": "tools/profile_r06.sh H: rocprofv3 --kernel-trace --stats, and one --pmc pass per counter set"}
    if "SQ_WAVE_CYCLES" in sq:
        wc = sq["SQ_WAVE_CYCLES"]
        rec["SQ"] = sq
        rec["derived"] = {"wave_cycles_issuing": sq["SQ_ACTIVE_INST_ANY"] / wc, "wave_cycles_issue_stalled": sq["SQ_WAIT_INST_ANY"] / wc,
                          "wave_cycles_parked_waitcnt_or_barrier": sq["SQ_WAIT_ANY"] / wc}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in sq and "GRBM_GUI_ACTIVE" in sq:
        cyc = sq["GRBM_GUI_ACTIVE"] / 8
        rec["mfma"] = {"gpu_cycles_per_launch": cyc, "clock_GHz_under_load": cyc / (ms * 1e6),
                       "mfma_pipe_utilisation": sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc),
                       "lds_bank_conflict_cycles": sq.get("SQ_LDS_BANK_CONFLICT")}
    t["config4_bf16x6_gpus1"] = fetch + write
    t["_r06_x6_config4"] = rec
    print(json.dumps(rec, indent=1))
else:
    print("headline PMC not collected this time:", ms, sorted(c))
json.dump(t, open(tpath, "w"), indent=1)
