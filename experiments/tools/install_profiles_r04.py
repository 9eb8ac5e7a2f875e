#!/usr/bin/env python3
"""install_profiles_r04.py: copy the summaries tools/profile_r04.sh left under gpurun_out/prof_r04/ into profiles/ (the tracked,
judged copies, r04_*) and extend profiles/traffic.json: per profiled kernel the memory-side bytes per launch (FETCH_SIZE x 2
correction of the guide + WRITE_SIZE), the SQ counters and what they derive to (clock under load, MFMAs issued, MFMA-pipe
utilisation, share of wave cycles issuing / stalled / parked)."""
import csv, glob, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", "prof_r04"), os.path.join(ROOT, "profiles")
for f in sorted(glob.glob(os.path.join(src, "*.csv")) + glob.glob(os.path.join(src, "*_bench*.json")) + glob.glob(os.path.join(src, "config*_bench.json"))):
    shutil.copy(f, os.path.join(dst, "r04_" + os.path.basename(f)))
# (workload, kernel substring, traffic.json key, algorithmic flops per launch, algorithmic bytes per launch, what the bytes are)
R4, R3, R5 = 81920, 40960, 163840
JOBS = [
    ("x6_config4", "catalog_ce_x3_pipe_kernel<128, 2, 3>", "config4_bf16x6_gpus1", 4.0 * R4 * 1e6 * 128,
     768e6 + R4 * 128 * 4 + 2 * 2 * R4 * 130 * 4, "table image [N, 384] bf16 once + rx + the 2 ranges' partials written and read"),
    ("bf16_config3", "catalog_ce_bf16_pipe_kernel<64, 4>", "config3_bf16_gpus1", 4.0 * R3 * 1e5 * 64,
     12.8e6 + R3 * 64 * 4 + 8 * R3 * 66 * 4, "bf16 table once + rx + the ranges' partials"),
    ("bf16_config5", "catalog_ce_bf16_pipe_kernel<256, 2>", "config5_bf16_gpus1", 4.0 * R5 * 1e7 * 256,
     5.12e9 + R5 * 256 * 4 + 2 * R5 * 258 * 4, "bf16 table (5.12 GB) once + rx + the one range's partials written and read"),
    ("x3_config5", "catalog_ce_x3_pipe_kernel<256, 1, 2>", "config5_bf16x3_gpus1", 4.0 * R5 * 1e7 * 256,
     10.24e9 + R5 * 256 * 4 + 20 * R5 * 258 * 4, "two [N, 256] bf16 images (10.24 GB) once + rx + the 20 ranges' partials"),
]
tpath = os.path.join(dst, "traffic.json")
t = json.load(open(tpath))
for name, kern, key, flops, alg_bytes, what in JOBS:
    c = {}
    short = kern[:34]
    for n in ("FETCH_SIZE", "WRITE_SIZE", "SQ1", "SQ2"):
        p = os.path.join(src, f"{name}_pmc_{n}.csv")
        if not os.path.exists(p):
            continue
        for r in csv.DictReader(open(p)):
            if short.split("<")[0] in r["kernel"] and ("pipe_kernel<" + kern.split("<")[1][:6]) in r["kernel"]:
                c[r["counter"]] = float(r["mean_per_dispatch"])
    avg_ns = None
    if not os.path.exists(os.path.join(src, f"{name}_kernel_stats.csv")):   # this part of tools/profile_r04.sh was not re-run: entry kept
        continue
    for r in csv.DictReader(open(os.path.join(src, f"{name}_kernel_stats.csv"))):
        if kern in r["Name"]:
            avg_ns = float(r["AverageNs"])
    if not c or avg_ns is None:
        print("missing", name, sorted(c), avg_ns)
        continue
    fetch, write = c["FETCH_SIZE"] * 1024 * 2, c["WRITE_SIZE"] * 1024
    cycles = c["GRBM_GUI_ACTIVE"] / 8
    wc = c["SQ_WAVE_CYCLES"]
    peak = 2500e12
    t[key] = fetch + write
    t["_r04_" + name] = {
        "round": "r04", "kernel": kern, "kernel_trace_avg_ms": avg_ns / 1e6,
        "FETCH_SIZE_KB_mean": c["FETCH_SIZE"], "WRITE_SIZE_KB_mean": c["WRITE_SIZE"],
        "fetch_correction": "x2 (gfx950: 128-B requests tallied at 64 B for 16 B/lane streaming reads; MI355X_MICROARCH.md HBM section)",
        "memory_side_bytes_per_launch": fetch + write, "memory_side_GBps": (fetch + write) / avg_ns,
        "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_are": what, "traffic_over_algorithmic": (fetch + write) / alg_bytes,
        "note": "FETCH_SIZE / WRITE_SIZE count the L2s' memory-side requests: Infinity-Cache hits are included (an upper bound of HBM bytes)",
        "SQ": {k: c[k] for k in sorted(c) if k.startswith(("SQ_", "GRBM"))},
        "derived": {"gpu_cycles_per_launch": cycles, "clock_GHz_under_load": cycles / avg_ns,
                    "mfma_issued": c["SQ_VALU_MFMA_BUSY_CYCLES"] / 16, "mfma_pipe_utilisation": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cycles),
                    "algorithmic_TFLOPs": flops / avg_ns / 1e3, "frac_of_dense_bf16_peak": flops / (avg_ns * 1e-9) / peak,
                    "wave_cycles_issuing": c["SQ_ACTIVE_INST_ANY"] / wc, "wave_cycles_issue_stalled": c["SQ_WAIT_INST_ANY"] / wc,
                    "wave_cycles_parked_waitcnt_or_barrier": c["SQ_WAIT_ANY"] / wc},
        "collected_with": "tools/profile_r04.sh: rocprofv3 --pmc <set> --kernel-trace, one pass per set (SQ1, SQ2, FETCH_SIZE, WRITE_SIZE)"}
    print(name, json.dumps(t["_r04_" + name]["derived"]), "traffic x algorithmic: %.1f" % ((fetch + write) / alg_bytes), "%.0f GB/s" % ((fetch + write) / avg_ns))
json.dump(t, open(tpath, "w"), indent=1)
