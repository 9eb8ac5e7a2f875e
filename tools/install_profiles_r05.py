#!/usr/bin/env python3
"""install_profiles_r05.py: copy the summaries tools/profile_r05.sh left under gpurun_out/prof_r05/ into profiles/ (the tracked,
judged copies, r05_*) and refresh profiles/traffic.json: the candidate-mode kernel's memory-side bytes (new keys
config4_cand{1000,50}_gpus1), the gather kernel's FETCH + WRITE against its algorithmic bytes, and - when part B was run - the
config-4 bf16 / bf16x3 entries, which still carried round-1 / round-3 kernels' bytes."""
import csv, glob, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", "prof_r05"), os.path.join(ROOT, "profiles")
for f in sorted(glob.glob(os.path.join(src, "*.csv")) + glob.glob(os.path.join(src, "*_bench*.json")) +
                glob.glob(os.path.join(src, "config*_bench.json")) + glob.glob(os.path.join(src, "gather_timer_*.txt"))):
    shutil.copy(f, os.path.join(dst, "r05_" + os.path.basename(f)))
tpath = os.path.join(dst, "traffic.json")
t = json.load(open(tpath))
R4 = 81920


def counters(name, kern):
    c = {}
    for n in ("FETCH_SIZE", "WRITE_SIZE"):
        p = os.path.join(src, f"{name}_pmc_{n}.csv")
        if os.path.exists(p):
            for r in csv.DictReader(open(p)):
                if kern in r["kernel"] and r["counter"] == n:
                    c[n] = float(r["mean_per_dispatch"])
    return c


def avg_ms(name, kern):
    p = os.path.join(src, f"{name}_kernel_stats.csv")
    if not os.path.exists(p):
        return None
    for r in csv.DictReader(open(p)):
        if kern in r["Name"]:
            return float(r["AverageNs"]) / 1e6
    return None


FETCH_NOTE = "x2 (gfx950: 128-B requests tallied at 64 B for 16 B/lane reads; MI355X_MICROARCH.md HBM section)"
# (workload, kernel substring in the PMC csv / the kernel stats, traffic.json key, algorithmic bytes per launch, what they are)
JOBS = [
    ("cand1000_config4", "candidate_ce_kernel<128, true, false>", "config4_cand1000_gpus1", R4 * 1001 * 512.0 + R4 * (2 * 512 + 24),
     "REQUESTED bytes: 1000 candidate rows + the target row of 512 B per slate slot, rx read, dx + nll + lse written "
     "(a 0.51 GB table: every row is re-read ~82 times, from the caches)"),
    ("cand50_config4", "candidate_ce_kernel<128, true, false>", "config4_cand50_gpus1", R4 * 51 * 512.0 + R4 * (2 * 512 + 24),
     "REQUESTED bytes: 50 candidate rows + the target row per slate slot, rx, dx, nll, lse"),
    ("bf16_config4", "catalog_ce_bf16_pipe_kernel<128, 4>", "config4_bf16_gpus1", 256e6 + R4 * 128 * 4 + 2 * 2 * R4 * 130 * 4,
     "bf16 table once + rx + the ranges' partials written and read"),
    ("bf16_config3", "catalog_ce_bf16_pipe_kernel<64, 4>", "config3_bf16_gpus1", 12.8e6 + 40960 * 64 * 4 + 3 * 40960 * 66 * 4,
     "bf16 table once + rx + the 3 ranges' partials"),
    ("bf16x3_config4", "catalog_ce_x3_pipe_kernel<128, 2, 2>", "config4_bf16x3_gpus1", 512e6 + R4 * 128 * 4 + 2 * 2 * R4 * 130 * 4,
     "table image [N, 256] bf16 once + rx + the ranges' partials written and read"),
]
for name, kern, key, alg, what in JOBS:
    c, ms = counters(name, kern.split("<")[0]), avg_ms(name, kern)
    if len(c) < 2 or ms is None:
        print("not profiled this time:", name, sorted(c), ms)
        continue
    fetch, write = c["FETCH_SIZE"] * 1024 * 2, c["WRITE_SIZE"] * 1024
    t[key] = fetch + write
    t["_r05_" + name] = {"round": "r05", "kernel": kern, "kernel_trace_avg_ms": ms, "FETCH_SIZE_KB_mean": c["FETCH_SIZE"],
                         "WRITE_SIZE_KB_mean": c["WRITE_SIZE"], "fetch_correction": FETCH_NOTE,
                         "memory_side_bytes_per_launch": fetch + write, "memory_side_GBps": (fetch + write) / ms / 1e6,
                         "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_are": what,
                         "traffic_over_algorithmic": (fetch + write) / alg,
                         "note": "FETCH_SIZE / WRITE_SIZE count the L2s' memory-side requests: Infinity-Cache hits are included",
                         "collected_with": "tools/profile_r05.sh: rocprofv3 --pmc <counter> --kernel-trace, one pass per counter"}
    sq = {}
    for n in ("SQ1", "SQ2"):
        pth = os.path.join(src, f"{name}_pmc_{n}.csv")
        if os.path.exists(pth):
            for r in csv.DictReader(open(pth)):
                if kern.split("<")[0] in r["kernel"]:
                    sq[r["counter"]] = float(r["mean_per_dispatch"])
    if "SQ_WAVE_CYCLES" in sq:
        wc = sq["SQ_WAVE_CYCLES"]
        t["_r05_" + name]["SQ"] = sq
        if "SQ_VALU_MFMA_BUSY_CYCLES" in sq:   # an MFMA kernel: pipe utilisation as in round 4's entries
            cyc = sq["GRBM_GUI_ACTIVE"] / 8
            flops = 4.0 * 40960 * 1e5 * 64 if name == "bf16_config3" else None
            t["_r05_" + name]["mfma"] = {"gpu_cycles_per_launch": cyc, "clock_GHz_under_load": cyc / (ms * 1e6),
                                         "mfma_pipe_utilisation": sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc),
                                         "lds_bank_conflict_cycles": sq.get("SQ_LDS_BANK_CONFLICT"),
                                         "frac_of_dense_bf16_peak": (flops / (ms * 1e-3) / 2500e12) if flops else None}
        t["_r05_" + name]["derived"] = {
            "wave_cycles_issuing": sq["SQ_ACTIVE_INST_ANY"] / wc, "wave_cycles_issue_stalled": sq["SQ_WAIT_INST_ANY"] / wc,
            "wave_cycles_parked_waitcnt_or_barrier": sq["SQ_WAIT_ANY"] / wc,
            **({"clock_GHz_under_load": sq["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e6), "waves": sq.get("SQ_WAVES"),
                "valu_instructions_per_wave": sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"]} if "SQ_WAVES" in sq else {})}
        print(name, json.dumps(t["_r05_" + name]["derived"]))
    print(name, "%.3f ms" % ms, "memory side %.2f GB = %.2f x algorithmic, %.0f GB/s" % ((fetch + write) / 1e9, (fetch + write) / alg,
                                                                                       (fetch + write) / ms / 1e6))
for stale, fresh in (("_round1", "_r05_bf16_config4"), ("_source", "_r05_bf16x3_config4")):
    if fresh in t and stale in t:
        del t[stale]   # the entries that described earlier rounds' kernels under the current keys

# the gather (K1): rocprofv3's average duration must agree with the dispatch-event timer; FETCH + WRITE = the algorithmic bytes
gk = "gather_rows_coal_kernel<32"   # the D = 128 instance (the <16> one is the one-row launch floor)
gs = os.path.join(src, "gather_kernel_stats.csv")
if os.path.exists(gs):
    ms = None
    for r in csv.DictReader(open(gs)):
        if gk in r["Name"]:
            ms = float(r["AverageNs"]) / 1e6
    c = {}
    for n in ("FETCH_SIZE", "WRITE_SIZE"):
        p = os.path.join(src, f"gather_pmc_{n}.csv")
        if os.path.exists(p):
            for line in open(p):
                parts = next(csv.reader([line]))
                if len(parts) == 4 and gk in parts[0] and parts[1] == n:
                    c[n] = float(parts[3])
    timer = {}
    p = os.path.join(src, "gather_timer_plain.txt")
    if os.path.exists(p):
        for line in open(p):
            if line.startswith("{"):
                d = json.loads(line)
                timer[d["kernel"]] = d
    timer_under = {}
    p = os.path.join(src, "gather_timer_under_rocprof.txt")
    if os.path.exists(p):
        for line in open(p):
            if line.startswith("{"):
                d = json.loads(line)
                timer_under[d["kernel"]] = d
    floor_trace = None   # the one-row launches of the D = 64 instance: what a dispatch costs in the traced process
    for r in csv.DictReader(open(gs)):
        if "gather_rows_coal_kernel<16" in r["Name"]:
            floor_trace = float(r["AverageNs"]) / 1e3
    if ms is not None and len(c) == 2:
        alg = 98304 * (2 * 512 + 8)
        fetch, write = c["FETCH_SIZE"] * 1024 * 2, c["WRITE_SIZE"] * 1024    # (16 B/lane row reads: the guide's x2 correction)
        t["_r05_gather"] = {"round": "r05", "kernel": "gather_rows_coal_kernel<32, true>", "kernel_trace_avg_us": ms * 1e3,
                            "dispatch_event_timer_avg_us": timer.get("gather", {}).get("us_avg"),
                            "dispatch_event_timer_avg_us_under_rocprofv3": timer_under.get("gather", {}).get("us_avg"),
                            "launch_floor_us": {"dispatch_event_timer_plain": timer.get("floor", {}).get("us_avg"),
                                                "dispatch_event_timer_under_rocprofv3": timer_under.get("floor", {}).get("us_avg"),
                                                "rocprofv3_kernel_trace": floor_trace,
                                                "what": "the same kernel family moving ONE row: the cost of a dispatch with nothing to move"},
                            "timing_note": "the dispatch-event timer (HIP events attached to the launch) in a plain process; rocprofv3's "
                                           "kernel trace of the same program; and the same timer INSIDE the traced process (the event pair "
                                           "then also spans the profiler's own packets: not a kernel duration)",
                            "FETCH_SIZE_KB_mean": c["FETCH_SIZE"], "WRITE_SIZE_KB_mean": c["WRITE_SIZE"], "fetch_correction": FETCH_NOTE,
                            "memory_side_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": alg,
                            "traffic_over_algorithmic": (fetch + write) / alg,
                            "GBps_by_kernel_trace": alg / ms / 1e6, "frac_of_8TBps_by_kernel_trace": alg / ms / 8e9,
                            "collected_with": "tools/profile_gather.sh (tools/profile_r05.sh B): cold-cache launches of the north-star "
                                              "gather shape, rocprofv3 --kernel-trace --stats and one --pmc pass per counter"}
        print("gather", json.dumps(t["_r05_gather"]))
json.dump(t, open(tpath, "w"), indent=1)
