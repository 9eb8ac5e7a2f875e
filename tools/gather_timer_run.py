#!/usr/bin/env python3
"""gather_timer_run.py [reps]: `reps` cold-cache launches of the standalone gather (K1) at the north-star shape (98 304 rows of 512 B
= the item + user + pivot rows of one config-4 step) and of the train step's fused gather (assemble_inputs), each timed by the HIP
events attached to its own dispatch (bench.kernel_timer_run).  Run plainly, and under rocprofv3 --kernel-trace --stats / --pmc:
the kernel trace's average durations must agree with the numbers printed here (tools/profile_gather.sh)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from pivotcvae_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["4"]
N, S, D, B = cfg["N"], cfg["S"], cfg["D"], cfg["B"]
g = torch.Generator(device=dev).manual_seed(3)
E = torch.rand(N, D, device=dev, generator=g)
U = torch.rand(bench.N_USER, D, device=dev, generator=g)
n_idx = B * (S + 2)
idx = torch.randint(0, N, (n_idx,), device=dev, generator=g)
out = torch.empty(n_idx, D, device=dev)
s, r, u = bench.synthetic_batch(cfg, B, dev)
flush = torch.empty(128 * 1024 * 1024, device=dev)
# the launch floor: the same kernel family on ONE row (a D = 64 table: another template instance, so rocprofv3's per-kernel stats
# keep it apart) - what a dispatch costs with nothing to move, in a plain process and in a traced one
E64 = torch.rand(1024, 64, device=dev, generator=g)
idx1 = torch.zeros(1, dtype=torch.int64, device=dev)
out1 = torch.empty(1, 64, device=dev)
res = {"gather": [], "assemble": [], "floor": []}
for it in range(reps + 2):
    flush.fill_(float(it)); torch.cuda.synchronize()
    a = bench.kernel_timer_run(lambda: ops.gather_rows(E, idx, out=out), bench.TIMER_GATHER)
    flush.fill_(float(it) + 0.5); torch.cuda.synchronize()
    b = bench.kernel_timer_run(lambda: ops.assemble_inputs(E, U, s, r, u, bench.Z), bench.TIMER_ASSEMBLE)
    c = bench.kernel_timer_run(lambda: ops.gather_rows(E64, idx1, out=out1), bench.TIMER_GATHER)
    if it >= 2:
        res["gather"] += a; res["assemble"] += b; res["floor"] += c
gb = n_idx * (2 * D * 4 + 8)
ab = B * (4 * ((S + 1) * D + (S * D + S + 1 + D) + (S + 1 + D) + (S + 1 + 2 * D) + D) + 8 * (S + 1))
for k, nb in (("gather", gb), ("assemble", ab), ("floor", 1 * (2 * 64 * 4 + 8))):
    t = sum(res[k]) / len(res[k])
    print(json.dumps({"kernel": k, "launches": len(res[k]), "us_avg": t * 1e3, "us_min": min(res[k]) * 1e3, "us_max": max(res[k]) * 1e3,
                      "bytes": nb, "GBps": nb / t / 1e6, "frac_of_8TBps": nb / t / 8e9}))
