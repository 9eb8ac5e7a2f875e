#!/usr/bin/env python3
"""Embedding-gather roofline (K1): (S+2)*B rows of D fp32 from an N-row table, bytes = rows*D*4 read + written + 8 B/index.

Two timings, both from cold caches (512 MB written between measurements, > the 256 MB Infinity Cache):
  single   one launch between one HIP event pair (the pair itself costs ~2.4 us on top of the kernel: an EMPTY kernel
           measures 6.0 us event-to-event and 3.6 us in rocprofv3's kernel trace)
  train    --tables T launches back to back between one event pair, each on its OWN cold table / index set / output
           (no launch re-reads what an earlier one brought in), time / T
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pivotcvae_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument("--mult", type=int, default=1)
ap.add_argument("--tables", type=int, default=4)
args = ap.parse_args()
N, D, B, S = 1_000_000, 128, 8192 * args.mult, 10
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
T = args.tables
tables = [torch.rand(N, D, device=dev, generator=g) for _ in range(T)]
idxs = [torch.randint(0, N, (B * (S + 2),), device=dev, generator=g) for _ in range(T)]
outs = [torch.empty(B * (S + 2), D, device=dev) for _ in range(T)]
flush = torch.empty(128 * 1024 * 1024, device=dev)  # 512 MB: evicts the 256 MB Infinity Cache between launches
nbytes = idxs[0].numel() * (2 * D * 4 + 8)
res = {}
for mode, k in (("single", 1), ("train", T)):
    best, tot, n = 1e9, 0.0, 16
    for it in range(n + 3):
        flush.fill_(float(it))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for j in range(k):
            ops.gather_rows(tables[j], idxs[j], out=outs[j])
        e1.record(); torch.cuda.synchronize()
        if it >= 3:
            t = e0.elapsed_time(e1) / k; best = min(best, t); tot += t
    res[mode] = {"avg_us": tot / n * 1e3, "min_us": best * 1e3, "GBps_avg": nbytes / (tot / n * 1e-3) / 1e9, "frac_avg": nbytes / (tot / n * 1e-3) / 8e12,
                 "frac_best": nbytes / (best * 1e-3) / 8e12}
for j in range(T):
    assert torch.equal(outs[j], tables[j][idxs[j]])
print(json.dumps({"rows": idxs[0].numel(), "bytes": nbytes, **res}))
