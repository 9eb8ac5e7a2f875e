#!/usr/bin/env python3
"""Embedding-gather roofline (K1): (S+2)*B rows of D fp32 from an N-row table, bytes = rows*D*4 read + written + 8 B/index."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pivotcvae_amd import ops
N, D, B, S = 1_000_000, 128, 8192, 10
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
table = torch.rand(N, D, device=dev, generator=g)
idx = torch.randint(0, N, (B * (S + 2),), device=dev, generator=g)
out = torch.empty(B * (S + 2), D, device=dev)
flush = torch.empty(128 * 1024 * 1024, device=dev)  # 512 MB: evicts the 256 MB Infinity Cache between launches
best, tot, n = 1e9, 0.0, 20
for it in range(n + 3):
    flush.fill_(float(it))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gather_rows(table, idx, out=out); e1.record(); torch.cuda.synchronize()
    if it >= 3:
        t = e0.elapsed_time(e1); best = min(best, t); tot += t
nbytes = idx.numel() * (2 * D * 4 + 8)
assert torch.equal(out, table[idx])
print(json.dumps({"kernel": "gather_rows_vec4_kernel<16,true>", "rows": idx.numel(), "bytes": nbytes, "avg_us": tot / n * 1e3, "min_us": best * 1e3,
                  "achieved_GBps_avg": nbytes / (tot / n * 1e-3) / 1e9, "achieved_GBps_best": nbytes / (best * 1e-3) / 1e9,
                  "peak_GBps": 8000.0, "frac_avg": nbytes / (tot / n * 1e-3) / 8e12, "cache": "cold (512 MB written between launches)"}))
