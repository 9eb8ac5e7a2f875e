#!/usr/bin/env python3
"""screen_pmc_run.py [config]: the program tools/screen_pmc.sh profiles - eight batches of greedy generation (recommend(return_item=True)) at a
BASELINE config (default 3: N = 10^5, S = 10, D = 64, B = 4096), eager launches; the screening kernels carry the counters."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "3"])
model, _ = bench.build_model(cfg, dev, "bf16")
B, S = cfg["B"], cfg["S"]
g = torch.Generator(device=dev).manual_seed(7)
u = torch.randint(0, bench.N_USER, (B, 1), device=dev, generator=g)
ctx = (torch.rand(B, S, device=dev, generator=g) < 0.5).float()
eps = torch.randn(B, bench.Z, device=dev, generator=g)
with torch.no_grad():
    for _ in range(8):
        items, _ = model.recommend(ctx, u, return_item=True, eps=eps)
    torch.cuda.synchronize()
print("ids checksum", int(items.sum()))
