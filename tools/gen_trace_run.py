#!/usr/bin/env python3
"""gen_trace_run.py [config]: a few greedy generate batches (recommend(return_item=True)), to be run under rocprofv3 --kernel-trace --stats"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "4"])
model, st = bench.build_model(cfg, dev, "bf16")
B, S = cfg["B"], cfg["S"]
g = torch.Generator(device=dev).manual_seed(7)
u = torch.randint(0, bench.N_USER, (B, 1), device=dev, generator=g)
ctx = (torch.rand(B, S, device=dev, generator=g) < 0.5).float()
eps = torch.randn(B, bench.Z, device=dev, generator=g)
with torch.no_grad():
    for _ in range(4):
        items, _ = model.recommend(ctx, u, return_item=True, eps=eps)
torch.cuda.synchronize()
print("done")
