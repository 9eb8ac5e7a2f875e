#!/usr/bin/env python3
"""Per-launch breakdown of the MLP GEMMs of one config-4 train step: kind, M x N x K, microseconds (HIP events), TFLOP/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from pivotcvae_amd import ops, _hip
from pivotcvae_amd.train_generative import Trainer
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "4"])
if len(sys.argv) > 2:
    cfg["B"] = int(sys.argv[2])
model, st = bench.build_model(cfg, dev, "bf16")
trainer = Trainer(model, lr=bench.LR, beta=bench.BETA)
s, r, u = bench.synthetic_batch(cfg, cfg["B"], dev)
rec = []
L = _hip.lib()
orig = {k: getattr(L, k) for k in ("pcvae_linear_fwd", "pcvae_linear_bwd_input", "pcvae_linear_bwd_weight")}
def wrap(name, f):
    def g(*a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rc = f(*a); e1.record()
        if name == "pcvae_linear_fwd": M, N, K = a[7], a[8], a[9]
        elif name == "pcvae_linear_bwd_input": M, N, K = a[8], a[9], a[10]
        else: M, N, K = a[7], a[8], a[9]
        rec.append((name[13:], int(M), int(N), int(K), e0, e1))
        return rc
    return g
for _ in range(2):
    trainer.step(s, r, u)
for k, f in orig.items():
    setattr(L, k, wrap(k, f))
trainer.step(s, r, u)
torch.cuda.synchronize()
tot = 0
for name, M, N, K, e0, e1 in rec:
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    print(f"{name:12s} M={M:6d} N={N:5d} K={K:5d} {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s")
print("total", tot, "us")
