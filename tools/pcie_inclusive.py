#!/usr/bin/env python3
"""pcie_inclusive.py [config] [steps]: the train step with the batch handed over from HOST memory every step (s, r, u as the
reference's DataLoader yields them, pinned or pageable -> .to(device)) against the same step on resident tensors (bench.py's `value`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pivotcvae_amd.train_generative import Trainer
c = sys.argv[1] if len(sys.argv) > 1 else "4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS[c])
dtype = {"3": "bf16", "5": "bf16"}.get(c, "bf16x6" if cfg["D"] == 128 else "f32")
model, _ = bench.build_model(cfg, dev, dtype)
model.set_mlp_precision("f32" if dtype in ("f32", "bf16x6") else "bf16x3")
tr = Trainer(model, lr=bench.LR, beta=bench.BETA, capture_graph=False)
s, r, u = bench.synthetic_batch(cfg, cfg["B"], dev)
host = [t.cpu() for t in (s, r, u)]
pinned = [t.pin_memory() for t in host]
nbytes = sum(t.numel() * t.element_size() for t in host)
def run(src):
    for _ in range(2):
        tr.step(*[t.to(dev, non_blocking=True) for t in src]) if src is not None else tr.step(s, r, u)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        if src is None: tr.step(s, r, u)
        else: tr.step(*[t.to(dev, non_blocking=True) for t in src])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps
res = {"resident": run(None), "pageable_host": run(host), "pinned_host": run(pinned)}
print({"config": c, "batch_bytes": nbytes, **{k: f"{v * 1e3:.3f} ms/step = {cfg['B'] / v:.0f} slates/s" for k, v in res.items()}})
