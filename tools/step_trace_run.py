#!/usr/bin/env python3
"""step_trace_run.py [config] [B] [dtype] [steps] [mlp arithmetic] [n_candidate]: a few EAGER train steps, to be run under rocprofv3
--kernel-trace (tools/step_trace_list.py then prints one step's launches in order)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from pivotcvae_amd.train_generative import Trainer
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "4"])
if len(sys.argv) > 2:
    cfg["B"] = int(sys.argv[2])
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
model, st = bench.build_model(cfg, dev, dtype)
if len(sys.argv) > 5:
    model.set_mlp_precision(sys.argv[5])
trainer = Trainer(model, lr=bench.LR, beta=bench.BETA, n_candidate=int(sys.argv[6]) if len(sys.argv) > 6 else None)
s, r, u = bench.synthetic_batch(cfg, cfg["B"], dev)
for _ in range(steps):
    trainer.step(s, r, u)
torch.cuda.synchronize()
print("done", steps)
