#!/usr/bin/env python3
"""soak_train.py [config] [steps] [catalog arithmetic] [batch]: a few hundred optimisation steps on fresh batches (hipGraph replay; the
config's catalog arithmetic - config 4: bf16x6 with exact-f32 MLP GEMMs, as the headline runs - or the one named): the loss must fall, nothing may go non-finite, the PSM stack and the frozen tables must stay bit-identical."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pivotcvae_amd.train_generative import Trainer
c = sys.argv[1] if len(sys.argv) > 1 else "3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS[c]
dtype = sys.argv[3] if len(sys.argv) > 3 else {"2": "f32", "3": "bf16", "4": "bf16x6"}.get(c, "bf16")
cfg = dict(cfg)
if len(sys.argv) > 4:
    cfg["B"] = int(sys.argv[4])
model, _ = bench.build_model(cfg, dev, dtype)
model.set_mlp_precision("f32" if dtype in ("f32", "bf16x6") else "bf16x3")
psm0 = {k: v.clone() for k, v in model.state_dict().items() if k.startswith(("psm_", "userEmbed"))}
tr = Trainer(model, lr=1e-3, beta=bench.BETA, capture_graph=cfg["B"] <= 4096)
# a small "dataset": 16 batches drawn once, visited round robin (the model can fit them: the loss must go down)
data = [bench.synthetic_batch(cfg, cfg["B"], dev, seed=100 + i) for i in range(4)]
hist = []
for i in range(steps):
    s, r, u = data[i % len(data)]
    loss, rec, kld = tr.step(s, r, u)
    if i % 25 == 0 or i == steps - 1:
        hist.append((i, float(loss), float(rec), float(kld)))
        print(hist[-1], flush=True)
assert all(torch.isfinite(torch.tensor(h[1:])).all() for h in hist)
assert hist[-1][2] < hist[0][2] - 0.05, "the reconstruction term did not fall"
for k, v in psm0.items():
    assert torch.equal(model.state_dict()[k], v), k
print("soak ok: rec", hist[0][2], "->", hist[-1][2], "capture", tr.capture_graph, tr.capture_failed)
