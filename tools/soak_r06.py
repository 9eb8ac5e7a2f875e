#!/usr/bin/env python3
"""soak_r06.py: 300 optimisation steps (hipGraph replay, 8 batches in rotation) per loss mode x pivot rule x MLP arithmetic at config 2's
shape - the round-6 switches over a TRAJECTORY, not one step: candidate sets drawn from a dataset id range (n_items < N), the
bf16x6 / bf16x3 MLP GEMMs against the exact-f32 ones (same seeds, same draws: the loss curves must stay together), frozen tensors
(tables, the PSM stack) bit-unchanged, parameters finite, the reconstruction term falling.

    python tools/soak_r06.py > profiles/r06_soak_mlp_arithmetics.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pivotcvae_amd.train_generative import Trainer  # noqa: E402

dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS["2"])
N = cfg["N"]
STEPS = 300
print(f"# config 2 shape (N={N}, S={cfg['S']}, D={cfg['D']}, B={cfg['B']}), {STEPS} steps, lr 1e-3, graph replay; loss terms at steps 0 / 100 / 200 / 299")
for mode in ({"n_candidate": 1000, "n_items": N - 1500}, {"n_neg": 1000}, {}):
    for model_key in ("pivotcvae_gt_pi", "pivotcvae_sgt_pi"):
        curves = {}
        for mlp in ("f32", "bf16x6", "bf16x3"):
            c2 = dict(cfg, model=model_key)
            m, _ = bench.build_model(c2, dev, "f32")
            m.set_mlp_precision(mlp)
            frozen = {k: v.clone() for k, v in m.state_dict().items() if k.startswith(("psm_", "docEmbed", "userEmbed"))}
            tr = Trainer(m, lr=1e-3, beta=0.001, capture_graph=True, **mode)
            hist = []
            for step in range(STEPS):
                s, r, u = bench.synthetic_batch(c2, c2["B"], dev, seed=step % 8)      # 8 batches in rotation
                if "n_items" in mode:
                    s = s % mode["n_items"]                                          # the dataset only uses ids below max_iid + 1
                l, rec, k = tr.step(s, r, u)
                if step % 100 == 0 or step == STEPS - 1:
                    hist.append((step, float(l), float(rec), float(k)))
            ok = all(torch.equal(m.state_dict()[k], v) for k, v in frozen.items())
            finite = all(torch.isfinite(p).all().item() for p in m.parameters())
            curves[mlp] = hist
            print(mode, model_key, f"mlp={mlp:6s}", "graph" if tr._graph is not None else "eager",
                  " ".join(f"[{st}: {a:.4f} = {b:.4f} + b*{c:.2f}]" for st, a, b, c in hist), "frozen unchanged:", ok, "finite:", finite, flush=True)
            assert ok and finite and hist[-1][2] < hist[0][2]
        for other, tol in (("bf16x6", 2e-3), ("bf16x3", 5e-3)):
            dev_rel = max(abs(a[2] - b[2]) / abs(b[2]) for a, b in zip(curves[other], curves["f32"]))
            print(f"    reconstruction term, {other} vs f32 along the trajectory: max relative difference {dev_rel:.2e} (bound {tol:g})", flush=True)
            assert dev_rel < tol, (other, dev_rel)
print("soak ok")
