#!/usr/bin/env python3
"""per-tensor gradient error of the bf16x3 MLP / catalog arithmetics against the reference's gradients (stated goldens)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PCVAE_GEMM_SMALL_BELOW"] = sys.argv[2] if len(sys.argv) > 2 else "0"
import torch
from tests.gpu_util import build_from_golden, dev
from tests.helpers import load
from pivotcvae_amd.train_generative import Trainer
name = sys.argv[1] if len(sys.argv) > 1 else "stated_config2_gt_pi"
g = load(name)
want = g.sub("grad")
for mlp, cat in (("f32", "f32"), ("bf16x3", "f32"), ("f32", "bf16x3"), ("bf16x3", "bf16x3")):
    m = build_from_golden(g).set_catalog_precision(cat).set_mlp_precision(mlp)
    tr = Trainer(m, lr=g.meta["lr"], beta=g.meta["beta"])
    tr.local_phase(dev(g.t("s")), dev(g.t("r")), dev(g.t("u")), dev(g.t("full/eps")))
    row = []
    for k, p in m.named_parameters():
        if k in want:
            w = want[k]
            d = (p.grad.cpu() - w).abs()
            row.append(f"{k.split('.')[0]}.{k.split('.')[1][0]} {float(d.max()) / float(w.abs().max()):.1e}")
    print(f"mlp {mlp:6s} cat {cat:6s} |", " ".join(row))
