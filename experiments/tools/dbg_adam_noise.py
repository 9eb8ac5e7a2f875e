"""dbg_adam_noise.py: three Adam steps of the D = 64 stated golden on the HIP path - which tensors leave the strict tolerance, how many
elements, and how large the reference gradient is there (noise-following weights vs a LeakyReLU kink met between steps)."""
import sys, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tests.gpu_util import build_from_golden, dev
from tests.helpers import load
from pivotcvae_amd.train_generative import Trainer
g = load("stated_d64_gt_pi")
m = build_from_golden(g); m.set_catalog_precision("f32")
tr = Trainer(m, lr=g.meta["lr"], beta=g.meta["beta"])
s, r, u = dev(g.t("s")), dev(g.t("r")), dev(g.t("u"))
for step in range(3):
    tr.step(s, r, u, eps=dev(g.t(f"adam/eps{step}")))
    if step in (0, 2):
        sd = m.state_dict()
        for k, v in g.sub(f"adam/step{step + 1}").items():
            a, b = sd[k].cpu(), torch.as_tensor(v)
            diff = (a - b).abs(); off = diff > 3e-6 + 1e-4 * b.abs()
            if off.any():
                gr = torch.as_tensor(g.sub("grad")[k]) if k in g.sub("grad") else None
                gs = gr[off].abs() if gr is not None else None
                print(step + 1, k, tuple(a.shape), int(off.sum()), f"{float(off.float().mean()):.2e}", "max diff %.2e" % float(diff.max()),
                      "ref |grad| at those: min %.1e max %.1e; tensor grad scale %.1e" % (float(gs.min()), float(gs.max()), float(gr.abs().max())) if gs is not None else "")
