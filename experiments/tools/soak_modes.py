import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from pivotcvae_amd.train_generative import Trainer
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS["2"])
for mode in ({"n_candidate": 1000}, {"n_neg": 1000}, {}):
    for model_key in ("pivotcvae_gt_pi", "pivotcvae_sgt_pi"):
        c2 = dict(cfg, model=model_key)
        m, _ = bench.build_model(c2, dev, "f32")
        frozen = {k: v.clone() for k, v in m.state_dict().items() if k.startswith(("psm_", "docEmbed", "userEmbed"))}
        tr = Trainer(m, lr=1e-3, beta=0.001, capture_graph=True, **mode)
        hist = []
        for step in range(300):
            s, r, u = bench.synthetic_batch(c2, c2["B"], dev, seed=step % 8)      # 8 batches in rotation
            l, rec, k = tr.step(s, r, u)
            if step % 50 == 0 or step == 299:
                hist.append((step, round(float(l), 4), round(float(rec), 4), round(float(k), 2)))
        ok = all(torch.equal(m.state_dict()[k], v) for k, v in frozen.items())
        finite = all(torch.isfinite(p).all().item() for p in m.parameters())
        print(mode, model_key, "graph" if tr._graph is not None else "eager", hist, "frozen unchanged:", ok, "finite:", finite, flush=True)
        assert ok and finite and hist[-1][2] < hist[0][2]
print("soak ok")
