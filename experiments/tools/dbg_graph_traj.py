"""dbg_graph_traj.py [B] [steps]: the ELBO terms of every step of a config-4 run, eager against hipGraph replay, from the same initial
state (the two trajectories must agree to rounding: same eps stream, same kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pivotcvae_amd.train_generative import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS["4"], B=B)
out = {}
for mode in ("eager", "graph", "graph2"):
    model, _ = bench.build_model(cfg, dev, "bf16x3")
    model.set_mlp_precision("bf16x3")
    tr = Trainer(model, lr=bench.LR, beta=bench.BETA, capture_graph=mode != "eager")
    s, r, u = bench.synthetic_batch(cfg, B, dev)
    if os.environ.get("BENCHLIKE") == "1":
        (s, r, u), lo = tr.shard(s, r, u)
        s, r, u = s.contiguous(), r.contiguous(), u.contiguous()
    nosync = os.environ.get("NOSYNC") == "1"   # NOSYNC=1: enqueue every step without reading anything back in between
    res = []
    for _ in range(steps):
        t = tr.step(s, r, u, global_batch=B, row_offset=0) if os.environ.get("BENCHLIKE") == "1" else tr.step(s, r, u)
        res.append(t if nosync else [float(v) for v in t])
    torch.cuda.synchronize()
    out[mode] = [[float(v) for v in t] for t in res]
    print(mode, "capture_failed:", tr.capture_failed)
for i in range(steps):
    print(i, " | ".join(f"{m}: rec {out[m][i][1]:.6f} kld {out[m][i][2]:.4f}" for m in out))
