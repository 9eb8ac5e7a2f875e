"""dbg_graph_race.py [B] [steps] [runs]: the last step's KLD of `steps` hipGraph-replayed steps whose results are DROPPED at once (as a
benchmark loop does), several fresh runs, against the eager value."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pivotcvae_amd.train_generative import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS["4"], B=B)
for mode in ["eager"] + ["graph"] * runs:
    model, _ = bench.build_model(cfg, dev, "bf16x3")
    model.set_mlp_precision("bf16x3")
    tr = Trainer(model, lr=bench.LR, beta=bench.BETA, capture_graph=mode != "eager")
    s, r, u = bench.synthetic_batch(cfg, B, dev)
    alt = [bench.synthetic_batch(cfg, B, dev, seed=50 + i) for i in range(3)] if os.environ.get("ALT") == "1" else None
    flow = os.environ.get("BENCHFLOW", "")
    if flow:   # pieces of bench.py's StepTimer flow: s = shard + contiguous, k = global_batch / row_offset keywords, y = synchronize after 2 steps
        lo = 0
        if "s" in flow:
            (s, r, u), lo = tr.shard(s, r, u)
            s, r, u = s.contiguous(), r.contiguous(), u.contiguous()
        kw = dict(global_batch=B, row_offset=lo) if "k" in flow else {}
        for i in range(2):
            if alt: s, r, u = alt[i % 3]
            tr.step(s, r, u, **kw)
        if "y" in flow:
            torch.cuda.synchronize()
        for i in range(2, steps):
            if alt: s, r, u = alt[i % 3]   # a different batch every step: the graph's input buffers are refilled by eager copies
            loss, rec, kld = tr.step(s, r, u, **kw)
    else:
        for _ in range(steps - 1):
            tr.step(s, r, u)          # result dropped
        loss, rec, kld = tr.step(s, r, u)
    torch.cuda.synchronize()
    print(mode, f"rec {float(rec):.6f} kld {float(kld):.4f}", "checksum of the parameters", float(tr.opt.flat.double().sum()), flush=True)
    del tr, model
