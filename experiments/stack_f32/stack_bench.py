"""GPU: time the whole-stack kernels against the layer-by-layer GEMM launches on a model's stacks (kernel-side, hipGraph replay of
the forward + input-gradient chain of each stack).  python tools/stack_bench.py [config ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # noqa: E402
from pivotcvae_amd.train_generative import Trainer   # noqa: E402


def step_time(cfg, fused, B, iters=200):
    os.environ["PCVAE_STACK_FUSED"] = "1" if fused else "0"
    dev = torch.device("cuda:0")
    cfg = dict(cfg, B=B)
    model, _ = bench.build_model(cfg, dev, "f32")
    s, r, u = bench.synthetic_batch(cfg, B, dev)
    tr = Trainer(model, lr=bench.LR, beta=bench.BETA, capture_graph=True)
    for _ in range(5):
        tr.step(s, r, u)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        tr.step(s, r, u)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


if __name__ == "__main__":
    for name in (sys.argv[1:] or ["2", "3"]):
        cfg = bench.CONFIGS[name]
        for B in sorted({cfg["B"], 256, 1024, 2048, 4096}):
            a, b = step_time(cfg, True, B), step_time(cfg, False, B)
            print(f"config {name} B={B:5d}: step {a:.3f} ms with stack kernels (where eligible), {b:.3f} ms layer by layer", flush=True)
