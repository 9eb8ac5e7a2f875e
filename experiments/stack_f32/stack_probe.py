"""GPU: kernel time of pcvae_stack_fwd / pcvae_stack_bwd on one stack shape (hipGraph of 20 launches, per-launch average).
python tools/stack_probe.py M K0 N1,N2,.. [dx_cols]   (PCVAE_LIB selects a variant library)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pivotcvae_amd import ops   # noqa: E402


def timed(fn, reps=20, iters=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (iters * reps) * 1e6


if __name__ == "__main__":
    M, K0 = int(sys.argv[1]), int(sys.argv[2])
    widths = [int(w) for w in sys.argv[3].split(",")]
    dx = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    dev = "cuda:0"
    x = torch.randn(M, K0, device=dev)
    layers, K = [], K0
    for i, n in enumerate(widths):
        layers.append((torch.randn(n, K, device=dev) / K ** 0.5, torch.randn(n, device=dev), 1 if i < len(widths) - 1 else 0))
        K = n
    ys = ops.stack_fwd_raw([(x, layers)])[0]
    g = torch.randn(M, widths[-1], device=dev)
    f = timed(lambda: ops.stack_fwd_raw([(x, layers)], outs=[ys]))
    b = timed(lambda: ops.stack_bwd_raw([(x, layers)], [ys], [g], [dx]))
    print(f"{os.environ.get('PCVAE_LIB', 'product'):40s} M={M} K0={K0} {widths}: fwd {f:.1f} us, bwd {b:.1f} us", flush=True)
