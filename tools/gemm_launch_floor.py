#!/usr/bin/env python3
"""gemm_launch_floor.py: what a forward GEMM launch of the MLP stacks costs as a function of its K loop, WITHOUT the host: 20 dependent
launches captured in a hipGraph, replayed 20 times -> us per launch, for [8192 x K] . [256 x K]^T at K = 32 .. 2048 (1 .. 64 chunks of
32) in each arithmetic; intercept = the launch's fixed cost (dispatch of 512 workgroups, parameter load, first global -> LDS round
trip, epilogue, drain), slope = the K loop per chunk.  Beside it: the cheapest possible node (a 16-byte fill) the same way.

    python tools/gemm_launch_floor.py > profiles/r06_gemm_launch_floor.txt"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pivotcvae_amd import ops  # noqa: E402

dev = "cuda:0"
M, N = 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 256


def graph_time(fn, reps=20, inner=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * inner) * 1e6


tiny = torch.zeros(4, device=dev)
print(f"# M = {M}, N = {N}; us per launch inside a replayed hipGraph of 20 dependent launches")
print(f"16-byte fill kernel (ops.zero_): {graph_time(lambda: ops.zero_(tiny)):.2f} us per launch")
Ks = [32, 64, 128, 256, 512, 1024, 2048]
rows = {}
for arith in ("f32", "bf16x3", "bf16x6"):
    ts = []
    for K in Ks:
        x = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev)
        with ops.mlp_arith(arith):
            os.environ["PCVAE_GEMM_SMALL_BELOW"] = "0"      # the 64 x 64 DMA tiles at every size
            ts.append(graph_time(lambda: ops.linear_fwd_raw(x, W, b, 1, out=y)))
    rows[arith] = ts
    # least squares over chunks
    ch = [k // 32 for k in Ks]
    n = len(ch)
    sx, sy, sxx, sxy = sum(ch), sum(ts), sum(c * c for c in ch), sum(c * t for c, t in zip(ch, ts))
    slope = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    icpt = (sy - slope * sx) / n
    print(f"{arith:7s} " + "  ".join(f"K={k}: {t:6.2f}" for k, t in zip(Ks, ts)) + f"   fit: {icpt:.2f} us + {slope:.3f} us per 32-deep chunk")
