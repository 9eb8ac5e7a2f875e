#!/usr/bin/env python3
"""flake_gemm.py [iters]: the layer GEMMs of tests/test_hip_kernels.py::test_mlp_forward_backward, repeated, each result against fp64"""
import sys, torch
sys.path.insert(0, '.')
from pivotcvae_amd import ops
DEV = 'cuda:0'
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for (M, K, N) in [(7, 102, 24), (300, 1419, 256), (129, 283, 1152), (64, 16, 64)]:
    x, W, g = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.3), rnd(M, N, seed=8)
    xd, Wd, gd = x.to(DEV), W.to(DEV), g.to(DEV)
    ref_dW = (g.double().t() @ x.double()).float()
    ref_y = (x.double() @ W.double().t()).float()
    ref_dx = (g.double() @ W.double()).float()
    nb = [0, 0, 0]
    for it in range(iters):
        dW, db = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
        ops.linear_bwd_weight_raw(gd, xd, dW, db)
        y = ops.linear_fwd_raw(xd, Wd, None, 0)
        dx = ops.linear_bwd_input_raw(gd, Wd)
        for i, (got, ref) in enumerate(((dW, ref_dW), (y, ref_y), (dx, ref_dx))):
            d = (got.cpu() - ref).abs()
            if d.max() > 1e-2:
                nb[i] += 1
                if nb[i] <= 2:
                    idx = (d > 1e-2).nonzero()
                    print(f"  M={M} K={K} N={N} {'dW y dx'.split()[i]} iter {it}: {idx.shape[0]} bad, max {d.max():.3f}, rows {idx[:,0].min().item()}..{idx[:,0].max().item()} cols {idx[:,1].min().item()}..{idx[:,1].max().item()}")
    print(f"M={M} K={K} N={N}: bad dW/y/dx = {nb} of {iters}")
