#!/usr/bin/env python3
"""mlp_pmc_run.py: the program tools/mlp_pmc.sh profiles - the largest MLP GEMM of a config-4 step (enc_1 forward: [8192 x 1419] . [256 x 1419]^T,
45 chunks of the K loop per workgroup, 512 workgroups) launched 6 times per arithmetic; the kernel's template argument (0 f32, 1 bf16x3,
2 bf16x6) separates the three in the counter files."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pivotcvae_amd import ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
M, N, K = 8192, 256, 1419
x, W, b = torch.rand(M, K, device=dev, generator=g) - 0.5, (torch.rand(N, K, device=dev, generator=g) - 0.5) * 0.1, torch.zeros(N, device=dev)
for arith in ops.MLP_PRECISIONS:
    with ops.mlp_arith(arith):
        for _ in range(6):
            y = ops.linear_fwd_raw(x, W, b, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.linear_fwd_raw(x, W, b, 1)
        e1.record()
        torch.cuda.synchronize()
        print(f"{arith:7s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per launch (back to back), checksum {float(y.double().sum()):.6f}")
