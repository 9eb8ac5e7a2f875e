#!/usr/bin/env python3
"""gemm_x3_bench.py: the large MLP GEMMs of a config-4 step (M = 8192) in exact f32 MFMA and in bf16x3, 30 launches back to back
between one HIP event pair each (forward, input gradient, weight gradient through the grouped entry point)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pivotcvae_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
layers = [("enc_1", 256, 1419), ("enc_2", 256, 256), ("scm_1", 256, 283), ("scm_3", 1152, 256), ("prior_1", 128, 139), ("enc_hd", 32, 256)]
def timed(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = {False: 0.0, True: 0.0}
for name, N, K in layers:
    x, W, b = torch.rand(M, K, device=dev, generator=g) - 0.5, (torch.rand(N, K, device=dev, generator=g) - 0.5) * 0.1, torch.zeros(N, device=dev)
    gy = torch.rand(M, N, device=dev, generator=g) - 0.5
    dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    fl = 2.0 * M * N * K
    row = [f"{name:8s} N={N:5d} K={K:5d}"]
    for x3 in (False, True):
        with ops.mlp_arith(x3):
            t_f = timed(lambda: ops.linear_fwd_raw(x, W, b, 1))
            t_x = timed(lambda: ops.linear_bwd_input_raw(gy, W))
            t_w = timed(lambda: ops.linear_bwd_weight_raw(gy, x, dW, db))
        tot[x3] += t_f + t_x + t_w
        row.append(f"{'x3 ' if x3 else 'f32'} fwd {t_f:6.1f} us ({fl / t_f / 1e6 / 157.3:4.2f})  dX {t_x:6.1f} ({fl / t_x / 1e6 / 157.3:4.2f})  dW {t_w:6.1f} ({fl / t_w / 1e6 / 157.3:4.2f})")
    print(" | ".join(row))
print("sum f32 %.1f us, x3 %.1f us" % (tot[False], tot[True]))
