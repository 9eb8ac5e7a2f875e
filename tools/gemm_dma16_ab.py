#!/usr/bin/env python3
"""gemm_dma16_ab.py [M]: every large MLP GEMM of a config-4 step, per arithmetic (f32 / bf16x3 / bf16x6) and kind (fwd / dX / dW): time of 30
back-to-back launches and a bit checksum of the result.  Run once with PCVAE_GEMM_DMA16=0 and once with =1 (the switch is read
once per process): times compare the 4-byte and the 16-byte LDS-DMA fill of the same LDS images, checksums must be IDENTICAL."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pivotcvae_amd import ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
layers = [("enc_1", 256, 1419), ("enc_2", 256, 256), ("scm_1", 256, 283), ("scm_3", 1152, 256), ("prior_1", 128, 139), ("enc_hd", 32, 256),
          ("ragged", 130, 97)]


def timed(fn, n=30):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def bits(t):
    return int(t.contiguous().view(torch.int32).to(torch.int64).sum().item()) & 0xffffffffffff


print(f"# PCVAE_GEMM_DMA16={os.environ.get('PCVAE_GEMM_DMA16', '(default 1)')}  M={M}")
tot = {}
for name, N, K in layers:
    Mx = M if name != "ragged" else 1001
    x, W, b = torch.rand(Mx, K, device=dev, generator=g) - 0.5, (torch.rand(N, K, device=dev, generator=g) - 0.5) * 0.1, torch.zeros(N, device=dev)
    gy = torch.rand(Mx, N, device=dev, generator=g) - 0.5
    for arith in ("f32", "bf16x3", "bf16x6"):
        with ops.mlp_arith(arith):
            y = ops.linear_fwd_raw(x, W, b, 1)
            dx = ops.linear_bwd_input_raw(gy, W)
            dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
            ops.linear_bwd_weight_raw(gy, x, dW, db)
            cs = (bits(y), bits(dx), bits(dW), bits(db))
            t_f = timed(lambda: ops.linear_fwd_raw(x, W, b, 1))
            t_x = timed(lambda: ops.linear_bwd_input_raw(gy, W))
            t_w = timed(lambda: ops.linear_bwd_weight_raw(gy, x, dW, db))
        tot[arith] = tot.get(arith, 0.0) + t_f + t_x + t_w
        print(f"{name:8s} N={N:5d} K={K:5d} {arith:7s} fwd {t_f:6.1f} us  dX {t_x:6.1f}  dW {t_w:6.1f}   checksums {cs[0]:012x} {cs[1]:012x} {cs[2]:012x} {cs[3]:012x}")
print("sums: " + "  ".join(f"{k} {v:.1f} us" for k, v in tot.items()))
