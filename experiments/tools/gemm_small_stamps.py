"""GPU, probe build (-DGEMM_STAMPS): where a workgroup of the small-tile GEMM spends its time, and the launch's duration in a graph.
PCVAE_LIB=build/variants/gemm_STAMPS.so python tools/gemm_small_stamps.py M K N"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pivotcvae_amd import ops   # noqa: E402
from pivotcvae_amd._hip import lib   # noqa: E402

M, K, N = (int(v) for v in sys.argv[1:4])
dev = "cuda:0"
x = torch.randn(M, K, device=dev)
W = torch.randn(N, K, device=dev) / K ** 0.5
b = torch.randn(N, device=dev)
y = torch.empty(M, N, device=dev)
for _ in range(3):
    ops.linear_fwd_raw(x, W, b, 1, out=y)
torch.cuda.synchronize()
if os.environ.get("PCVAE_LIB"):
    buf = (ctypes.c_ulonglong * 64)()
    fn = lib().pcvae_gemm_stamps
    fn.argtypes = [ctypes.c_void_p]
    assert fn(buf) == 0
    t = list(buf)
    n = 1 + (K + 63) // 64 + 2
    print("stamps (ticks since start):", [t[i] - t[0] for i in range(1, n)])
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20):
        ops.linear_fwd_raw(x, W, b, 1, out=y)
g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print(f"[{M} x {K}] . [{N} x {K}]^T: {(time.perf_counter() - t0) / 400 * 1e6:.2f} us per launch in a graph")
