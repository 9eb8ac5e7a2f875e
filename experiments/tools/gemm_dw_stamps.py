"""GPU, probe build (-DGEMM_STAMPS): a weight-gradient launch of config-2 size - stamps of workgroup 0 (K loop, partial stores
acknowledged, arrival) and of tile 0's last arriver (reduction loads, final add), and the launch's duration in a graph.
PCVAE_LIB=build/variants/gemm_STAMPS.so python tools/gemm_dw_stamps.py M K N"""
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
g = torch.randn(M, N, device=dev)
dW = torch.zeros(N, K, device=dev)
db = torch.zeros(N, device=dev)


def run():
    grp = ops.GemmGroup()
    grp.dw(g, x, dW, db)
    grp.launch()


for _ in range(3):
    run()
torch.cuda.synchronize()
if os.environ.get("PCVAE_LIB"):
    buf = (ctypes.c_ulonglong * 64)()
    fn = lib().pcvae_gemm_stamps
    fn.argtypes = [ctypes.c_void_p]
    assert fn(buf) == 0
    t = list(buf)
    print("workgroup 0: K loop %d, partials acknowledged +%d, arrival +%d ticks" % (t[1] - t[0], t[2] - t[1], t[3] - t[2]))
    print("last arriver of tile 0: reduction loads %d, bias + final add %d ticks" % (t[33] - t[32], t[34] - t[33]))
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(20):
        run()
gr.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    gr.replay()
torch.cuda.synchronize()
print(f"dW [{N} x {K}] over {M} rows: {(time.perf_counter() - t0) / 400 * 1e6:.2f} us per launch in a graph")
