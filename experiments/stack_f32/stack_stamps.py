"""GPU, probe build (-DSTACK_STAMPS): where a workgroup of pcvae_stack_fwd spends its time (shader-clock stamps of wave 0 of workgroup 0).
PCVAE_LIB=build/variants/stack_STAMPS.so python tools/stack_stamps.py M K0 N1,N2,.."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pivotcvae_amd import ops   # noqa: E402
from pivotcvae_amd._hip import lib   # noqa: E402

M, K0 = int(sys.argv[1]), int(sys.argv[2])
widths = [int(w) for w in sys.argv[3].split(",")]
dev = "cuda:0"
x = torch.randn(M, K0, device=dev)
layers, K = [], K0
for i, n in enumerate(widths):
    layers.append((torch.randn(n, K, device=dev) / K ** 0.5, torch.randn(n, device=dev), 1 if i < len(widths) - 1 else 0))
    K = n
for _ in range(3):
    ys = ops.stack_fwd_raw([(x, layers)])[0]
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
fn = lib().pcvae_stack_stamps
fn.argtypes = [ctypes.c_void_p]
assert fn(buf) == 0
t = list(buf)
names = ["start", "tile_in issued+waited", "barrier"] + [f"L{l} {w}" for l in range(len(widths)) for w in ("begin", "computed", "barrier", "stored")]
for i in range(1, 3 + 4 * len(widths)):
    print(f"{names[i]:28s} +{t[i] - t[i - 1]:8d} ticks   (total {t[i] - t[0]})")
