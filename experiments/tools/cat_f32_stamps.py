"""GPU, probe build (-DCAT_STAMPS): the exact-f32 catalog kernel at a small size - 100 MHz wall-clock stamps of the first and the
last workgroup of the launch (start, prologue done, catalog range done, partials stored), relative to the first workgroup's start.
PCVAE_LIB=build/variants/cat_STAMPS.so python tools/cat_f32_stamps.py R N D"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pivotcvae_amd import ops   # noqa: E402
from pivotcvae_amd._hip import lib   # noqa: E402

R, N, D = (int(v) for v in sys.argv[1:4])
dev = "cuda:0"
E = torch.randn(N, D, device=dev)
E = E / E.norm(dim=1, keepdim=True)
rx = torch.randn(R, D, device=dev) * 0.3
tgt = torch.randint(0, N, (R,), device=dev)
table = ops.CatalogTable(E)
for _ in range(3):
    ops.catalog_ce_raw(rx, table, tgt)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
fn = lib().pcvae_cat_stamps
fn.argtypes = [ctypes.c_void_p]
assert fn(buf) == 0
t = list(buf)
for name, o in (("first workgroup", 0), ("last workgroup", 32)):
    print(name, "us since the first workgroup's start:", [round((t[o + i] - t[0]) / 100.0, 2) for i in range(4)])
