"""Diagnostic: run the STAMPS build of the fast kernel once and print per-phase cycle shares per wave."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pivotcvae_amd import ops, _hip
R, N, D = 81920, 1_000_000, 128
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
E = torch.rand(N, D, device=dev, generator=g) * 2 - 1
E = E / E.norm(dim=1, keepdim=True)
rx = (torch.rand(R, D, device=dev, generator=g) * 2 - 1) * 0.3
tgt = torch.randint(0, N, (R,), device=dev, generator=g)
table = ops.CatalogTable(E)
for _ in range(2):
    ops.catalog_ce_raw(rx, table, tgt, prec=_hip.PREC_BF16)
torch.cuda.synchronize()
ws = ops._ws_cache[rx.device]
f = ws.view(torch.float32)
nsplit = 4  # plan for this shape
pU = f[2 * nsplit * R: 2 * nsplit * R + nsplit * R * D].view(nsplit, R, D)
names = ["S-chain issue", "tr-issue+softmax", "U-chain (incl. tr wait)", "seam wait+barrier"]
tot = 0
vals = []
for k in range(4):
    a = pU[:, 32 * k::256, 0].float().mean().item()       # waves 0..3 carry tm[k]
    b = pU[:, 32 * (k + 4)::256, 1].float().mean().item()  # waves 4..7 carry tm[k]
    vals.append((a, b))
tot_a = sum(v[0] for v in vals); tot_b = sum(v[1] for v in vals)
subtiles = (N // nsplit) / 32
for k, (a, b) in enumerate(vals):
    print(f"{names[k]:28s} waves0-3: {a / subtiles:8.1f} cyc/subtile ({100 * a / tot_a:4.1f}%)   waves4-7: {b / subtiles:8.1f} ({100 * b / tot_b:4.1f}%)")
print("sum per subtile", tot_a / subtiles, tot_b / subtiles)
