"""dbg_x3_big.py [D] [R] [N]: bf16x3 lse at N = 10M against a chunked fp64 softmax on the device (long fp32 accumulation chains,
HISTORY.md 3.1b: the range cap of catalog_plan)."""
import sys; sys.path.insert(0,'/root/repo')
import torch
from pivotcvae_amd import ops, _hip
from pivotcvae_amd._hip import PREC_BF16X3
D = int(sys.argv[1]) if len(sys.argv) > 1 else 256
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
N = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
dev = "cuda:0"
def ref(rx, E, chunk=125000):
    R = rx.shape[0]
    m = torch.full((R,), -float("inf"), device=dev, dtype=torch.float64); s = torch.zeros(R, device=dev, dtype=torch.float64)
    for c0 in range(0, E.shape[0], chunk):
        lg = rx.double() @ E[c0:c0 + chunk].double().t()
        mn = torch.maximum(m, lg.max(1)[0]); s = s * torch.exp(m - mn) + torch.exp(lg - mn[:, None]).sum(1); m = mn
    return m + torch.log(s)
g = torch.Generator(device=dev).manual_seed(5)
E = torch.rand(N, D, device=dev, generator=g) * 2 - 1; E = E / E.norm(dim=1, keepdim=True)
for scale in (1.5, 6.0):
    rx = (torch.rand(R, D, device=dev, generator=g) * 2 - 1) * scale
    tgt = torch.randint(0, N, (R,), device=dev, generator=g)
    nll, lse, dx = ops.catalog_ce_raw(rx, ops.CatalogTable(E), tgt, prec=PREC_BF16X3)
    pick = torch.arange(0, R, max(1, R // 256), device=dev)
    want = ref(rx[pick], E)
    err = (lse[pick].double() - want)
    L = _hip.lib()
    print('D', D, 'R', R, 'N', N, 'scale', scale, 'lse mean', float(want.mean()), 'err mean', float(err.mean()), 'abs max', float(err.abs().max()), 'std', float(err.std()),
          'variant', L.pcvae_catalog_ce_variant(R, N, D, 2))
    big = pick[err.abs() > 3e-5]
    print('  rows with |err|>3e-5:', big[:24].tolist(), ' mod 16:', sorted(set((big % 16).tolist())))
