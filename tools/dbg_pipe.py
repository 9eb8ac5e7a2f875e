import sys, torch
sys.path.insert(0, '.')
from tests.test_hip_bf16 import emulate_fast, rnd
from oracle import pivotcvae_oracle as orc
from pivotcvae_amd import ops
from pivotcvae_amd._hip import PREC_BF16
for R, N, D in [(128, 32, 256), (128, 64, 256), (128, 96, 256), (128, 128, 256), (128, 160, 256), (128, 256, 256), (128, 320, 256), (128, 352, 256), (128, 512, 256)]:
    rx, E = rnd(R, D, seed=1, scale=2.0 * (128.0 / D) ** 0.5), orc.normalize_rows(rnd(N, D, seed=2))
    tgt = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(3))
    nll, lse, dx = ops.catalog_ce_raw(rx.cuda(), E.cuda(), tgt.cuda(), prec=PREC_BF16)
    wn, wl, wd = emulate_fast(rx, E, tgt)
    el = (lse.cpu() - wl)
    print(R, N, D, "lse err max", el.abs().max().item(), "mean", el.mean().item(), "expected if one tile missing", float(torch.log(torch.tensor(1 - 32.0 / N))),
          "dx err", (dx.cpu() - wd).abs().max().item(), "dx scale", wd.abs().max().item())
    bad = el.abs() > 1e-4
    print("   bad rows:", bad.nonzero().flatten()[:20].tolist(), "count", int(bad.sum()))
