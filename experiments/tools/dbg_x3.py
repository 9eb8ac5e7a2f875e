"""dbg_x3.py: the bf16x3 catalog kernel against the C oracle on one shape, per output (lse / nll / dx) and per 16-row column tile -
the script the queued-operand hazard was bisected with (HISTORY.md 3.1b)."""
import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from pivotcvae_amd import ops
from pivotcvae_amd._hip import PREC_BF16X3
from oracle import pivotcvae_oracle as orc
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale
R, N, D = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 224, 128
rx, E = rnd(R, D, seed=1, scale=2.0), orc.normalize_rows(rnd(N, D, seed=2))
tgt = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(3))
nll, lse, dx = ops.catalog_ce_raw(rx.cuda(), ops.CatalogTable(E.cuda()), tgt.cuda(), prec=PREC_BF16X3)
from oracle import catalog_oracle as co
wn, wl, wd = co.ce(rx.numpy(), E.numpy(), tgt.numpy())
l = lse.cpu().numpy(); d = dx.cpu().numpy()
bad = ~np.isfinite(l)
print("N", N, "nan rows:", bad.sum(), "of", R, "first bad rows", np.nonzero(bad)[0][:20])
err = np.abs(l - wl); err[bad] = 0
print("max lse err (finite rows)", err.max(), "rows with err>1e-4:", np.nonzero(err > 1e-4)[0][:40])
de = np.abs(d - wd); de[~np.isfinite(de)] = 0
print("max dx err", de.max(), "bad dx rows", np.nonzero(de.max(1) > 1e-4)[0][:40], "bad dx cols", np.nonzero(de.max(0) > 1e-4)[0][:40])
dn = ~np.isfinite(d).all(1)
print("dx nan rows:", dn.sum(), np.nonzero(dn)[0][:24])
print("lse bad sample", l[:4], "ref", wl[:4])
ok = np.isfinite(d).all(1)
print("dx err on finite rows", (np.abs(d - wd)[ok]).max() if ok.any() else None)
ws = list(ops._ws_cache.values())[0]
f = ws.view(torch.float32)
ns = 1
pm = f[:ns * R].cpu().numpy(); pl_ = f[ns * R:2 * ns * R].cpu().numpy(); pU = f[2 * ns * R:2 * ns * R + ns * R * D].cpu().numpy().reshape(R, D)
print("pl rows 0..3", pl_[:4], "rows 16..19", pl_[16:20])
print("pU nan per row (first 36):", np.isnan(pU).sum(1)[:36])
print("pU row0 nan cols", np.nonzero(np.isnan(pU[0]))[0][:40])
