"""dbg_x3_shapes.py [D]: the bf16x3 kernel against the C oracle over the shapes of tests/test_hip_x3.py, one line per shape (which
fill / drain / tail path is off)."""
import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np
from pivotcvae_amd import ops
from pivotcvae_amd._hip import PREC_BF16X3
from oracle import catalog_oracle as co, pivotcvae_oracle as orc
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed); return (torch.rand(*shape, generator=g) * 2 - 1) * scale
D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for R, N in ((128, 224), (128, 256), (256, 448)):
    rx, E = rnd(R, D, seed=1, scale=2.0), orc.normalize_rows(rnd(N, D, seed=2))
    tgt = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(3))
    nll, lse, dx = ops.catalog_ce_raw(rx.cuda(), ops.CatalogTable(E.cuda()), tgt.cuda(), prec=PREC_BF16X3)
    wn, wl, wd = co.ce(rx.numpy(), E.numpy(), tgt.numpy())
    bad = (~np.isclose(lse.cpu().numpy(), wl, rtol=2e-6, atol=2e-6)).nonzero()[0]
    print(D, R, N, 'bad lse rows', list(map(int, bad)))
    if len(bad):
        b = int(bad[0]); print(' row', b, 'lse', float(lse[b]), 'want', wl[b], 'exp diff', float(np.exp(lse[b].item()) - np.exp(wl[b])), ' sum exp of last tile logits', float(np.exp((rx[b] @ E[N-32:].T)).sum()), 'first tile', float(np.exp((rx[b] @ E[:32].T)).sum()))
