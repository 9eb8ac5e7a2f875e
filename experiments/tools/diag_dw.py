"""diag_dw.py M K N: the weight gradient of one layer against fp64, three launches; for wrong elements: which 32-row chunk or 64-row
batch split is missing or doubled, which tiles and which wave quadrants / accumulator rows (used to track down the lost fp32
atomic updates between XCDs: HISTORY.md 3.1d)."""
import sys, torch
sys.path.insert(0, '.')
from pivotcvae_amd import ops
DEV='cuda:0'
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale
M,K,N=int(sys.argv[1]),int(sys.argv[2]),int(sys.argv[3])
x, g = rnd(M, K, seed=1), rnd(M, N, seed=8)
xd, gd = x.to(DEV), g.to(DEV)
ref = (g.double().t() @ x.double())
parts = [(g[i:i+32].double().t() @ x[i:i+32].double()) for i in range(0, M, 32)]
parts64 = [(g[i:i+64].double().t() @ x[i:i+64].double()) for i in range(0, M, 64)]
for it in range(3):
    dW, db = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
    ops.linear_bwd_weight_raw(gd, xd, dW, db)
    d = dW.cpu().double() - ref
    bad = d.abs() > 1e-2
    if bad.any():
        idx = bad.nonzero()
        cls = {}
        for (r, c) in idx[:3000].tolist():
            e = d[r, c].item(); lab = "other"
            for s, p in enumerate(parts):
                if abs(e + p[r, c].item()) < 1e-3: lab = f"missing_chunk{s}"
                if abs(e - p[r, c].item()) < 1e-3: lab = f"double_chunk{s}"
            for s, p in enumerate(parts64):
                if abs(e + p[r, c].item()) < 1e-3: lab = f"missing_split{s}"
                if abs(e - p[r, c].item()) < 1e-3: lab = f"double_split{s}"
            cls[lab] = cls.get(lab, 0) + 1
        tiles = {}
        for r, c in idx.tolist():
            key = (r // 64, c // 64); tiles[key] = tiles.get(key, 0) + 1
        sub = {}
        for r, c in idx.tolist():
            key = ((r % 64) // 32, (c % 64) // 32); sub[key] = sub.get(key, 0) + 1
        print("iter", it, "bad", idx.shape[0], cls)
        print("   tiles (by,bx):count", sorted(tiles.items())[:40])
        rows_by_q = {}
        for r, c in idx.tolist():
            rows_by_q.setdefault(((r % 64) // 32, (c % 64) // 32), {}).setdefault(r % 32, []).append(c % 32)
        for q in sorted(rows_by_q):
            print("   quadrant", q, {row: len(cols) for row, cols in sorted(rows_by_q[q].items())})
        print("   wave quadrant (wm,wn):count", sorted(sub.items()), " rows-in-tile range", (idx[:,0] % 64).min().item(), (idx[:,0] % 64).max().item(), "cols-in-tile", (idx[:,1] % 64).min().item(), (idx[:,1] % 64).max().item())
    else:
        print("iter", it, "ok")
