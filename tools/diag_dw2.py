import sys, torch
sys.path.insert(0, '.')
from pivotcvae_amd import ops
DEV='cuda:0'
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale
M,K,N=300,1419,256
x, g = rnd(M, K, seed=1), rnd(M, N, seed=8)
xd, gd = x.to(DEV), g.to(DEV)
ref = (g.double().t() @ x.double())
for mode in ("plain", "sync_after_zero", "double_launch_check", "fresh_inputs"):
    nbad = 0
    for it in range(10):
        dW, db = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
        if mode == "sync_after_zero": torch.cuda.synchronize()
        if mode == "fresh_inputs":
            xd, gd = x.to(DEV), g.to(DEV)
        ops.linear_bwd_weight_raw(gd, xd, dW, db)
        d = (dW.cpu().double() - ref).abs()
        nbad += int((d > 1e-2).any())
    print(mode, "bad iterations", nbad, "of 10")
