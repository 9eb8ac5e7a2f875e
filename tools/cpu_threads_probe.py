"""How does the CPU oracle's train step scale with torch threads on this host? (picks the bench baseline's thread count)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pivotcvae_oracle as orc
import bench
cfg = bench.CONFIGS["4"]; st = bench.structs(cfg["S"], cfg["D"])
torch.manual_seed(0)
N, S, D = cfg["N"], cfg["S"], cfg["D"]
e, u = orc.synthetic_tables(N, bench.N_USER, D)
sd = {"docEmbed.weight": orc.normalize_rows(e), "userEmbed.weight": orc.normalize_rows(u)}
def lin(name, i, o):
    sd[name + ".weight"] = torch.randn(o, i) * 0.05; sd[name + ".bias"] = torch.zeros(o)
for pre in ("enc", "psm", "scm", "prior"):
    s_ = st[pre]
    for i in range(len(s_) - 1): lin(f"{pre}_{i+1}", s_[i], s_[i+1])
for h, w in (("encmu", 256), ("enclogvar", 256), ("priorMu", 128), ("priorLogvar", 128)): lin(h, w, 16)
ocfg = orc.Config("pivotcvae_gt_pi", S, D, 16, False, st)
B = 8
g = torch.Generator().manual_seed(1)
s = torch.randint(0, N, (B, S), generator=g); uu = torch.randint(0, bench.N_USER, (B, 1), generator=g)
r = (torch.rand(B, S, generator=g) < 0.5).float(); eps = torch.randn(B, 16, generator=g)
print("cpu_count", os.cpu_count())
for th in (8, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1): break
    torch.set_num_threads(th)
    orc.loss_and_grads(sd, ocfg, s, r, uu, eps, 0.001)
    t0 = time.perf_counter(); orc.loss_and_grads(sd, ocfg, s, r, uu, eps, 0.001); dt = time.perf_counter() - t0
    print(f"threads {th}: {dt:.2f} s/step", flush=True)
