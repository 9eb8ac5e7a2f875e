"""graph_nodes_run.py [variant]: a few hipGraph-replayed steps of a mid-size model (to be run under rocprofv3 --kernel-trace --stats): the
kernel list shows whether any runtime blit (__amd_rocclr_fillBuffer* / copyBuffer* = a memset / memcpy NODE) sits in the captured step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pivotcvae_amd as pa
from pivotcvae_amd.train_generative import Trainer
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import pivotcvae_oracle as orc
variant = sys.argv[1] if len(sys.argv) > 1 else "pivotcvae_pt_pi"
S, D, Z, N, NU, B, H, HP = 5, 128, 16, 50000, 40, 512, 64, 32
C = S + 1
e_raw, u_raw = orc.synthetic_tables(N, NU, D, seed=0)
st = dict(enc=[S * D + C + D, H, H], psm=[Z + C + D, H, H, D], scm=[Z + C + 2 * D, H, H, (S - 1) * D], prior=[C + D, HP, HP])
torch.manual_seed(0)
m = pa.PIVOTCVAE_MODELS[variant](torch.nn.Embedding.from_pretrained(e_raw), torch.nn.Embedding.from_pretrained(u_raw), S, D, Z, C,
                                 st["enc"], st["psm"], st["scm"], st["prior"], False, "cuda:0")
g = torch.Generator().manual_seed(1)
s = torch.randint(0, N, (B, S), generator=g).cuda(); u = torch.randint(0, NU, (B, 1), generator=g).cuda()
r = (torch.rand(B, S, generator=g) < 0.5).float().cuda()
tr = Trainer(m, lr=1e-3, beta=0.001, capture_graph=True)
for _ in range(6):
    out = tr.step(s, r, u)
torch.cuda.synchronize()
print("captured:", tr._graph is not None, "failed:", tr.capture_failed, "kld", float(out[2]))
