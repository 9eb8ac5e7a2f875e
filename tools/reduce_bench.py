import torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pivotcvae_amd import ops
dev="cuda:0"
ts=[torch.randn(8192,16,device=dev)*0.3 for _ in range(4)]
x=torch.randn(81920,device=dev)
def t(f,n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
print("kld_fwd us", t(lambda: ops.kld(*ts)))
