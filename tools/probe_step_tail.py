import sys, time; sys.path.insert(0,'/root/repo')
import torch, bench
from pivotcvae_amd.train_generative import Trainer
from pivotcvae_amd import ops
dev=torch.device('cuda',0)
for c in ('2','3'):
    cfg=bench.CONFIGS[c]
    model,_=bench.build_model(cfg,dev,'f32' if c=='2' else 'bf16'); model.set_mlp_precision('f32' if c=='2' else 'bf16x3')
    tr=Trainer(model,lr=3e-4,beta=0.001,capture_graph=True)
    s,r,u=bench.synthetic_batch(cfg,cfg['B'],dev)
    for _ in range(5): tr.step(s,r,u)
    def run(n, full=True):
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(n):
            if full: tr.step(s,r,u)
            else:
                st=tr._static
                ops.philox_normal_(st["eps"], seed=0, offset=0); tr._graph.replay()
        torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
    print('config',c,'full step us',round(run(200),1),'graph+philox only',round(run(200,False),1),'graph only', end=' ')
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(200): tr._graph.replay()
    torch.cuda.synchronize(); print(round((time.perf_counter()-t0)/200*1e6,1))
