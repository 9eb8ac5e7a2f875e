import torch, sys
sys.path.insert(0, '.')
from pivotcvae_amd import ops
torch.manual_seed(0)
for (M,N,K) in [(129,48,48),(300,256,1419),(8192,256,1419)]:
    gy = (torch.rand(M,N)*2-1); x = (torch.rand(M,K)*2-1)
    ref64 = gy.double().t() @ x.double()
    cpu32 = gy.t() @ x
    dW = torch.zeros(N,K,device='cuda'); db = torch.zeros(N,device='cuda')
    ops.linear_bwd_weight_raw(gy.cuda(), x.cuda(), dW, db)
    e_ours = (dW.cpu().double()-ref64).abs().max().item(); e_cpu = (cpu32.double()-ref64).abs().max().item()
    print('dW', M,N,K, 'ours max abs err', e_ours, 'cpu32', e_cpu, 'db err', (db.cpu().double()-gy.double().sum(0)).abs().max().item())
    W = (torch.rand(N,K)*2-1)
    y = ops.linear_fwd_raw(x.cuda(), W.cuda(), None, 0)
    r64 = x.double() @ W.double().t()
    print('fwd', (y.cpu().double()-r64).abs().max().item(), ((x@W.t()).double()-r64).abs().max().item())
    dx = ops.linear_bwd_input_raw(gy.cuda(), W.cuda())
    r64 = gy.double() @ W.double()
    print('dX', (dx.cpu().double()-r64).abs().max().item(), ((gy@W).double()-r64).abs().max().item())
