#!/usr/bin/env python3
"""Candidate statistics of the screened argmax (bench_argmax.py inputs): items per row within 2 eps of the pass-A threshold (what pass B
parks when thresholds never rise: ~26 per row) and with a row-wide running threshold (~5); the kernel, with per-lane thresholds that restart
in every catalog range, sits in between.  Each candidate costs its workgroup ~650 cycles (the wave that parks it arrives late at the next barrier)."""
import torch
dev="cuda:0"; D=128; N=1_000_000; R=256
g=torch.Generator(device=dev).manual_seed(0)
E=torch.rand(N,D,device=dev,generator=g)*2-1; E=E/E.norm(dim=1,keepdim=True)
x=(torch.rand(R,D,device=dev,generator=g)*2-1)
s=(x.bfloat16().float() @ E.bfloat16().float().t())
eps=(0.00390625*1.02+2e-5)*x.norm(dim=1)*1.0
pm=s[:, :N//16].max(dim=1).values
thr=pm-2*eps
cnt=(s>=thr[:,None]).sum(dim=1).float()
print("rows",R,"mean candidates over fixed passA threshold", cnt.mean().item(), "max", cnt.max().item(), "eps", eps.mean().item(), "xnorm", x.norm(dim=1).mean().item())
# with running per-row threshold (row-shared): sequential simulate on CPU for a few rows
s_c=s[:8].cpu(); thr_c=thr[:8].cpu(); e2=(2*eps[:8]).cpu()
tot=0
for r in range(8):
    t=thr_c[r].item(); c=0
    row=s_c[r]
    # process in tiles of 32
    for i in range(0,N,32):
        blk=row[i:i+32]; m=blk.max().item()
        if m>=t:
            c+=int((blk>=t).sum()); t=max(t,m-e2[r].item())
    tot+=c
print("row-shared running threshold: mean candidates", tot/8)
