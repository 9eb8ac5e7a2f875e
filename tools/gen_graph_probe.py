#!/usr/bin/env python3
"""gen_graph_probe.py [config]: greedy generation (recommend(return_item=True)) launched eagerly against the same chain replayed as a
hipGraph (eps given, static inputs): what the ~33 small launches of a batch cost beyond their kernels."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "3"])
model, _ = bench.build_model(cfg, dev, "bf16")
B, S = cfg["B"], cfg["S"]
g = torch.Generator(device=dev).manual_seed(7)
u = torch.randint(0, bench.N_USER, (B, 1), device=dev, generator=g)
ctx = (torch.rand(B, S, device=dev, generator=g) < 0.5).float()
eps = torch.randn(B, bench.Z, device=dev, generator=g)


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    items_e, _ = model.recommend(ctx, u, return_item=True, eps=eps)
    t_eager = timed(lambda: model.recommend(ctx, u, return_item=True, eps=eps))
    from pivotcvae_amd import ops
    warm = {}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), ops.workspace_holder(warm):
        for _ in range(2):
            model.recommend(ctx, u, return_item=True, eps=eps)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr), ops.workspace_holder(warm):
        items_g, mu_g = model.recommend(ctx, u, return_item=True, eps=eps)
    gr.replay()
    torch.cuda.synchronize()
    t_graph = timed(gr.replay)
print(f"config {sys.argv[1] if len(sys.argv) > 1 else '3'}: eager {t_eager:.3f} ms per batch, hipGraph replay {t_graph:.3f} ms ({t_eager / t_graph:.3f}x), "
      f"ids identical: {torch.equal(items_e, items_g)}")

# the in-loop evaluation (5 contexts x trials of recommend + click model + statistics), eager loop against hipGraph replay
from pivotcvae_amd.env.response_model import UserResponseModel_MLP  # noqa: E402
from pivotcvae_amd.train_generative import recommendation_test  # noqa: E402

D = cfg["D"]
torch.manual_seed(5)
resp = UserResponseModel_MLP(8, bench.N_USER - 1, D, S, [(S + 1) * D, 256, 256, S], dev, False)
resp.docEmbed = model.docEmbed
resp.maxItemId = cfg["N"] - 1
resp = resp.to(dev)
bs, trials = min(B, 1024), 4
res = {}
for graph in (False, True):
    recommendation_test(model, resp, bs, n_test_trial=1, capture_graph=graph)
    torch.cuda.synchronize()
    model._rng_offset = 0
    t0 = time.perf_counter()
    res[graph] = recommendation_test(model, resp, bs, n_test_trial=trials, seed=3, capture_graph=graph)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"recommendation_test, bs {bs}, {trials} trials x 5 contexts, {'hipGraph replay' if graph else 'eager loop'}: {dt * 1e3:.2f} ms "
          f"= {trials * 5 * bs / dt / 1e3:.0f} K slates/s generated and scored")
print("statistics identical:", torch.equal(res[False], res[True]))
