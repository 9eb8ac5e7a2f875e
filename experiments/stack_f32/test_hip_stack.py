"""-m gpu: the whole-stack kernels (csrc/stack_f32.hip: pcvae_stack_fwd / pcvae_stack_bwd) against an fp64 torch restatement of
y_l = act(W_l y_{l-1} + b_l) and its input-gradient chain (reference models/pivotcvae.py:159-174, 205-227, 229-240), against the
layer-by-layer GEMM route of the product, and the property the data-parallel path leans on: a row's results do not depend on the batch
it sits in."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LEAKY, NONE = 1, 0


def make_stack(M, K0, widths, seed, bias=True, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(M, K0, generator=g) * scale).to(DEV)
    layers, K = [], K0
    for i, n in enumerate(widths):
        W = (torch.randn(n, K, generator=g) / np.sqrt(K)).to(DEV)
        b = (torch.randn(n, generator=g) * 0.1).to(DEV) if bias else None
        layers.append((W, b, LEAKY if i < len(widths) - 1 else NONE))
        K = n
    return x, layers


def ref_fwd(x, layers):
    ys, h = [], x.double()
    for W, b, act in layers:
        h = h @ W.double().t() + (b.double() if b is not None else 0.0)
        if act == LEAKY:
            h = torch.where(h > 0, h, 0.01 * h)
        ys.append(h)
    return ys


def ref_bwd(layers, ys_dev, g, dx_cols):
    """the chain on the DEVICE's activations (so a LeakyReLU kink lands on the same side in both): gout[l] for l = n-1 .. 0"""
    n = len(layers)
    gout, cur = [None] * n, g.double()
    for l in range(n - 1, 0, -1):
        cur = cur @ layers[l][0].double()
        if layers[l - 1][2] == LEAKY:
            cur = cur * torch.where(ys_dev[l - 1] > 0, 1.0, 0.01).double()
        gout[l] = cur
    if dx_cols:
        gout[0] = (cur @ layers[0][0].double())[:, :dx_cols]
    return gout


SHAPES = [
    # M, K0, widths, dx_cols
    (1024, 198, [256, 256, 32], 0),        # config 2 encoder (+ packed heads)
    (1024, 38, [128, 128, 32], 0),         # config 2 prior
    (1024, 54, [256, 256, 128], 16),       # config 2 slate completion, z-only input gradient
    (1000, 1419, [256, 256, 32], 0),       # config 4 widths, ragged M
    (64, 715, [256, 256, 576], 16),        # config 3 slate completion: a 576-wide last layer
    (17, 23, [50, 24, 80], 23),            # nothing a multiple of anything
    (5, 7, [8], 7),                        # one narrow layer
    (33, 100, [16, 48, 64, 130], 100),     # four layers, widths on both sides of the column / reduction split
    (256, 64, [64], 64),
]


@pytest.mark.parametrize("M,K0,widths,dx_cols", SHAPES)
def test_stack_against_fp64(M, K0, widths, dx_cols):
    from pivotcvae_amd import ops
    x, layers = make_stack(M, K0, widths, seed=M + K0)
    ys = ops.stack_fwd_raw([(x, layers)])[0]
    want = ref_fwd(x, layers)
    for l, (y, w) in enumerate(zip(ys, want)):
        scale = float(w.abs().max())
        assert float((y.double() - w).abs().max()) <= 2e-6 * scale + 1e-7, f"layer {l}"
    g = torch.randn(M, widths[-1], generator=torch.Generator().manual_seed(3)).to(DEV)
    gout = ops.stack_bwd_raw([(x, layers)], [ys], [g], [dx_cols])[0]
    wantg = ref_bwd(layers, ys, g, dx_cols)
    for l in range(len(layers)):
        if wantg[l] is None:
            assert gout[l] is None
            continue
        scale = float(wantg[l].abs().max())
        assert float((gout[l].double() - wantg[l]).abs().max()) <= 2e-6 * scale + 1e-7, f"gout {l}"


def test_two_stacks_one_launch_and_no_bias():
    from pivotcvae_amd import ops
    a = make_stack(300, 198, [256, 256, 32], seed=1)
    b = make_stack(300, 38, [128, 32], seed=2, bias=False)
    ys = ops.stack_fwd_raw([a, b])
    solo = [ops.stack_fwd_raw([a])[0], ops.stack_fwd_raw([b])[0]]
    for s in range(2):
        for y, z in zip(ys[s], solo[s]):
            assert torch.equal(y, z)   # a stack's results do not depend on its launch companions
    ga = torch.randn(300, 32, device=DEV)
    gb = torch.randn(300, 32, device=DEV)
    both = ops.stack_bwd_raw([a, b], ys, [ga, gb], [0, 38])
    one_a = ops.stack_bwd_raw([a], [ys[0]], [ga], [0])[0]
    one_b = ops.stack_bwd_raw([b], [ys[1]], [gb], [38])[0]
    for t, u in zip(both[0] + both[1], one_a + one_b):
        assert (t is None and u is None) or torch.equal(t, u)
    want = ref_fwd(*b)
    assert float((ys[1][-1].double() - want[-1]).abs().max()) <= 2e-6 * float(want[-1].abs().max())


def test_rows_do_not_depend_on_the_batch():
    """shard a batch: every row's activations and gradients are BITWISE those of the whole batch (the data-parallel contract)"""
    from pivotcvae_amd import ops
    x, layers = make_stack(192, 198, [256, 256, 32], seed=7)
    ys = ops.stack_fwd_raw([(x, layers)])[0]
    g = torch.randn(192, 32, device=DEV)
    gout = ops.stack_bwd_raw([(x, layers)], [ys], [g], [198])[0]
    for lo, hi in ((0, 64), (64, 101), (101, 192)):
        xs = x[lo:hi].contiguous()
        ys_s = ops.stack_fwd_raw([(xs, layers)])[0]
        for y, z in zip(ys, ys_s):
            assert torch.equal(y[lo:hi], z)
        gs = ops.stack_bwd_raw([(xs, layers)], [ys_s], [g[lo:hi].contiguous()], [198])[0]
        for t, u in zip(gout, gs):
            assert torch.equal(t[lo:hi], u)


def test_output_window_and_strided_gradient():
    """the last layer writes a column window of a wider buffer (rx next to the pivot row), the upstream gradient is such a window"""
    from pivotcvae_amd import ops
    M, D = 100, 32
    x, layers = make_stack(M, 54, [256, 128], seed=11)
    buf = torch.full((M, D + 128), 7.0, device=DEV)
    mids = [torch.empty(M, 256, device=DEV)]
    ops.stack_fwd_raw([(x, layers)], outs=[mids + [buf[:, D:]]])
    plain = ops.stack_fwd_raw([(x, layers)])[0]
    assert torch.equal(buf[:, D:], plain[1]) and bool((buf[:, :D] == 7.0).all())
    gbuf = torch.randn(M, D + 128, device=DEV)
    a = ops.stack_bwd_raw([(x, layers)], [plain], [gbuf[:, D:]], [16])[0]
    b = ops.stack_bwd_raw([(x, layers)], [plain], [gbuf[:, D:].contiguous()], [16])[0]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("B", [64, 1024])
def test_train_step_fused_stacks_against_layer_gemms(B, monkeypatch):
    """one optimisation step with the stack kernels against the same step through the layer-by-layer GEMMs (PCVAE_STACK_FUSED=0)"""
    import bench
    from pivotcvae_amd.train_generative import Trainer
    cfg = dict(bench.CONFIGS["2"], B=B)
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("PCVAE_STACK_FUSED", fused)
        model, _ = bench.build_model(cfg, torch.device(DEV), "f32")
        s, r, u = bench.synthetic_batch(cfg, B, torch.device(DEV))
        eps = torch.randn(B, bench.Z, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
        tr = Trainer(model, lr=bench.LR, beta=bench.BETA)
        terms = [t.item() for t in tr.step(s, r, u, eps=eps)]
        res[fused] = (terms, tr.opt.grad.clone(), tr.opt.flat.clone())
    np.testing.assert_allclose(res["1"][0], res["0"][0], rtol=2e-6)
    ga, gb = res["1"][1], res["0"][1]
    scale = float(gb.abs().max())
    # a LeakyReLU kink that lands on the other side in the other k order moves a few entries by more than rounding: bound their share
    off = (ga - gb).abs() > 2e-5 * scale
    assert float(off.float().mean()) <= 2e-3 and float((ga - gb).abs().max()) <= 2e-2 * scale
    assert float((res["1"][2] - res["0"][2]).abs().max()) <= 2.001 * bench.LR
