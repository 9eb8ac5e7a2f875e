"""-m gpu: every C-ABI kernel against the oracle on the same seeded inputs (through ctypes -> libpcvae_hip.so).

Tolerances (fp32 path): GEMM-class results rtol 1e-5 (different summation order from the CPU BLAS),
catalog CE nll/lse rtol 2e-6 and gradient 2e-5 against the double-precision C oracle, greedy ids BIT-EXACT.
"""
import os

import numpy as np
import pytest
import torch

from oracle import catalog_oracle as co
from oracle import pivotcvae_oracle as orc
from tests import philox_ref

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from pivotcvae_amd import ops as _ops
    return _ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def unit_rows(N, D, seed=0):
    return orc.normalize_rows(rnd(N, D, seed=seed))


# ------------------------------------------------------------------------------------------ K1/K2
@pytest.mark.parametrize("D,group,pad", [(16, 1, 0), (128, 10, 0), (32, 5, 7), (20, 3, 1), (256, 1, 0)])
def test_gather_rows(ops, D, group, pad):
    N, B = 1000, 37
    table = rnd(N, D, seed=1)
    idx = torch.randint(0, N, (B * group,), generator=torch.Generator().manual_seed(2))
    idx[0], idx[-1] = 0, N - 1
    buf = torch.full((B, group * D + pad), -7.0, device=DEV)
    out = ops.gather_rows(table.to(DEV), idx.to(DEV), out=buf[:, : group * D] if pad else None, group=group)
    want = table[idx].reshape(B, group * D)
    assert torch.equal(out.cpu(), want)
    if pad:
        assert torch.all(buf[:, group * D:] == -7.0)  # neighbours of the window untouched


@pytest.mark.parametrize("D", [64, 128, 256])
@pytest.mark.parametrize("n_idx,group,pad", [(1, 1, 0), (15, 1, 0), (17, 1, 3), (33, 3, 0), (64, 1, 0), (65, 5, 5), (1000, 10, 0), (4099, 1, 0),
                                             (300_007, 1, 0)])
def test_gather_rows_coalesced_index_kernel(ops, D, n_idx, group, pad):
    """round 4: gather_rows_coal_kernel (D = 64 / 128 / 256: one coalesced index load per wave and batch of 64 / 32 / 16 rows, indices
    handed between lanes): batches that end inside a wave, one row, strided output windows with untouched neighbours, more rows
    than one sweep of the capped grid covers (2048 blocks x 4 waves x 16 rows at D = 256), first and last table row"""
    n_idx = n_idx // group * group or group
    N = 5000
    table = rnd(N, D, seed=1)
    idx = torch.randint(0, N, (n_idx,), generator=torch.Generator().manual_seed(n_idx))
    idx[0], idx[-1] = N - 1, 0
    B = n_idx // group
    buf = torch.full((B, group * D + pad), -7.0, device=DEV)
    out = ops.gather_rows(table.to(DEV), idx.to(DEV), out=buf[:, : group * D] if pad else None, group=group)
    assert torch.equal(out.cpu(), table[idx].reshape(B, group * D))
    if pad:
        assert torch.all(buf[:, group * D:] == -7.0)


def test_gather_empty_and_errors(ops):
    table = rnd(10, 16).to(DEV)
    out = ops.gather_rows(table, torch.zeros(0, dtype=torch.long, device=DEV))
    assert out.shape == (0, 16)
    with pytest.raises(RuntimeError):
        ops.gather_rows(rnd(10, 16), torch.zeros(1, dtype=torch.long))  # CPU tensors: loud failure, no fallback


def test_condition(ops):
    r = (rnd(64, 5, seed=3) > 0).float()
    r[0], r[1] = 0, 1
    assert torch.equal(ops.condition(r.to(DEV), 5).cpu(), orc.condition(r, 5))


def test_condition_context_narrower_than_the_slate(ops):
    """train_generative.py:179 builds a 5-column context whatever the slate size is; only its row sum is used.  The
    kernel once read S columns from it (out of bounds for S > 5: a memory fault at config 5 when the context happened to
    sit at the end of a mapped region).  Each context gets its own allocation-sized tensor here, S = 20."""
    S = 20
    for B in (1, 3, 1024):
        ctx = torch.zeros(B, 5)
        for i in range(5):
            ctx[:, i] = 1
            got = ops.condition(ctx.clone().to(DEV), S).cpu()
            assert torch.equal(got, orc.condition(ctx, S))
            assert got[:, i + 1].sum() == B


def test_concat_and_backward(ops):
    a, b, c = rnd(9, 4, seed=1), rnd(9, 6, seed=2), rnd(9, 16, seed=3)
    ad = a.to(DEV).requires_grad_(True)
    out = ops.concat([ad, b.to(DEV), c.to(DEV)])
    assert torch.equal(out.cpu(), torch.cat([a, b, c], 1))
    w = rnd(9, 26, seed=4)
    (out * w.to(DEV)).sum().backward()
    assert torch.equal(ad.grad.cpu(), w[:, :4])
    # five parts (two launches of the four-part kernel), one of them a column window of a wider buffer
    parts = [rnd(33, wd, seed=10 + i) for i, wd in enumerate((3, 1, 17, 8, 5))]
    wide = rnd(33, 40, seed=20)
    dev_parts = [q.to(DEV) for q in parts]
    dev_parts[2] = wide.to(DEV)[:, 7:24]
    parts[2] = wide[:, 7:24]
    assert torch.equal(ops.concat(dev_parts).cpu(), torch.cat(parts, 1))


# --------------------------------------------------------------------------------------------- K3
@pytest.mark.parametrize("M,K,N", [(7, 102, 24), (64, 16, 64), (300, 1419, 256), (129, 283, 1152), (5, 24, 4)])
@pytest.mark.parametrize("last_linear", [True, False])
def test_mlp_forward_backward(ops, M, K, N, last_linear):
    H = 48
    x = rnd(M, K, seed=1)
    Ws = [rnd(H, K, seed=2, scale=0.3), rnd(H, H, seed=3, scale=0.3), rnd(N, H, seed=4, scale=0.3)]
    bs = [rnd(H, seed=5), rnd(H, seed=6), rnd(N, seed=7)]
    gout = rnd(M, N, seed=8)

    def ref():
        xs = x.clone().requires_grad_(True)
        ps = [(w.clone().requires_grad_(True), b.clone().requires_grad_(True)) for w, b in zip(Ws, bs)]
        h = xs
        for i, (w, b) in enumerate(ps):
            h = h @ w.t() + b
            if not (last_linear and i == 2):
                h = torch.nn.functional.leaky_relu(h, 0.01)
        (h * gout).sum().backward()
        return h, xs.grad, ps

    want, wgx, wps = ref()
    xd = x.to(DEV).requires_grad_(True)
    pd = [(w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)) for w, b in zip(Ws, bs)]
    got = ops.mlp(xd, pd, last_linear)
    (got * gout.to(DEV)).sum().backward()
    # the MFMA accumulates K sequentially in fp32 (max abs error ~1e-4 at K=1419 against fp64, measured with
    # tools/diag_gemm.py; the CPU BLAS blocks its sums), so absolute tolerances scale with the reduction length
    tol = 1e-5 * max(1.0, K / 64.0)
    torch.testing.assert_close(got.cpu(), want, rtol=1e-4, atol=tol)
    torch.testing.assert_close(xd.grad.cpu(), wgx, rtol=1e-3, atol=tol)
    for (w, b), (rw, rb) in zip(pd, wps):
        torch.testing.assert_close(w.grad.cpu(), rw.grad, rtol=1e-3, atol=10 * tol)
        torch.testing.assert_close(b.grad.cpu(), rb.grad, rtol=1e-3, atol=10 * tol)


@pytest.mark.parametrize("M,K,H,Z,n_trunk,x_grad", [(300, 139, 48, 16, 2, False), (1100, 70, 256, 16, 1, True), (37, 20, 8, 3, 0, True)])
def test_trunk_with_two_heads_is_one_node(ops, M, K, H, Z, n_trunk, x_grad):
    """ops.mlp_heads (trunk + two linear heads, the heads' input gradients summed and masked in a GEMM epilogue) against
    torch autograd on the CPU, values and every gradient."""
    x = rnd(M, K, seed=1)
    dims = [K] + [H] * n_trunk
    trunk = [(rnd(dims[i + 1], dims[i], seed=10 + i, scale=0.3), rnd(dims[i + 1], seed=20 + i)) for i in range(n_trunk)]
    heads = [(rnd(Z, dims[-1], seed=30 + j, scale=0.3), rnd(Z, seed=40 + j)) for j in range(2)]
    ga, gb = rnd(M, Z, seed=50), rnd(M, Z, seed=51)

    xr = x.clone().requires_grad_(x_grad)
    pr = [(w.clone().requires_grad_(True), b.clone().requires_grad_(True)) for w, b in trunk + heads]
    h = xr
    for w, b in pr[:n_trunk]:
        h = torch.nn.functional.leaky_relu(h @ w.t() + b, 0.01)
    ya, yb = h @ pr[n_trunk][0].t() + pr[n_trunk][1], h @ pr[n_trunk + 1][0].t() + pr[n_trunk + 1][1]
    ((ya * ga).sum() + (yb * gb).sum()).backward()

    xd = x.to(DEV).requires_grad_(x_grad)
    pd = [(w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)) for w, b in trunk + heads]
    da, db = ops.mlp_heads(xd, pd[:n_trunk], pd[n_trunk], pd[n_trunk + 1])
    ((da * ga.to(DEV)).sum() + (db * gb.to(DEV)).sum()).backward()
    torch.testing.assert_close(da.cpu(), ya.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.cpu(), yb.detach(), rtol=1e-4, atol=1e-4)
    for (wd, bd), (wr, br) in zip(pd, pr):
        torch.testing.assert_close(wd.grad.cpu(), wr.grad, rtol=1e-4, atol=2e-4)
        torch.testing.assert_close(bd.grad.cpu(), br.grad, rtol=1e-4, atol=2e-4)
    if x_grad:
        torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-4, atol=2e-4)


def test_linear_large_shapes_take_the_64x64_lds_dma_tiles(ops):
    """>= 256 output tiles of 64 x 64: the LDS-DMA body of gemm_group_kernel; K = 283 makes every row start 4-byte but not
    16-byte aligned and leaves a ragged last k chunk (zero-page lanes); M and N leave ragged edge tiles (clamped rows)."""
    M, K, N = 2100, 283, 650
    x, W, b, g = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2), rnd(N, seed=3), rnd(M, N, seed=4)
    xd, Wd, bd, gd = x.to(DEV), W.to(DEV), b.to(DEV), g.to(DEV)
    y = ops.linear_fwd_raw(xd, Wd, bd, 0)
    torch.testing.assert_close(y.cpu(), (x.double() @ W.double().t() + b.double()).float(), rtol=1e-4, atol=1e-4)
    dx = ops.linear_bwd_input_raw(gd, Wd)
    torch.testing.assert_close(dx.cpu(), (g.double() @ W.double()).float(), rtol=1e-4, atol=2e-4)
    dW, db = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
    ops.linear_bwd_weight_raw(gd, xd, dW, db)
    torch.testing.assert_close(dW.cpu(), (g.double().t() @ x.double()).float(), rtol=1e-4, atol=5e-4)
    torch.testing.assert_close(db.cpu(), g.double().sum(0).float(), rtol=1e-4, atol=5e-4)


def test_grouped_launch_equals_single_launches(ops):
    """pcvae_linear_group: forward, input gradient (all columns, a column window, accumulating + masked) and weight gradient of
    unrelated ragged shapes in ONE launch against the single-layer entry points."""
    for M in (1500, 70):
        x, W, b, g = rnd(M, 283, seed=1).to(DEV), rnd(200, 283, seed=2, scale=0.2).to(DEV), rnd(200, seed=3).to(DEV), rnd(M, 200, seed=4).to(DEV)
        x2, W2 = rnd(M, 40, seed=5).to(DEV), rnd(33, 40, seed=6).to(DEV)
        act = torch.nn.functional.leaky_relu(rnd(M, 283, seed=7), 0.01).to(DEV)
        want_y = ops.linear_fwd_raw(x, W, b, 1)
        want_y2 = ops.linear_fwd_raw(x2, W2, None, 0)
        want_dx = ops.linear_bwd_input_raw(g, W, xact=act)
        want_dz = ops.linear_bwd_input_raw(g, W[:, :16])
        want_acc = ops.linear_bwd_input_acc_raw(g, W, act, want_dx.clone())
        want_dW, want_db = torch.zeros(200, 283, device=DEV), torch.zeros(200, device=DEV)
        ops.linear_bwd_weight_raw(g, x, want_dW, want_db)

        grp = ops.GemmGroup()
        y = grp.fwd(x, W, b, 1)
        y2 = grp.fwd(x2, W2, None, 0)
        dx = grp.dx(g, W, xact=act)
        full = torch.full((M, 283), 7.0, device=DEV)
        grp.dx(g, W, out=full[:, :16], cols=16)
        acc = grp.dx(g, W, xact=act, out=want_dx.clone(), accumulate=True)
        dW, db = torch.zeros(200, 283, device=DEV), torch.zeros(200, device=DEV)
        grp.dw(g, x, dW, db)
        grp.launch()
        # (not bitwise: alone, these shapes take the 32 x 32 tiles, whose four waves split K; in the group, 64 x 64 tiles)
        for got, want in ((y, want_y), (y2, want_y2), (dx, want_dx), (acc, want_acc), (full[:, :16], want_dz)):
            torch.testing.assert_close(got, want, rtol=1e-5, atol=2e-5)
        assert torch.all(full[:, 16:] == 7.0)
        small = ops.GemmGroup()   # no weight gradient, few tiles: the 32 x 32 body, bitwise equal to the single launches
        ys, dxs = small.fwd(x2, W2, None, 0), small.dx(g, W, xact=act)
        small.launch()
        if M == 70:
            assert torch.equal(ys, want_y2) and torch.equal(dxs, want_dx)
        torch.testing.assert_close(dW, want_dW, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(db, want_db, rtol=1e-5, atol=1e-5)
        ref = g.double().cpu().t() @ x.double().cpu()
        torch.testing.assert_close(dW.cpu(), ref.float(), rtol=1e-4, atol=5e-4)
    with pytest.raises(RuntimeError):   # seven problems are two launches through the helper, but the ABI itself takes at most six
        from pivotcvae_amd import _hip
        _hip.check(_hip.lib().pcvae_linear_group((_hip.GemmDesc * 7)(), 7, None, 0, None), "linear_group")


def test_weight_gradient_is_bitwise_reproducible(ops):
    """the batch splits of a weight gradient meet in the group's scratch buffer and are summed in split order by the last workgroup
    to arrive: run-to-run BITWISE equal, equal to fp64 at fp32 accuracy, accumulating into what the buffers held, counters left
    zero (a second, differently shaped launch in between); the single-layer C entry point (one split, no scratch) agrees."""
    from pivotcvae_amd import _hip
    M, N, K = 8192, 256, 1419
    g, x = rnd(M, N, seed=1).to(DEV), rnd(M, K, seed=2).to(DEV)
    ref = (g.double().cpu().t() @ x.double().cpu())
    outs = []
    for rep in range(3):
        dW, db = torch.full((N, K), 0.5, device=DEV), torch.full((N,), -1.0, device=DEV)
        ops.linear_bwd_weight_raw(g, x, dW, db)
        outs.append((dW.clone(), db.clone()))
        d2, b2 = torch.zeros(48, 70, device=DEV), torch.zeros(48, device=DEV)   # another shape through the same scratch buffer
        ops.linear_bwd_weight_raw(g[:300, :48], x[:300, :70], d2, b2)
        torch.testing.assert_close(d2.cpu(), (g[:300, :48].double().cpu().t() @ x[:300, :70].double().cpu()).float(), rtol=1e-4, atol=1e-4)
    for dW, db in outs[1:]:
        assert torch.equal(dW, outs[0][0]) and torch.equal(db, outs[0][1])
    torch.testing.assert_close(outs[0][0].cpu(), (ref + 0.5).float(), rtol=1e-4, atol=5e-4)
    torch.testing.assert_close(outs[0][1].cpu(), (g.double().cpu().sum(0) - 1.0).float(), rtol=1e-4, atol=5e-4)
    # the single-layer C entry point (no scratch buffer: one batch split) agrees to rounding
    dA, bA = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
    _hip.check(_hip.lib().pcvae_linear_bwd_weight(_hip.ptr(g), N, _hip.ptr(x), K, _hip.ptr(dA), K, _hip.ptr(bA), M, N, K, _hip.stream()),
               "linear_bwd_weight")
    torch.testing.assert_close(dA, outs[0][0] - 0.5, rtol=1e-5, atol=5e-4)   # 8192-term fp32 sums in two different orders


@pytest.mark.parametrize("M,N,K", [(256, 64, 64), (256, 256, 1408), (300, 256, 1419), (129, 1152, 283), (1000, 48, 283)])
def test_weight_gradient_many_batch_splits_repeated(ops, M, N, K):
    """Batch splits of one output tile run on different XCDs.  Combined with fp32 atomicAdd they lost updates line by line (a whole
    split's contribution missing in a few percent of the elements, launch after launch, with hipMemset or a fill kernel in front:
    tools/atomic_tile_probe.hip shows the same without any of our kernels); ten launches each, against fp64, zeroed by a fill kernel
    immediately before - the situation that failed."""
    g, x = rnd(M, N, seed=3), rnd(M, K, seed=4)
    gd, xd = g.to(DEV), x.to(DEV)
    want_W, want_b = (g.double().t() @ x.double()).float(), g.double().sum(0).float()
    for _ in range(10):
        dW, db = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
        ops.linear_bwd_weight_raw(gd, xd, dW, db)
        torch.testing.assert_close(dW.cpu(), want_W, rtol=1e-4, atol=2e-4 * max(1.0, (M / 64.0) ** 0.5))
        torch.testing.assert_close(db.cpu(), want_b, rtol=1e-4, atol=2e-4 * max(1.0, (M / 64.0) ** 0.5))


def test_linear_random_ragged_shapes(ops):
    """forty random (M, N, K) with ragged everything - K below one 32-chunk, M / N below one tile, odd leading dimensions, column
    windows - through forward, input gradient and weight gradient, each alone (32 x 32 or 64 x 64 tiles by size) and as one group
    (64 x 64 LDS-DMA tiles), against fp64 on the CPU.  PCVAE_FUZZ_SEEDS="1,2,.." runs other shape sequences as well (one-off
    campaigns; the default is the committed sequence)."""
    for seed in [int(v) for v in os.environ.get("PCVAE_FUZZ_SEEDS", "1234").split(",")]:
        _linear_ragged_sequence(ops, 1234 if seed == 1234 else 1234 + 7919 * seed)


def _linear_ragged_sequence(ops, seed):
    import random
    rng = random.Random(seed)
    for case in range(40):
        M = rng.choice([1, 2, 31, 33, 63, 64, 65, 127, 200, 1000, 1300])
        N = rng.choice([1, 3, 16, 31, 32, 33, 64, 65, 100, 130])
        K = rng.choice([1, 2, 7, 31, 32, 33, 63, 64, 65, 96, 97, 283])
        padx, padw, pady = rng.choice([0, 1, 3]), rng.choice([0, 2]), rng.choice([0, 5])
        xb, Wb = rnd(M, K + padx, seed=100 + case), rnd(N, K + padw, seed=200 + case, scale=0.3)
        b, g = rnd(N, seed=300 + case), rnd(M, N + pady, seed=400 + case)
        xd, Wd, bd, gd = xb.to(DEV)[:, padx:], Wb.to(DEV)[:, :K], b.to(DEV), g.to(DEV)[:, :N]
        x64, W64, g64 = xb[:, padx:].double(), Wb[:, :K].double(), g[:, :N].double()
        tol = dict(rtol=1e-4, atol=2e-5 * max(1.0, (max(K, N, M) / 64.0) ** 0.5))
        want_y = torch.nn.functional.leaky_relu(x64 @ W64.t() + b.double(), 0.01).float()
        want_dx, want_dW, want_db = (g64 @ W64).float(), (g64.t() @ x64).float(), g64.sum(0).float()
        # alone
        torch.testing.assert_close(ops.linear_fwd_raw(xd, Wd, bd, 1).cpu(), want_y, **tol)
        torch.testing.assert_close(ops.linear_bwd_input_raw(gd, Wd).cpu(), want_dx, **tol)
        dW, db = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
        ops.linear_bwd_weight_raw(gd, xd, dW, db)
        torch.testing.assert_close(dW.cpu(), want_dW, **tol)
        torch.testing.assert_close(db.cpu(), want_db, **tol)
        # as one group, outputs into column windows of wider buffers
        grp = ops.GemmGroup()
        ybuf, dxbuf = torch.full((M, N + 3), 9.0, device=DEV), torch.full((M, K + 2), 9.0, device=DEV)
        grp.fwd(xd, Wd, bd, 1, out=ybuf[:, 3:])
        grp.dx(gd, Wd, out=dxbuf[:, :K])
        dW2, db2 = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
        grp.dw(gd, xd, dW2, db2)
        grp.launch()
        torch.testing.assert_close(ybuf[:, 3:].cpu(), want_y, **tol)
        torch.testing.assert_close(dxbuf[:, :K].cpu(), want_dx, **tol)
        torch.testing.assert_close(dW2.cpu(), want_dW, **tol)
        torch.testing.assert_close(db2.cpu(), want_db, **tol)
        assert torch.all(ybuf[:, :3] == 9.0) and torch.all(dxbuf[:, K:] == 9.0), "wrote outside its window"


def test_linear_on_column_windows(ops):
    """inputs / outputs may be column windows of wider buffers (ld > width)."""
    big = rnd(33, 50, seed=1).to(DEV)
    W, b = rnd(20, 30, seed=2).to(DEV), rnd(20, seed=3).to(DEV)
    outbuf = torch.zeros(33, 64, device=DEV)
    ops.linear_fwd_raw(big[:, 5:35], W, b, 0, out=outbuf[:, 8:28])
    want = big[:, 5:35].cpu() @ W.cpu().t() + b.cpu()
    torch.testing.assert_close(outbuf[:, 8:28].cpu(), want, rtol=1e-5, atol=1e-5)
    assert torch.all(outbuf[:, :8] == 0) and torch.all(outbuf[:, 28:] == 0)


def test_dense_scores(ops):
    rx, E = rnd(35, 16, seed=1), unit_rows(203, 16, seed=2)
    rd = rx.to(DEV).requires_grad_(True)
    p = ops.dense_scores(rd, E.to(DEV))
    torch.testing.assert_close(p.cpu(), rx @ E.t(), rtol=1e-5, atol=1e-6)
    g = rnd(35, 203, seed=3)
    (p * g.to(DEV)).sum().backward()
    torch.testing.assert_close(rd.grad.cpu(), g @ E, rtol=1e-5, atol=1e-5)


# --------------------------------------------------------------------------------------------- K4
def test_reparam_with_given_eps(ops):
    mu, lv, eps = rnd(50, 16, seed=1), rnd(50, 16, seed=2), torch.randn(50, 16, generator=torch.Generator().manual_seed(3))
    md, ld = mu.to(DEV).requires_grad_(True), lv.to(DEV).requires_grad_(True)
    z, used = ops.reparam(md, ld, eps.to(DEV))
    torch.testing.assert_close(z.cpu(), orc.reparametrize(mu, lv, eps), rtol=1e-6, atol=1e-6)
    assert torch.equal(used.cpu(), eps)
    g = rnd(50, 16, seed=4)
    (z * g.to(DEV)).sum().backward()
    torch.testing.assert_close(md.grad.cpu(), g, rtol=0, atol=0)
    torch.testing.assert_close(ld.grad.cpu(), g * eps * 0.5 * torch.exp(0.5 * lv), rtol=1e-5, atol=1e-6)


def test_reparam_philox_stream(ops):
    B, Z = 8192, 16
    mu, lv = torch.zeros(B, Z, device=DEV), torch.zeros(B, Z, device=DEV)
    z1, e1 = ops.reparam(mu, lv, None, seed=7, offset=0)
    z2, e2 = ops.reparam(mu, lv, None, seed=7, offset=0)
    assert torch.equal(e1, e2) and torch.equal(z1, e1)  # deterministic; z = eps when mu=0, logvar=0
    e = e1.cpu().double()
    assert abs(e.mean().item()) < 0.01 and abs(e.std().item() - 1) < 0.01
    assert abs((e ** 3).mean().item()) < 0.03 and abs((e ** 4).mean().item() - 3) < 0.1
    # a shard that starts at slate 100 sees exactly the stream the full batch saw there (DP independence)
    _, es = ops.reparam(mu[100:300], lv[100:300], None, seed=7, offset=100 * Z)
    assert torch.equal(es, e1[100:300])
    _, e3 = ops.reparam(mu, lv, None, seed=8, offset=0)
    assert not torch.equal(e3, e1)


# --------------------------------------------------------------------------------------------- K7
def test_kld_forward_backward(ops):
    ts = [rnd(300, 16, seed=i) for i in range(4)]
    ref = [t.clone().requires_grad_(True) for t in ts]
    k = orc.kld(*ref)
    (k * 0.37).backward()
    dv = [t.to(DEV).requires_grad_(True) for t in ts]
    kd = ops.kld(*dv)
    (kd * 0.37).backward()
    np.testing.assert_allclose(kd.item(), k.item(), rtol=2e-6)
    for a, b in zip(dv, ref):
        torch.testing.assert_close(a.grad.cpu(), b.grad, rtol=1e-5, atol=1e-6)


def test_latent_is_reparam_plus_kld_with_one_backward(ops):
    """ops.latent (one autograd node, fused backward kernel) against the oracle's reparametrize + KLD under autograd, with both
    outputs used the way the loss uses them (z through a linear map, beta * KLD added)."""
    B, Z = 300, 16
    ts = [rnd(B, Z, seed=i) for i in range(4)]
    eps = torch.randn(B, Z, generator=torch.Generator().manual_seed(9))
    w = rnd(B, Z, seed=7)
    ref = [t.clone().requires_grad_(True) for t in ts]
    zr = orc.reparametrize(ref[0], ref[1], eps)
    kr = orc.kld(*ref)
    ((zr * w).sum() + 0.37 * kr).backward()
    dv = [t.to(DEV).requires_grad_(True) for t in ts]
    z, used, k = ops.latent(*dv, eps.to(DEV))
    ((z * w.to(DEV)).sum() + 0.37 * k).backward()
    torch.testing.assert_close(z.cpu(), zr.detach(), rtol=1e-6, atol=1e-6)
    assert torch.equal(used.cpu(), eps)
    np.testing.assert_allclose(k.item(), kr.item(), rtol=2e-6)
    for a, b in zip(dv, ref):
        torch.testing.assert_close(a.grad.cpu(), b.grad, rtol=1e-5, atol=1e-6)
    # only the KL term used (z dropped): the reparameterisation contributes nothing
    dv2 = [t.to(DEV).requires_grad_(True) for t in ts]
    _, _, k2 = ops.latent(*dv2, eps.to(DEV))
    (0.37 * k2).backward()
    ref2 = [t.clone().requires_grad_(True) for t in ts]
    (0.37 * orc.kld(*ref2)).backward()
    for a, b in zip(dv2, ref2):
        torch.testing.assert_close(a.grad.cpu(), b.grad, rtol=1e-5, atol=1e-6)
    # Philox path: the same stream as ops.reparam
    z3, e3, _ = ops.latent(*[t.to(DEV) for t in ts], None, seed=5, offset=32)
    z4, e4 = ops.reparam(ts[0].to(DEV), ts[1].to(DEV), None, seed=5, offset=32)
    assert torch.equal(e3, e4) and torch.equal(z3, z4)


# --------------------------------------------------------------------------------------------- K8
def test_adam_matches_oracle(ops):
    n = 5000
    p0, sd = rnd(n, seed=1), None
    state = {}
    pd = p0.to(DEV).clone()
    md, vd = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    sd = {"w": p0.clone()}
    for t in range(1, 6):
        g = rnd(n, seed=10 + t, scale=0.1)
        sd = orc.adam_step(sd, {"w": g}, state, 3e-4)
        ops.adam_step_(pd, g.to(DEV), md, vd, 3e-4, t)
        torch.testing.assert_close(pd.cpu(), sd["w"], rtol=1e-6, atol=1e-7)
    # zero gradient leaves parameters bit-identical (what "skip grad None" means for the PSM)
    before = pd.clone()
    m0, v0 = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    ops.adam_step_(pd, torch.zeros(n, device=DEV), m0, v0, 3e-4, 1)
    assert torch.equal(pd, before)


# ---------------------------------------------------------------------------------------- K5 / K6
CAT_SHAPES = [(35, 203, 16), (50, 321, 32), (130, 1000, 64), (257, 4099, 128), (40, 2500, 256), (128, 64, 32)]


@pytest.mark.parametrize("R,N,D", CAT_SHAPES)
def test_catalog_ce_full_softmax(ops, R, N, D):
    rx, E = rnd(R, D, seed=1, scale=2.0), unit_rows(N, D, seed=2)
    tgt = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(3))
    tgt[0], tgt[-1] = 0, N - 1
    nll, lse, dx = ops.catalog_ce_raw(rx.to(DEV), E.to(DEV), tgt.to(DEV))
    wn, wl, wd = co.ce(rx.numpy(), E.numpy(), tgt.numpy())
    np.testing.assert_allclose(lse.cpu().numpy(), wl, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(nll.cpu().numpy(), wn, rtol=2e-6, atol=3e-6)
    np.testing.assert_allclose(dx.cpu().numpy(), wd, rtol=2e-5, atol=2e-6)
    nll2, _, none = ops.catalog_ce_raw(rx.to(DEV), E.to(DEV), tgt.to(DEV), want_dx=False)
    assert none is None and torch.equal(nll2, nll)  # loss-only variant is the same arithmetic


@pytest.mark.parametrize("R,N,D", [(35, 203, 16), (130, 1000, 64), (64, 4099, 128)])
def test_catalog_ce_explicit_mask(ops, R, N, D):
    rx, E = rnd(R, D, seed=4, scale=2.0), unit_rows(N, D, seed=5)
    tgt = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(6))
    keep = (torch.rand(R, N, generator=torch.Generator().manual_seed(7)) < 0.25).to(torch.uint8)
    nll, lse, dx = ops.catalog_ce_raw(rx.to(DEV), E.to(DEV), tgt.to(DEV), keep_mask=keep.to(DEV))
    wn, wl, wd = co.ce(rx.numpy(), E.numpy(), tgt.numpy(), keep.numpy())
    np.testing.assert_allclose(nll.cpu().numpy(), wn, rtol=2e-6, atol=3e-6)
    np.testing.assert_allclose(dx.cpu().numpy(), wd, rtol=2e-5, atol=2e-6)


SPARSE_CASES = [(35, 5000, 16, 0.01), (70, 50000, 128, 0.002), (33, 20000, 64, 0.02), (20, 3000, 256, 0.3), (40, 100, 32, 0.05),
                (9, 40, 128, 0.5), (10, 50000, 64, 0.1), (12, 3000, 8, 0.05), (5, 1, 16, 0.5), (70, 63, 32, 0.2)]


@pytest.mark.parametrize("R,N,D,p", SPARSE_CASES)
def test_catalog_ce_sparse_is_the_documented_stream(ops, R, N, D, p):
    """The sparse path (n_neg << N: only the kept rows are read) against the oracle's dense masked CE with the SAME kept set,
    rebuilt on the host from the documented Philox stream (tests/philox_ref.sparse_keep_mask): exact-parity check, at the f32
    kernel's tolerances.  Cases: every width incl. a padded one (D = 8), more kept items than one LDS batch holds
    (50000 x 0.1 > 2048), segments shorter than a lane (N < 64), N = 1, targets first / last."""
    rx, E = rnd(R, D, seed=8, scale=2.0), unit_rows(N, D, seed=9)
    tgt = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(10))
    tgt[0], tgt[-1] = 0, N - 1
    seed, off = 1234567, 1000
    nll, lse, dx = ops.catalog_ce_sparse_raw(rx.to(DEV), E.to(DEV), tgt.to(DEV), p, seed=seed, row_offset=off)
    keep = philox_ref.sparse_keep_mask(R, N, p, seed, off)
    wn, wl, wd = co.ce(rx.numpy(), E.numpy(), tgt.numpy(), keep)
    np.testing.assert_allclose(lse.cpu().numpy(), wl, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(nll.cpu().numpy(), wn, rtol=2e-6, atol=3e-6)
    np.testing.assert_allclose(dx.cpu().numpy(), wd, rtol=2e-5, atol=2e-6)
    nll2, lse2, none = ops.catalog_ce_sparse_raw(rx.to(DEV), E.to(DEV), tgt.to(DEV), p, seed=seed, row_offset=off, want_dx=False)
    assert none is None and torch.equal(nll2, nll) and torch.equal(lse2, lse)
    # a shard of the rows draws what the whole batch drew for those rows (row_offset = global row index)
    h = R // 2
    if h:
        nll3, _, dx3 = ops.catalog_ce_sparse_raw(rx[h:].to(DEV), E.to(DEV), tgt[h:].to(DEV), p, seed=seed, row_offset=off + h)
        assert torch.equal(nll3, nll[h:]) and torch.equal(dx3, dx[h:])


def test_catalog_ce_sparse_kept_counts(ops):
    """Reads the kept COUNT of every row back out of the kernel: all table rows equal e0 and x = ln(3) e0, so every kept logit is
    ln 3 and L = N + 2 n_kept exactly.  Counts equal the host restatement's row by row, and are Binomial(N, p) in mean and variance."""
    R, N, D, p = 600, 100_000, 32, 0.01
    e0 = torch.zeros(D)
    e0[3] = 1.0
    E = e0.repeat(N, 1)
    rx = (np.log(3.0) * e0).repeat(R, 1).float()
    tgt = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(1))
    _, lse, _ = ops.catalog_ce_sparse_raw(rx.to(DEV), E.to(DEV), tgt.to(DEV), p, seed=99, row_offset=7, want_dx=False)
    L = torch.exp(lse.double().cpu())
    kept = ((L - N) / 2.0).round().long().numpy()
    assert np.abs((L - N) / 2.0 - kept).max() < 0.2
    keep = philox_ref.sparse_keep_mask(R, N, p, 99, 7)
    keep[np.arange(R), tgt.numpy()] = 1
    assert np.array_equal(kept, keep.sum(1))
    mean, var = kept.mean(), kept.var()
    assert abs(mean - (N * p + 1)) < 4 * (N * p / R) ** 0.5 + 1 and 0.8 * N * p < var < 1.25 * N * p


def test_catalog_ce_routes_small_keep_prob_to_the_sparse_path(ops):
    R, N, D = 40, 30000, 64
    rx, E = rnd(R, D, seed=8, scale=2.0), unit_rows(N, D, seed=9)
    tgt = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(10))
    p = 1000.0 / N * 0.5
    assert ops.sparse_ce_applies(p, N) and not ops.sparse_ce_applies(0.2, N)
    a = ops.catalog_ce_raw(rx.to(DEV), E.to(DEV), tgt.to(DEV), keep_prob=p, seed=3, row_offset=5)
    b = ops.catalog_ce_sparse_raw(rx.to(DEV), E.to(DEV), tgt.to(DEV), p, seed=3, row_offset=5)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    # autograd wrapper: mean-reduced loss and its gradient
    rd = rx.to(DEV).requires_grad_(True)
    loss = ops.catalog_ce(rd, E.to(DEV), tgt.to(DEV), p, 3, 5)
    loss.backward()
    np.testing.assert_allclose(loss.item(), b[0].mean().item(), rtol=1e-6)
    torch.testing.assert_close(rd.grad, b[2] / R, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("R,N,D,p", [(35, 203, 16, 0.3), (130, 4099, 64, 0.05)])
def test_catalog_ce_philox_mask_is_the_documented_stream(ops, R, N, D, p):
    """In-kernel Bernoulli mask == the host restatement of the same Philox stream -> exact CE check."""
    rx, E = rnd(R, D, seed=8, scale=2.0), unit_rows(N, D, seed=9)
    tgt = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(10))
    seed, off = 1234567, 1000
    nll, _, dx = ops.catalog_ce_raw(rx.to(DEV), E.to(DEV), tgt.to(DEV), keep_prob=p, seed=seed, row_offset=off)
    keep = philox_ref.keep_mask(R, N, p, seed, off)
    assert abs(keep.mean() - p) < 4 * np.sqrt(p * (1 - p) / keep.size) + 1e-3
    wn, _, wd = co.ce(rx.numpy(), E.numpy(), tgt.numpy(), keep)
    np.testing.assert_allclose(nll.cpu().numpy(), wn, rtol=2e-6, atol=3e-6)
    np.testing.assert_allclose(dx.cpu().numpy(), wd, rtol=2e-5, atol=2e-6)
    # sharding independence: rows [10, 20) as their own call with row_offset + 10
    nll_s, _, _ = ops.catalog_ce_raw(rx[10:20].to(DEV), E.to(DEV), tgt[10:20].to(DEV), keep_prob=p, seed=seed,
                                     row_offset=off + 10)
    assert torch.equal(nll_s, nll[10:20])


def test_catalog_ce_autograd_mean(ops):
    R, N, D = 70, 500, 32
    rx, E = rnd(R, D, seed=1, scale=2.0), unit_rows(N, D, seed=2)
    tgt = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(3))
    rr = rx.clone().requires_grad_(True)
    want = torch.nn.functional.cross_entropy(rr @ E.t(), tgt)
    (want * 1.7).backward()
    rd = rx.to(DEV).requires_grad_(True)
    got = ops.catalog_ce(rd, E.to(DEV), tgt.to(DEV))
    (got * 1.7).backward()
    np.testing.assert_allclose(got.item(), want.item(), rtol=2e-6)
    torch.testing.assert_close(rd.grad.cpu(), rr.grad, rtol=2e-5, atol=1e-7)


def test_catalog_ce_extreme_logits(ops):
    """online-softmax rescale branch: one row's maximum jumps late in the catalog, logits up to +-60."""
    R, N, D = 64, 2048, 32
    E = unit_rows(N, D, seed=2)
    rx = rnd(R, D, seed=1, scale=0.1)
    rx[3] = E[N - 5] * 60.0   # max found in the very last tile
    rx[4] = E[40] * 60.0      # max found in the second tile
    rx[5] = -E[77] * 50.0     # most logits strongly negative
    tgt = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(3))
    nll, lse, dx = ops.catalog_ce_raw(rx.to(DEV), E.to(DEV), tgt.to(DEV))
    wn, wl, wd = co.ce(rx.numpy(), E.numpy(), tgt.numpy())
    np.testing.assert_allclose(lse.cpu().numpy(), wl, rtol=3e-6, atol=3e-6)
    np.testing.assert_allclose(nll.cpu().numpy(), wn, rtol=3e-6, atol=2e-5)
    np.testing.assert_allclose(dx.cpu().numpy(), wd, rtol=5e-5, atol=5e-6)


@pytest.mark.parametrize("R,N,D", CAT_SHAPES + [(300, 20000, 32)])
def test_catalog_argmax_bit_exact(ops, R, N, D):
    x, E = rnd(R, D, seed=11, scale=2.0), unit_rows(N, D, seed=12)
    idx, best = ops.catalog_argmax(x.to(DEV), E.to(DEV), return_best=True)
    wi, wb = co.argmax(x.numpy(), E.numpy())
    np.testing.assert_array_equal(idx.cpu().numpy(), wi)  # ids bit-exact
    np.testing.assert_array_equal(best.cpu().numpy(), wb)  # and so is the winning score (same fmaf chain)


def test_catalog_argmax_ties_pick_lowest_index(ops):
    N, D = 3000, 32
    E = unit_rows(N, D, seed=1)
    dup = [7, 38, 1500, 2999]      # identical rows in different lane halves / tiles / splits
    E[dup] = E[7].clone()
    x = torch.stack([E[7] * 3.0, E[7] * 0.5, rnd(D, seed=2)])
    idx = ops.catalog_argmax(x.to(DEV), E.to(DEV)).cpu().numpy()
    wi, _ = co.argmax(x.numpy(), E.numpy())
    np.testing.assert_array_equal(idx, wi)
    assert idx[0] == 7 and idx[1] == 7


@pytest.mark.parametrize("R,N,D", [(1, 32, 128), (37, 1000, 128), (300, 20001, 128), (256, 65536, 128), (513, 131075, 128),
                                   (64, 300000, 128), (1, 32, 64), (300, 20001, 64), (513, 131075, 64), (70, 300000, 64),
                                   (1, 32, 256), (300, 20001, 256), (260, 131075, 256), (64, 300000, 256)])
def test_catalog_argmax_screened_bit_exact(ops, R, N, D):
    """bf16 screening + exact fp32 rescoring returns the same ids AND the same winning scores as the fp32 chain."""
    x, E = rnd(R, D, seed=21, scale=2.0), unit_rows(N, D, seed=22)
    idx, best = ops.catalog_argmax(x.to(DEV), E.to(DEV), return_best=True, screened=True)
    wi, wb = co.argmax(x.numpy(), E.numpy())
    np.testing.assert_array_equal(idx.cpu().numpy(), wi)
    np.testing.assert_array_equal(best.cpu().numpy(), wb)
    idx2 = ops.catalog_argmax(x.to(DEV), E.to(DEV), screened=False)
    assert torch.equal(idx2, idx)


@pytest.mark.parametrize("D", [64, 128, 256])
def test_catalog_argmax_screened_near_ties_and_duplicates(ops, D):
    """Rows whose bf16 images collide: duplicates (lowest index wins), items that differ by less than one bf16 ulp
    (the fp32 rescoring must separate them), queries of very different norm, negative-only scores."""
    N = 70000
    E = unit_rows(N, D, seed=5)
    dup = [11, 4097, 33000, 69999]
    E[dup] = E[11].clone()
    E[500] = E[20000] * (1.0 + 3e-4)          # same bf16 image (almost), slightly larger fp32 score
    E[60001] = E[123] * (1.0 - 1e-6)
    x = torch.stack([E[11] * 3.0, E[11] * 1e-3, E[20000] * 2.0, E[123] * 5.0, -E[9] * 4.0, rnd(D, seed=6) * 50.0,
                     torch.zeros(D)])
    idx, best = ops.catalog_argmax(x.to(DEV), E.to(DEV), return_best=True, screened=True)
    wi, wb = co.argmax(x.numpy(), E.numpy())
    np.testing.assert_array_equal(idx.cpu().numpy(), wi)
    np.testing.assert_array_equal(best.cpu().numpy(), wb)
    assert idx[0] == 11 and idx[1] == 11 and idx[2] == 500 and idx[3] == 123 and idx[6] == 0


@pytest.mark.parametrize("D", [64, 128, 256])
def test_catalog_argmax_screened_pipelined_kernels_and_list_overflow(ops, D, monkeypatch):
    """The software-pipelined screening kernels forced onto a small shape (PCVAE_PIPE_MIN_TILES, read per launch): fill slot, steady
    trips, fenced last slots, ragged tail - and a candidate flood: sixteen all-zero queries in ONE wave's rows tie every item of the
    catalog at 0, so that wave's quarter of the workgroup's candidate list (508 entries at D = 64, 1536 beyond; 8-byte entries,
    round 6) overflows and pass B is redone by the exact two-waves kernel.  Ids and winning scores as the fp32 chain's, lowest index on
    the ties."""
    monkeypatch.setenv("PCVAE_PIPE_MIN_TILES", "1")
    N, R = 40003, 600
    E = unit_rows(N, D, seed=31)
    x = rnd(R, D, seed=32, scale=2.0)
    x[:16] = 0.0
    x[300] = E[777] * 3.0
    x[599] = E[N - 1] * 2.0
    idx, best = ops.catalog_argmax(x.to(DEV), E.to(DEV), return_best=True, screened=True)
    wi, wb = co.argmax(x.numpy(), E.numpy())
    np.testing.assert_array_equal(idx.cpu().numpy(), wi)
    np.testing.assert_array_equal(best.cpu().numpy(), wb)
    assert int(idx[:16].abs().sum()) == 0 and idx[300] == 777 and idx[599] == N - 1
    # the same without the flood: nothing overflows, the pipelined pass B alone answers
    x[:16] = rnd(16, D, seed=33, scale=2.0)
    idx, best = ops.catalog_argmax(x.to(DEV), E.to(DEV), return_best=True, screened=True)
    wi, wb = co.argmax(x.numpy(), E.numpy())
    np.testing.assert_array_equal(idx.cpu().numpy(), wi)
    np.testing.assert_array_equal(best.cpu().numpy(), wb)


def test_catalog_argmax_screened_unnormalised_table(ops):
    """Row norms spread over 3 decades: the bound uses the LARGEST row norm, small rows can still win nothing wrongly."""
    N, D, R = 50000, 128, 200
    g = torch.Generator().manual_seed(9)
    E = torch.randn(N, D, generator=g) * torch.logspace(-2, 1, N)[torch.randperm(N, generator=g)].unsqueeze(1)
    x = rnd(R, D, seed=10)
    idx, best = ops.catalog_argmax(x.to(DEV), E.to(DEV), return_best=True, screened=True)
    wi, wb = co.argmax(x.numpy(), E.numpy())
    np.testing.assert_array_equal(idx.cpu().numpy(), wi)
    np.testing.assert_array_equal(best.cpu().numpy(), wb)


def test_catalog_argmax_screened_rejects_other_widths(ops):
    with pytest.raises(ValueError):
        ops.catalog_argmax(rnd(4, 32, seed=1).to(DEV), unit_rows(100, 32, seed=2).to(DEV), screened=True)


@pytest.mark.parametrize("R,N,D,scale,seed,off", [(500, 1000, 128, 3.0, 5, 0), (300, 37, 16, 6.0, 9, 1 << 35), (257, 20011, 64, 1.0, 1, 77),
                                                  (64, 5003, 256, 2.0, 3, 5), (130, 999, 32, 4.0, 2, 0), (90, 500, 20, 3.0, 4, 9)])
def test_catalog_sample_rejection_stream(ops, R, N, D, scale, seed, off):
    """pcvae_catalog_sample draws by rejection: proposals k = 0, 1, .. of a row from Philox (seed, GLOBAL row, k), the sample is the
    first accepted one.  The host restatement (tests/philox_ref.sample_reject) reproduces the ids EXACTLY on every row whose
    accept / reject decisions are beyond fp32 rounding; shards reproduce the whole batch; a padded width (D = 20) is the same draw."""
    x, E = rnd(R, D, seed=seed + 10, scale=scale), unit_rows(N, D, seed=seed + 20)
    table = ops.CatalogTable(E.to(DEV))
    idx = ops.catalog_sample(x.to(DEV), table, seed=seed, row_offset=off).cpu().numpy()
    want, k, safe = philox_ref.sample_reject(x.numpy(), E.numpy(), seed, off)
    assert (want >= 0).all() and safe.mean() > 0.99
    np.testing.assert_array_equal(idx[safe], want[safe])
    assert k.max() >= 2 and np.mean(k == 0) < 0.9            # rejections do happen: later proposals are exercised
    lo = R // 3
    part = ops.catalog_sample(x[lo:].contiguous().to(DEV), table, seed=seed, row_offset=off + lo).cpu().numpy()
    np.testing.assert_array_equal(part, idx[lo:])
    assert not np.array_equal(ops.catalog_sample(x.to(DEV), table, seed=seed + 1, row_offset=off).cpu().numpy(), idx)


def test_catalog_sample_falls_back_to_gumbel_max_when_every_proposal_is_rejected(ops):
    """a catalog whose items ALL score far below zero for some rows (sigmoid ~ 1e-13: no proposal is ever accepted): those rows
    are drawn by the exact Gumbel-max kernel over the whole catalog, the others keep their rejection draw; frequencies of the
    fallback rows follow Categorical(sigmoid(scores)) - the logits differ by up to 4 there, so the draw is far from uniform"""
    N, D, R = 48, 16, 12000
    g = torch.Generator().manual_seed(3)
    base = torch.zeros(D)
    base[0] = 1.0
    E = orc.normalize_rows(base + 0.4 * (torch.rand(N, D, generator=g) - 0.5))     # every item near the first axis
    hard = -30.0 * base + 3.0 * (torch.rand(D, generator=g) - 0.5)
    hard[0] = -30.0
    easy = rnd(1, D, seed=4, scale=2.0).reshape(-1)
    x = torch.stack([hard if i % 2 == 0 else easy for i in range(R)])
    table = ops.CatalogTable(E.to(DEV))
    idx = ops.catalog_sample(x.to(DEV), table, seed=11).cpu().numpy()
    want, _k, safe = philox_ref.sample_reject(x.numpy(), E.numpy(), 11)
    assert (want[0::2] == -1).all() and (want[1::2] >= 0).all()
    np.testing.assert_array_equal(idx[1::2][safe[1::2]], want[1::2][safe[1::2]])
    sc = (hard.double() @ E.double().t())
    probs = torch.softmax(torch.nn.functional.logsigmoid(sc), dim=0).numpy()
    assert probs.max() / probs.min() > 5
    freq = np.bincount(idx[0::2], minlength=N) / (R // 2)
    assert np.abs(freq - probs).max() < 5 * np.sqrt(probs.max() / (R // 2))
    np.testing.assert_array_equal(ops.catalog_sample(x.to(DEV), table, seed=11).cpu().numpy(), idx)


def test_catalog_sample_low_mean_sigmoid_mixes_both_samplers_exactly(ops):
    """the worst case of the rejection sampler - scores mostly far below zero (mean sigmoid ~ 4e-3: an un-normalised
    table, a large-norm PSM output) - is bounded by the proposal cap (512: 32 rounds at D = 128, not round 5's 256): ~10 % of such
    rows reject every proposal and are drawn by the Gumbel-max kernel, the rest by rejection, and the MIXTURE is still
    Categorical(sigmoid(scores)); the host restatement flags exactly the rows that went to the fallback."""
    N, D, R = 48, 16, 30000
    g = torch.Generator().manual_seed(3)
    base = torch.zeros(D)
    base[0] = 1.0
    E = orc.normalize_rows(base + 0.4 * (torch.rand(N, D, generator=g) - 0.5))     # every item near the first axis
    q = -5.6 * base + 1.0 * (torch.rand(D, generator=g) - 0.5)
    q[0] = -5.6
    x = q.expand(R, D).contiguous()
    sc = q.double() @ E.double().t()
    sig = torch.sigmoid(sc)
    assert 1e-3 < float(sig.mean()) < 1e-2
    idx = ops.catalog_sample(x.to(DEV), ops.CatalogTable(E.to(DEV)), seed=21).cpu().numpy()
    want, k, safe = philox_ref.sample_reject(x.numpy(), E.numpy(), 21)
    fell = want < 0
    assert 0.03 < fell.mean() < 0.3 and k[~fell].max() >= 256       # both samplers at work, late proposals exercised
    ok = safe & ~fell
    np.testing.assert_array_equal(idx[ok], want[ok])
    probs = (sig / sig.sum()).numpy()
    assert probs.max() / probs.min() > 1.5
    freq = np.bincount(idx, minlength=N) / R
    assert np.abs(freq - probs).max() < 5 * np.sqrt(probs.max() / R)
    freq_fb = np.bincount(idx[fell], minlength=N) / fell.sum()             # the fallback rows on their own follow the same law
    assert np.abs(freq_fb - probs).max() < 5 * np.sqrt(probs.max() / fell.sum())


def test_catalog_sample_distribution(ops):
    """Categorical(sigmoid(scores)) (rejection sampling): empirical frequencies match the probabilities."""
    N, D, R = 40, 16, 20000
    E = unit_rows(N, D, seed=1)
    q = rnd(1, D, seed=2, scale=3.0)
    probs = torch.sigmoid(q @ E.t()).reshape(-1)
    probs = (probs / probs.sum()).numpy()
    idx = ops.catalog_sample(q.expand(R, D).contiguous().to(DEV), E.to(DEV), seed=5).cpu().numpy()
    freq = np.bincount(idx, minlength=N) / R
    assert np.abs(freq - probs).max() < 5 * np.sqrt(probs.max() / R)
    idx2 = ops.catalog_sample(q.expand(R, D).contiguous().to(DEV), E.to(DEV), seed=5).cpu().numpy()
    np.testing.assert_array_equal(idx, idx2)


# --------------------------------------------------------------------------------------------- K9
def test_candidate_scores(ops):
    R, N, D, Cn = 35, 203, 16, 13
    rx, E = rnd(R, D, seed=1), unit_rows(N, D, seed=2)
    cand = torch.randint(0, N, (R, Cn), generator=torch.Generator().manual_seed(3))
    rd = rx.to(DEV).requires_grad_(True)
    p = ops.candidate_scores(rd, E.to(DEV), cand.to(DEV))
    want = torch.bmm(E[cand], rx.reshape(R, D, 1)).reshape(R, Cn)
    torch.testing.assert_close(p.cpu(), want, rtol=1e-5, atol=1e-6)
    g = rnd(R, Cn, seed=4)
    (p * g.to(DEV)).sum().backward()
    torch.testing.assert_close(rd.grad.cpu(), torch.einsum("rc,rcd->rd", g, E[cand]), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("R,N,D,Cn", [(35, 203, 16, 13), (130, 5000, 128, 1000), (7, 64, 32, 1), (66, 1000, 64, 257),
                                      (9, 300, 256, 70), (33, 97, 20, 50), (5, 50, 128, 4500), (1, 1, 16, 3)])
def test_candidate_ce_fused_given_sets(ops, R, N, D, Cn):
    """pcvae_candidate_ce on GIVEN candidate sets (what a batch of the reference's dataset carries) against the oracle's
    bmm + CrossEntropyLoss + autograd in fp64: duplicates count in the denominator, the target sits at any column, widths without a
    kernel are padded (D = 20), Cn beyond one LDS batch (4500), Cn = 1, N = 1."""
    rx, E = rnd(R, D, seed=1, scale=3.0), unit_rows(N, D, seed=2)
    g = torch.Generator().manual_seed(3)
    cand = torch.randint(0, N, (R, Cn), generator=g)
    tgt = torch.randint(0, Cn, (R,), generator=g)
    cand[0] = cand[0, 0]                    # a row of one repeated id: nll = ln Cn, zero gradient
    want_nll, want_lse, want_dx = orc.candidate_ce(rx, E, cand, tgt)
    table = ops.CatalogTable(E.to(DEV))
    nll, lse, dx, tcol = ops.candidate_ce_raw(rx.to(DEV), table, cand=cand.to(DEV), cand_target=tgt.to(DEV), want_target=True)
    assert torch.equal(tcol.cpu(), tgt)
    np.testing.assert_allclose(nll.cpu().double().numpy(), want_nll.numpy(), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(lse.cpu().double().numpy(), want_lse.numpy(), rtol=2e-6, atol=2e-6)
    tol = max(2e-6, 1e-7 * Cn ** 0.5)       # fp32 accumulation of Cn terms per row (the reference's bmm + softmax are fp32 too)
    assert (dx.cpu().double() - want_dx).abs().max() <= tol * max(1.0, float(want_dx.abs().max()))
    np.testing.assert_allclose(nll[0].item(), np.log(Cn), rtol=1e-6, atol=1e-6)
    # forward only (no dx buffer), and the scaled direction
    nll2, _, none, _ = ops.candidate_ce_raw(rx.to(DEV), table, cand=cand.to(DEV), cand_target=tgt.to(DEV), want_dx=False)
    assert none is None and torch.equal(nll2, nll)
    dx3 = ops.candidate_ce_raw(rx.to(DEV), table, cand=cand.to(DEV), cand_target=tgt.to(DEV), dx_scale=0.25)[2]
    torch.testing.assert_close(dx3, dx * 0.25, rtol=1e-6, atol=1e-9)
    # the autograd op: mean reduction, any upstream gradient; equal to the materialised route (K9 scores + dense CE)
    rd = rx.to(DEV).requires_grad_(True)
    loss = ops.candidate_ce(rd, table, cand=cand.to(DEV), cand_target=tgt.to(DEV))
    (loss * 0.7).backward()
    np.testing.assert_allclose(loss.item(), want_nll.mean().item(), rtol=2e-6)
    assert (rd.grad.cpu().double() - want_dx * 0.7 / R).abs().max() <= tol * max(1.0, float(want_dx.abs().max())) / R
    rm = rx.to(DEV).requires_grad_(True)
    lm = ops.dense_ce(ops.candidate_scores(rm, E.to(DEV), cand.to(DEV)), tgt.to(DEV))
    (lm * 0.7).backward()
    np.testing.assert_allclose(loss.item(), lm.item(), rtol=2e-6)
    torch.testing.assert_close(rd.grad, rm.grad, rtol=1e-4, atol=2 * tol / R)   # (row 0: both routes hold rounding noise around 0)


@pytest.mark.parametrize("R,S,N,D,Cn,seed,off", [(40, 5, 60, 16, 50, 77, 10), (64, 10, 30011, 128, 1000, 13, 0),
                                                 (12, 3, 3000, 64, 5000, 5, 7), (9, 1, 7, 32, 2049, 1, 1 << 33),
                                                 (9, 2, 100003, 32, 2049, 2, 3)])
def test_candidate_ce_fused_draws_the_documented_stream(ops, R, S, N, D, Cn, seed, off):
    """cand == NULL: the kernel draws the sets itself - exactly pcvae_candidate_draw's sets (host restatement: tests/philox_ref.py +
    the oracle's first-hit / overwrite rule), so the results are BITWISE those of the given-sets mode on the drawn ids; both
    branches of the rule occur, hits beyond the first LDS batch included (Cn = 5000 / 2049); shards reproduce the whole batch."""
    gen = torch.Generator().manual_seed(seed)
    sl = torch.randint(0, N, (R, S), generator=gen)
    rx, E = rnd(R * S, D, seed=1, scale=3.0), unit_rows(N, D, seed=2)
    table = ops.CatalogTable(E.to(DEV))
    want_raw = torch.from_numpy(philox_ref.candidate_raw(R * S, Cn, N, seed, off)).view(R, S, Cn)
    wc, wt = orc.candidate_targets(sl, want_raw)
    nll, lse, dx, tcol = ops.candidate_ce_raw(rx.to(DEV), table, Cn, sl.to(DEV).reshape(-1), seed, off, want_target=True)
    assert torch.equal(tcol.cpu(), wt.reshape(-1))
    if N < 40 * Cn:
        assert int((wt > 0).sum()) > 0
    if N > 50:
        assert int((wt == 0).sum()) > 0
    if Cn > 2048 and N == 3000:
        assert int((wt >= 2048).sum()) > 0        # first hits that only the scan of the later batches finds
    n2, l2, d2, _ = ops.candidate_ce_raw(rx.to(DEV), table, cand=wc.to(DEV), cand_target=wt.to(DEV))
    assert torch.equal(nll, n2) and torch.equal(lse, l2) and torch.equal(dx, d2)
    cd, td = ops.candidate_draw(sl.to(DEV), N, Cn, seed=seed, row_offset=off)
    assert torch.equal(cd.cpu(), wc) and torch.equal(td.cpu(), wt)
    want_nll, _, want_dx = orc.candidate_ce(rx, E, wc, wt)
    np.testing.assert_allclose(nll.cpu().double().numpy(), want_nll.numpy(), rtol=2e-6, atol=2e-6)
    assert (dx.cpu().double() - want_dx).abs().max() <= max(2e-6, 1e-7 * Cn ** 0.5) * max(1.0, float(want_dx.abs().max()))
    # a shard of the rows, keyed by its global offset, reproduces its part bit for bit; another seed does not
    lo = (R // 3) * S
    a, b, c, _ = ops.candidate_ce_raw(rx[lo:].contiguous().to(DEV), table, Cn, sl.reshape(-1)[lo:].contiguous().to(DEV), seed, off + lo)
    assert torch.equal(a, nll[lo:]) and torch.equal(b, lse[lo:]) and torch.equal(c, dx[lo:])
    assert not torch.equal(ops.candidate_ce_raw(rx.to(DEV), table, Cn, sl.to(DEV).reshape(-1), seed + 1, off)[1], lse)


@pytest.mark.parametrize("R,S,N,n_items,D,Cn,seed,off", [(40, 5, 1001, 700, 32, 200, 7, 0), (16, 10, 30011, 30000, 128, 1000, 3, 11),
                                                         (9, 2, 100003, 1, 64, 33, 5, 1 << 33), (12, 3, 5000, 4097, 16, 2100, 9, 2)])
def test_candidate_ce_fused_draws_from_the_dataset_id_range(ops, R, S, N, n_items, D, Cn, seed, off):
    """VERDICT r5 missing #3 - data_loader.py:23 (max_iid = np.max(slates)) and :46 (randint(max_iid + 1, ...)): the candidate ids
    come from the DATASET's range [0, max_iid + 1), not from the table's row count (the simulators build n_item + 1 rows,
    env/response_model.py:30).  With n_items < N the in-kernel sets are the documented stream taken mod n_items (host restatement:
    philox_ref.candidate_raw(.., n_items, ..)), no id >= n_items is ever proposed, results are bitwise those of the given-sets
    mode on those ids and meet the oracle; pcvae_candidate_draw(n_items) draws the same sets; n_items > N is refused."""
    gen = torch.Generator().manual_seed(seed)
    sl = torch.randint(0, n_items, (R, S), generator=gen)      # the dataset's slates only use ids below max_iid + 1
    rx, E = rnd(R * S, D, seed=1, scale=3.0), unit_rows(N, D, seed=2)
    table = ops.CatalogTable(E.to(DEV))
    want_raw = torch.from_numpy(philox_ref.candidate_raw(R * S, Cn, n_items, seed, off)).view(R, S, Cn)
    assert int(want_raw.max()) < n_items
    wc, wt = orc.candidate_targets(sl, want_raw)
    nll, lse, dx, tcol = ops.candidate_ce_raw(rx.to(DEV), table, Cn, sl.to(DEV).reshape(-1), seed, off, want_target=True,
                                              n_items=n_items)
    assert torch.equal(tcol.cpu(), wt.reshape(-1))
    n2, l2, d2, _ = ops.candidate_ce_raw(rx.to(DEV), table, cand=wc.to(DEV), cand_target=wt.to(DEV))
    assert torch.equal(nll, n2) and torch.equal(lse, l2) and torch.equal(dx, d2)
    cd, td = ops.candidate_draw(sl.to(DEV), n_items, Cn, seed=seed, row_offset=off)
    assert torch.equal(cd.cpu(), wc) and torch.equal(td.cpu(), wt)
    want_nll, _, want_dx = orc.candidate_ce(rx, E, wc, wt)
    np.testing.assert_allclose(nll.cpu().double().numpy(), want_nll.numpy(), rtol=2e-6, atol=2e-6)
    assert (dx.cpu().double() - want_dx).abs().max() <= max(2e-6, 1e-7 * Cn ** 0.5) * max(1.0, float(want_dx.abs().max()))
    if n_items > 1:   # the table-wide draw is a different set (ABI 2's behaviour = n_items None)
        full = ops.candidate_ce_raw(rx.to(DEV), table, Cn, sl.to(DEV).reshape(-1), seed, off)
        assert not torch.equal(full[1], lse)
    with pytest.raises(ValueError):
        ops.candidate_ce_raw(rx.to(DEV), table, Cn, sl.to(DEV).reshape(-1), seed, off, n_items=N + 1)
    # the autograd op and the model-level entry carry it too
    rd = rx.to(DEV).requires_grad_(True)
    loss = ops.candidate_ce(rd, table, Cn, sl.to(DEV).reshape(-1), seed, off, n_items=n_items)
    loss.backward()
    np.testing.assert_allclose(loss.item(), want_nll.mean().item(), rtol=2e-6)


@pytest.mark.parametrize("R,N,D,Cn", [(66, 3000, 64, 257), (130, 5000, 128, 1000), (9, 300, 256, 70), (5, 50, 128, 4500)])
def test_gather_kernels_on_bf16_rows(ops, R, N, D, Cn):
    """prec = bf16 (the stated arithmetic of configs 3 / 5): the fused candidate kernel and the sparse mask kernel gather rows of the
    bf16 table (half the bytes), widen them exactly and keep products and sums in fp32 - so against the oracle on the
    bf16-ROUNDED table they meet the fp32 tolerances (same draws as the fp32 mode: the streams do not depend on the precision)."""
    from pivotcvae_amd._hip import PREC_BF16
    rx, E = rnd(R, D, seed=1, scale=3.0), unit_rows(N, D, seed=2)
    Eb = E.to(torch.bfloat16).float()
    g = torch.Generator().manual_seed(3)
    cand = torch.randint(0, N, (R, Cn), generator=g)
    tgt = torch.randint(0, Cn, (R,), generator=g)
    table = ops.CatalogTable(E.to(DEV))
    tol = max(2e-6, 1e-7 * Cn ** 0.5)
    want_nll, want_lse, want_dx = orc.candidate_ce(rx, Eb, cand, tgt)
    nll, lse, dx, _ = ops.candidate_ce_raw(rx.to(DEV), table, cand=cand.to(DEV), cand_target=tgt.to(DEV), prec=PREC_BF16)
    np.testing.assert_allclose(nll.cpu().double().numpy(), want_nll.numpy(), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(lse.cpu().double().numpy(), want_lse.numpy(), rtol=2e-6, atol=2e-6)
    assert (dx.cpu().double() - want_dx).abs().max() <= tol * max(1.0, float(want_dx.abs().max()))
    f32 = ops.candidate_ce_raw(rx.to(DEV), table, cand=cand.to(DEV), cand_target=tgt.to(DEV))
    assert not torch.equal(f32[0], nll) and (f32[0] - nll).abs().max() < 3e-2     # another table, the same problem
    # drawn in-kernel: the same sets in both precisions
    feat = torch.randint(0, N, (R,), generator=g)
    a = ops.candidate_ce_raw(rx.to(DEV), table, Cn, feat.to(DEV), 5, 3, want_target=True, prec=PREC_BF16)
    b = ops.candidate_ce_raw(rx.to(DEV), table, Cn, feat.to(DEV), 5, 3, want_target=True)
    assert torch.equal(a[3], b[3])
    # the sparse mask kernel on bf16 rows against the dense masked oracle on the rounded table, the kept set rebuilt on the host
    p = min(0.02, 300.0 / N)
    tg = torch.randint(0, N, (R,), generator=g)
    nll, lse, dx = ops.catalog_ce_sparse_raw(rx.to(DEV), table, tg.to(DEV), p, seed=77, row_offset=9, prec=PREC_BF16)
    keep = philox_ref.sparse_keep_mask(R, N, p, 77, 9)
    wn, wl, wd = co.ce(rx.numpy(), Eb.numpy(), tg.numpy(), keep)
    np.testing.assert_allclose(lse.cpu().numpy(), wl, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(nll.cpu().numpy(), wn, rtol=2e-6, atol=3e-6)
    np.testing.assert_allclose(dx.cpu().numpy(), wd, rtol=2e-5, atol=2e-6)
    # ... and catalog_ce routes a masked call there ONLY when bf16 rows are asked for explicitly (round 6: a bf16 catalog
    # arithmetic alone leaves the gather kernels on the fp32 table - the reference's arithmetic)
    r2 = ops.catalog_ce_raw(rx.to(DEV), table, tg.to(DEV), keep_prob=p, seed=77, row_offset=9, prec=PREC_BF16, gather_bf16=True)
    assert torch.equal(r2[0], nll) and torch.equal(r2[2], dx)
    r3 = ops.catalog_ce_raw(rx.to(DEV), table, tg.to(DEV), keep_prob=p, seed=77, row_offset=9, prec=PREC_BF16)
    f3 = ops.catalog_ce_sparse_raw(rx.to(DEV), table, tg.to(DEV), p, seed=77, row_offset=9)
    assert torch.equal(r3[0], f3[0]) and torch.equal(r3[2], f3[2]) and not torch.equal(r3[0], nll)


def test_candidate_ce_fused_bad_ids_poison_their_row_only(ops):
    """the reference raises an index error on an id outside the table; the kernel cannot raise: that row's outputs are NaN, every
    other row is untouched, nothing is read out of bounds"""
    R, N, D, Cn = 12, 50, 64, 33
    rx, E = rnd(R, D, seed=1), unit_rows(N, D, seed=2)
    g = torch.Generator().manual_seed(3)
    cand = torch.randint(0, N, (R, Cn), generator=g)
    tgt = torch.randint(0, Cn, (R,), generator=g)
    table = ops.CatalogTable(E.to(DEV))
    ref = ops.candidate_ce_raw(rx.to(DEV), table, cand=cand.to(DEV), cand_target=tgt.to(DEV))
    cand2, tgt2 = cand.clone(), tgt.clone()
    cand2[2, 5], cand2[7, 0], tgt2[4], tgt2[9] = N, -1, Cn, -3
    got = ops.candidate_ce_raw(rx.to(DEV), table, cand=cand2.to(DEV), cand_target=tgt2.to(DEV))
    badrows = torch.tensor([2, 4, 7, 9])
    good = torch.ones(R, dtype=torch.bool)
    good[badrows] = False
    for a, b in zip(got[:3], ref[:3]):
        assert torch.isnan(a.cpu()[badrows]).all() and torch.equal(a.cpu()[good], b.cpu()[good])
    feat = torch.randint(0, N, (R,), generator=g)
    feat[3] = N + 5      # drawn mode: a true item outside the catalog
    d = ops.candidate_ce_raw(rx.to(DEV), table, Cn, feat.to(DEV), 1, 0)
    assert torch.isnan(d[0].cpu()[3]) and torch.isnan(d[2].cpu()[3]).all() and torch.isfinite(d[0].cpu()[good & (torch.arange(R) != 3)]).all()
    from pivotcvae_amd import _hip
    rc = _hip.lib().pcvae_candidate_ce(_hip.ptr(rx.to(DEV)), R, _hip.ptr(table.weight), _hip.PREC_F32, N, D, Cn, None, 0, 0, None, None,
                                       _hip.ptr(torch.empty(R, device=DEV)), None, None, 1.0, None, None, 0, _hip.stream())
    assert rc == -1 and b"candidate_ce" in _hip.lib().pcvae_last_error()
    # (ABI 3) an id range larger than the table is refused on the host, before any launch
    feat_d = feat.clamp(max=N - 1).to(DEV)
    rc = _hip.lib().pcvae_candidate_ce(_hip.ptr(rx.to(DEV)), R, _hip.ptr(table.weight), _hip.PREC_F32, N, D, Cn, _hip.ptr(feat_d), 0, 0,
                                       None, None, _hip.ptr(torch.empty(R, device=DEV)), None, None, 1.0, None, None, N + 1, _hip.stream())
    assert rc == -1 and b"n_items" in _hip.lib().pcvae_last_error()


# ------------------------------------------------------------------------- argument validation
def test_bad_arguments_are_rejected_before_launch(ops):
    from pivotcvae_amd import _hip
    rx = rnd(8, 24).to(DEV)  # D=24: not a width the C ABI is instantiated for (the host op pads it, see below)
    E = rnd(50, 24).to(DEV)
    nll = torch.empty(8, device=DEV)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    rc = _hip.lib().pcvae_catalog_ce(_hip.ptr(rx), 8, _hip.ptr(E), None, 50, 24, 0, 0.0,
                                     _hip.ptr(torch.zeros(8, dtype=torch.long, device=DEV)), 1.0, 0, 0, None, _hip.ptr(nll),
                                     None, None, _hip.ptr(ws), ws.numel(), _hip.stream())
    assert rc == -1 and b"unsupported D" in _hip.lib().pcvae_last_error()
    with pytest.raises(ValueError, match="D <= 256"):
        ops.catalog_ce_raw(rnd(8, 320).to(DEV), rnd(50, 320).to(DEV), torch.zeros(8, dtype=torch.long, device=DEV))
    with pytest.raises(RuntimeError):
        ops.linear_fwd_raw(rnd(4, 8).to(DEV), rnd(3, 9).to(DEV), None, 0)


@pytest.mark.parametrize("D", [8, 24, 100])
def test_catalog_ops_pad_unsupported_widths(ops, D):
    """The reference's default --dim is 8 (train_generative.py:302).  Widths the kernels are not instantiated for are zero-padded
    to the next supported one on the host side: zero columns change no logit (fmaf(0, 0, acc) == acc exactly) and no gradient
    component of the real columns.  CE / gradient / greedy ids / sampler against the oracle at the ORIGINAL width."""
    R, N = 70, 333
    rx, E = rnd(R, D, seed=1, scale=2.0), unit_rows(N, D, seed=2)
    tgt = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(3))
    table = ops.CatalogTable(E.to(DEV))
    nll, lse, dx = ops.catalog_ce_raw(rx.to(DEV), table, tgt.to(DEV))
    wn, wl, wd = co.ce(rx.numpy(), E.numpy(), tgt.numpy())
    assert dx.shape == (R, D)
    np.testing.assert_allclose(lse.cpu().numpy(), wl, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(nll.cpu().numpy(), wn, rtol=2e-6, atol=3e-6)
    np.testing.assert_allclose(dx.cpu().numpy(), wd, rtol=2e-5, atol=2e-6)
    idx, best = ops.catalog_argmax(rx.to(DEV), table, return_best=True)
    oidx, obest = co.argmax(rx.numpy(), E.numpy())
    assert np.array_equal(idx.cpu().numpy(), oidx) and np.array_equal(best.cpu().numpy(), obest)
    smp = ops.catalog_sample(rx.to(DEV), table, seed=5)
    assert smp.shape == (R,) and int(smp.min()) >= 0 and int(smp.max()) < N


# ----------------------------------------------------------------- BASELINE.json full catalog size
@pytest.mark.parametrize("prec_name", ["f32", "bf16"])
def test_full_size_catalog_properties(ops, prec_name):
    """N = 1M, D = 128 (the north-star table) with a small row count: size-independent properties.
      * lse / nll / gradient direction agree with a chunked dense computation done by torch ON THE DEVICE
        (fp64 accumulation of fp32 logits) - an independent path through the same data;
      * sum_n softmax_n = 1  <=>  dx + E[target] = sum_n p_n E_n has norm <= max ||E_n|| = 1;
      * greedy decode is idempotent on table rows: argmax_n <E_i, E_n> = i (unit-norm, distinct rows).
    """
    from pivotcvae_amd._hip import PREC_NAMES
    N, D, R = 1_000_000, 128, 96
    g = torch.Generator(device=DEV).manual_seed(5)
    E = torch.rand(N, D, device=DEV, generator=g) * 2 - 1
    E = E / E.norm(dim=1, keepdim=True)
    rx = (torch.rand(R, D, device=DEV, generator=g) * 2 - 1) * 1.5
    tgt = torch.randint(0, N, (R,), device=DEV, generator=g)
    table = ops.CatalogTable(E)
    nll, lse, dx = ops.catalog_ce_raw(rx, table, tgt, prec=PREC_NAMES[prec_name])
    # chunked dense reference on the device
    m = torch.full((R,), -float("inf"), device=DEV, dtype=torch.float64)
    ssum = torch.zeros(R, device=DEV, dtype=torch.float64)
    num = torch.zeros(R, D, device=DEV, dtype=torch.float64)
    for c0 in range(0, N, 125_000):
        lg = (rx @ E[c0:c0 + 125_000].t()).double()
        mn = torch.maximum(m, lg.max(1)[0])
        sc = torch.exp(m - mn)
        pe = torch.exp(lg - mn[:, None])
        ssum = ssum * sc + pe.sum(1)
        num = num * sc[:, None] + pe @ E[c0:c0 + 125_000].double()
        m = mn
    want_lse = m + torch.log(ssum)
    zt = (rx.double() * E[tgt].double()).sum(1)
    want_dx = num / ssum[:, None] - E[tgt].double()
    tol = dict(f32=(2e-6, 2e-5), bf16=(2e-3, 2e-2))[prec_name]
    torch.testing.assert_close(lse.double(), want_lse, rtol=tol[0], atol=tol[0] * 10)
    torch.testing.assert_close(nll.double(), want_lse - zt, rtol=tol[0], atol=max(tol[0] * 10, 2e-5) if prec_name == "f32" else 3e-2)
    assert (dx.double() - want_dx).abs().max() < tol[1] * want_dx.abs().max()
    assert float((dx + E[tgt] if prec_name == "f32" else dx + E[tgt].to(torch.bfloat16).float()).norm(dim=1).max()) <= 1.0 + 1e-3
    if prec_name == "f32":
        pick = torch.randint(0, N, (64,), device=DEV, generator=g)
        idx = ops.catalog_argmax(E[pick].contiguous(), table)
        assert torch.equal(idx, pick)


@pytest.mark.parametrize("R,C", [(35, 13), (300, 1000), (5, 64), (129, 65)])
def test_dense_ce(ops, R, C):
    p = rnd(R, C, seed=1, scale=4.0)
    tgt = torch.randint(0, C, (R,), generator=torch.Generator().manual_seed(2))
    pr = p.clone().requires_grad_(True)
    want = torch.nn.functional.cross_entropy(pr, tgt)
    (want * 0.7).backward()
    pd = p.to(DEV).requires_grad_(True)
    got = ops.dense_ce(pd, tgt.to(DEV))
    (got * 0.7).backward()
    np.testing.assert_allclose(got.item(), want.item(), rtol=2e-6)
    torch.testing.assert_close(pd.grad.cpu(), pr.grad, rtol=1e-5, atol=1e-8)


# ------------------------------------------------------------------ fused training-path kernels
@pytest.mark.parametrize("no_user", [False, True])
@pytest.mark.parametrize("B,S,D,Z,ncols", [(37, 5, 16, 4, 5), (130, 10, 128, 16, 10), (9, 20, 256, 16, 5), (21, 3, 6, 3, 3)])
def test_assemble_inputs(ops, B, S, D, Z, ncols, no_user):
    """condition + gathers + the reference's concatenations in one launch == the pieces put together with torch on the host
    (models/pivotcvae.py:250-258, :166, :201, :213, :231, :194); ncols < S: the in-loop evaluation's 5-column context."""
    N, NU, C = 500, 40, S + 1
    E, U = rnd(N, D, seed=1), rnd(NU, D, seed=2)
    g = torch.Generator().manual_seed(3)
    s = torch.randint(0, N, (B, S), generator=g)
    u = torch.randint(0, NU, (B, 1), generator=g)
    r = (torch.rand(B, ncols, generator=g) < 0.5).float()
    enc, pri, scm, rx = ops.assemble_inputs(E.to(DEV), None if no_user else U.to(DEV), s.to(DEV), r.to(DEV), None if no_user else u.to(DEV), Z)
    cond = orc.condition(r, S)
    assert cond.shape == (B, C)
    emb = E[s.reshape(-1)].reshape(B, S * D)
    parts = [emb, cond] + ([] if no_user else [U[u.reshape(-1)]])
    assert torch.equal(enc.cpu(), torch.cat(parts, 1))
    assert torch.equal(pri.cpu(), torch.cat(parts[1:], 1))
    tail = [cond, E[s[:, 0]]] + ([] if no_user else [U[u.reshape(-1)]])
    assert torch.equal(scm.cpu()[:, Z:], torch.cat(tail, 1))       # the z window [0, Z) is the latent kernel's
    assert torch.equal(rx.cpu()[:, :D], E[s[:, 0]])


def test_latent_packed_equals_the_unpacked_operators(ops):
    """reparametrize + KL on packed head outputs [mu | logvar], z written into a column window: same values and gradients as
    ops.latent on the four separate tensors (which the goldens pin)."""
    B, Z, W = 70, 8, 29
    ye, yp = rnd(B, 2 * Z, seed=1, scale=0.7), rnd(B, 2 * Z, seed=2, scale=0.7)
    eps = rnd(B, Z, seed=3)
    yed, ypd = ye.to(DEV).requires_grad_(True), yp.to(DEV).requires_grad_(True)
    buf = torch.full((B, W), -3.0, device=DEV)
    out, e_used, k = ops.latent_packed(yed, ypd, buf, eps.to(DEV), Z=Z)
    ts = [t.to(DEV).requires_grad_(True) for t in (ye[:, :Z].contiguous(), ye[:, Z:].contiguous(), yp[:, :Z].contiguous(), yp[:, Z:].contiguous())]
    z2, e2, k2 = ops.latent(*ts, eps.to(DEV))
    assert out.data_ptr() == buf.data_ptr() and torch.equal(out[:, :Z], z2) and torch.all(out[:, Z:] == -3.0)
    assert torch.equal(e_used, e2)
    np.testing.assert_allclose(k.item(), k2.item(), rtol=1e-6)
    gz = rnd(B, W, seed=4).to(DEV)
    torch.autograd.backward([out, k], [gz, torch.tensor(0.37, device=DEV)])
    torch.autograd.backward([z2, k2], [gz[:, :Z].contiguous(), torch.tensor(0.37, device=DEV)])
    torch.testing.assert_close(yed.grad, torch.cat([ts[0].grad, ts[1].grad], 1), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(ypd.grad, torch.cat([ts[2].grad, ts[3].grad], 1), rtol=1e-6, atol=1e-7)
    # Philox path: the same stream as ops.reparam / ops.latent
    o3, e3, _ = ops.latent_packed(ye.to(DEV), yp.to(DEV), torch.zeros(B, W, device=DEV), None, seed=5, offset=32, Z=Z)
    z4, e4 = ops.reparam(ts[0].detach(), ts[1].detach(), None, seed=5, offset=32)
    assert torch.equal(e3, e4) and torch.equal(o3[:, :Z], z4)
    # large B: the KL partials of up to 64 blocks are combined in block order (deterministic)
    Bb = 20000
    ye, yp = rnd(Bb, 2 * Z, seed=6, scale=0.5).to(DEV), rnd(Bb, 2 * Z, seed=7, scale=0.5).to(DEV)
    ks = [ops.latent_packed(ye, yp, torch.zeros(Bb, Z, device=DEV), torch.zeros(Bb, Z, device=DEV), Z=Z)[2].item() for _ in range(3)]
    assert ks[0] == ks[1] == ks[2]
    want = orc.kld(ye[:, :Z].cpu(), ye[:, Z:].cpu(), yp[:, :Z].cpu(), yp[:, Z:].cpu()).item()
    np.testing.assert_allclose(ks[0], want, rtol=1e-5)


def test_downsample_dense_is_the_documented_stream(ops):
    """train_generative.downsample on dense logits (train_generative.py:36-42) as one kernel: exactly pred * (onehot(target) OR mask)
    with the mask the host restatement of the dense masked CE's Philox stream rebuilds; backward applies the same mask."""
    from pivotcvae_amd.train_generative import downsample
    R, N = 23, 1001
    pred = rnd(R, N, seed=1, scale=3.0)
    slate = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(2))
    pd = pred.to(DEV).requires_grad_(True)
    out = downsample(pd, slate.to(DEV), n_neg=250, seed=77, row_offset=5)
    keep = torch.from_numpy(philox_ref.keep_mask(R, N, 250.0 / N, 77, 5)).bool()
    keep[torch.arange(R), slate] = True
    assert torch.equal(out.cpu(), torch.where(keep, pred, torch.zeros_like(pred)))
    assert 0.15 < keep.float().mean() < 0.35
    (out * 2.0).sum().backward()
    assert torch.equal(pd.grad.cpu(), keep.float() * 2.0)
    full = downsample(pred.to(DEV), slate.to(DEV), n_neg=N)     # n_neg == N: identity
    assert torch.equal(full.cpu(), pred)
    with pytest.raises(RuntimeError):
        downsample(pred.to(DEV), slate.to(DEV), n_neg=N + 1)


def test_split_hand_off_stress_two_thousand_grouped_launches(ops):
    """Round 3 hardening of the weight gradients' batch-split hand-off (csrc/gemm_f32.hip: sc1 stores of the partial tiles, an
    explicit s_waitcnt vmcnt(0) in every thread, barrier, an integer arrival counter, sc1 loads by the last workgroup).  The failure
    mode it replaced - and the one a missing wait would bring back - is a silently wrong gradient in a few percent of the elements,
    timing dependent.  So: > 2 000 grouped launches over random ragged (M, N, K), 1 - 3 weight gradients per launch (+ an unrelated
    forward GEMM riding along in some), split counts from 1 up to the cost model's maximum of 64, issued BACK TO BACK without any
    host synchronisation; every repetition of a launch must equal its own first result BITWISE, and the first one fp64."""
    import random
    rng = random.Random(20260)
    cases = []
    # shapes that force many splits (few output tiles, long batch), the model's own layers, and ragged everything
    forced = [(4096, 64, 64), (8192, 32, 16), (4097, 64, 65), (2000, 256, 139), (8192, 256, 1419), (1024, 1152, 256), (6000, 100, 33)]
    for i in range(42):
        probs = []
        for _ in range(rng.choice([1, 1, 2, 3])):
            if rng.random() < 0.35:
                M, N, K = rng.choice(forced)
            else:
                M = rng.choice([1, 63, 64, 65, 200, 513, 1000, 2049, 4100, 8192])
                N = rng.choice([1, 16, 33, 64, 65, 128, 256, 300])
                K = rng.choice([1, 7, 32, 64, 65, 130, 283, 400])
            probs.append((M, N, K))
        cases.append((probs, rng.random() < 0.4))
    REPS = 50
    launches = 0
    max_splits = 0
    for ci, (probs, with_fwd) in enumerate(cases):
        ins = [(rnd(M, N, seed=7000 + 10 * ci + j).to(DEV), rnd(M, K, seed=8000 + 10 * ci + j).to(DEV)) for j, (M, N, K) in enumerate(probs)]
        outs = [[(torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)) for (M, N, K) in probs] for _ in range(REPS)]
        if with_fwd:
            xf, Wf, bf = rnd(300, 70, seed=9000 + ci).to(DEV), rnd(40, 70, seed=9100 + ci).to(DEV), rnd(40, seed=9200 + ci).to(DEV)
            yf = [torch.empty(300, 40, device=DEV) for _ in range(REPS)]
        for rep in range(REPS):            # no host sync anywhere in this loop
            grp = ops.GemmGroup()
            for (g, x), (dW, db) in zip(ins, outs[rep]):
                grp.dw(g, x, dW, db)
            if with_fwd:
                grp.fwd(xf, Wf, bf, 1, out=yf[rep])
            grp.launch()
            launches += 1
        for j, ((M, N, K), (g, x)) in enumerate(zip(probs, ins)):
            max_splits = max(max_splits, min(64, (M + 63) // 64))
            first_W, first_b = outs[0][j]
            for rep in range(1, REPS):
                assert torch.equal(outs[rep][j][0], first_W), (ci, j, rep, probs)
                assert torch.equal(outs[rep][j][1], first_b), (ci, j, rep, probs)
            atol = 2e-5 * max(1.0, (M / 64.0) ** 0.5)
            torch.testing.assert_close(first_W.cpu(), (g.double().cpu().t() @ x.double().cpu()).float(), rtol=1e-4, atol=atol)
            torch.testing.assert_close(first_b.cpu(), g.double().cpu().sum(0).float(), rtol=1e-4, atol=atol)
        if with_fwd:
            for rep in range(1, REPS):
                assert torch.equal(yf[rep], yf[0])
    assert launches >= 2000 and max_splits == 64


@pytest.mark.parametrize("arith,atol0,err_vs_scale", [("bf16x3", 6e-5, 2e-5), ("bf16x6", 2e-5, 2e-6)])
@pytest.mark.parametrize("tiles", ["default", "dma_tiles_only"])
def test_linear_bf16x3_arithmetic_random_ragged_shapes(ops, tiles, monkeypatch, arith, atol0, err_vs_scale):
    """Round 3: the MLP GEMMs in bf16x3 (ops.mlp_arith(True): operands split into bf16 hi + lo in registers, three bf16 MFMAs per
    product, fp32 accumulate) - forward, input gradient (plain / accumulating / LeakyReLU-masked), weight + bias gradient, alone
    and grouped, ragged everything, against fp64.  Tolerance: rtol 1e-4 as for the exact-f32 kernel; the absolute term is 3x the
    f32 kernel's: an operand carries 16 mantissa bits (hi + lo), so a product is good to ~6e-6 relative (fp32: 6e-8) and a sum of
    n such products to ~6e-6 sqrt(n) |typical product| - measured 4.6e-4 on the 8192-slate weight gradient of enc_1 with
    uniform(-1, 1) operands, i.e. 1.5e-5 of the tensor's scale (the catalog bf16x3 kernel's gradient is held to 2e-5 of scale).
    A launch that asks for bf16x3 takes the 64 x 64 tiles (= the bf16x3 body) at any size (round 4: the arithmetic of a layer
    does not depend on the batch), so both parametrisations run the same kernels here; PCVAE_GEMM_SMALL_BELOW=0 stays as the pin
    the exact-f32 reference launches of this test use.
    Round 6, arith = "bf16x6" (ops.mlp_arith("bf16x6"): three bf16 components per operand = the fp32 value exactly, six MFMAs per
    product): the same sweep at the EXACT-F32 kernel's own tolerance (atol 2e-5 sqrt(n / 64)), and on the large layer an error
    against fp64 within the f32 kernel's bound (2e-6 of scale) - fp32-exact products on the bf16 matrix cores."""
    import random
    if tiles == "dma_tiles_only":
        monkeypatch.setenv("PCVAE_GEMM_SMALL_BELOW", "0")
    rng = random.Random(4321)
    shapes = [(8192, 256, 1419), (4096, 1152, 256), (2048, 32, 256)]
    for _ in range(24):
        shapes.append((rng.choice([1, 31, 64, 65, 200, 1000, 1300, 2100]), rng.choice([1, 16, 33, 64, 65, 130, 256]),
                       rng.choice([1, 7, 32, 33, 64, 97, 283, 300])))
    with ops.mlp_arith(arith):
        for case, (M, N, K) in enumerate(shapes):
            padx, pady = rng.choice([0, 1, 3]), rng.choice([0, 5])
            xb, Wb = rnd(M, K + padx, seed=1100 + case), rnd(N, K, seed=1200 + case, scale=0.3)
            b, g = rnd(N, seed=1300 + case), rnd(M, N + pady, seed=1400 + case)
            xd, Wd, bd, gd = xb.to(DEV)[:, padx:], Wb.to(DEV), b.to(DEV), g.to(DEV)[:, :N]
            x64, W64, g64 = xb[:, padx:].double(), Wb.double(), g[:, :N].double()
            tol = dict(rtol=1e-4, atol=atol0 * max(1.0, (max(K, N, M) / 64.0) ** 0.5))
            y64 = x64 @ W64.t() + b.double()
            want_y = torch.nn.functional.leaky_relu(y64, 0.01).float()
            y = ops.linear_fwd_raw(xd, Wd, bd, 1)
            torch.testing.assert_close(y.cpu(), want_y, **tol)
            torch.testing.assert_close(ops.linear_bwd_input_raw(gd, Wd).cpu(), (g64 @ W64).float(), **tol)
            xact = rnd(M, K, seed=1500 + case).to(DEV)   # LeakyReLU mask keyed on another activated tensor
            want_m = ((g64 @ W64) * torch.where(xact.cpu().double() > 0, 1.0, 0.01)).float()
            torch.testing.assert_close(ops.linear_bwd_input_raw(gd, Wd, xact=xact).cpu(), want_m, **tol)
            acc = torch.full((M, K), 0.25, device=DEV)
            ops.linear_bwd_input_acc_raw(gd, Wd, None, acc)
            torch.testing.assert_close(acc.cpu(), (g64 @ W64 + 0.25).float(), **tol)
            outs = []
            for rep in range(2):
                grp = ops.GemmGroup()
                assert grp.x3 and grp.mode == {"bf16x3": 1, "bf16x6": 2}[arith]
                dW, db = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
                ybuf = torch.full((M, N + 3), 9.0, device=DEV)
                grp.dw(gd, xd, dW, db)
                grp.fwd(xd, Wd, bd, 0, out=ybuf[:, 3:])
                grp.launch()
                outs.append((dW, db, ybuf))
            torch.testing.assert_close(outs[0][0].cpu(), (g64.t() @ x64).float(), **tol)
            torch.testing.assert_close(outs[0][1].cpu(), g64.sum(0).float(), **tol)
            torch.testing.assert_close(outs[0][2][:, 3:].cpu(), y64.float(), **tol)
            assert torch.equal(outs[0][2][:, :3], torch.full((M, 3), 9.0, device=DEV))
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])   # bitwise reproducible
    assert not ops.GemmGroup().x3
    # the two arithmetics really differ (the flag reaches the kernel) and agree to fp32-equivalent accuracy on a large layer
    M, N, K = 4096, 256, 1419
    xd, Wd = rnd(M, K, seed=1).to(DEV), rnd(N, K, seed=2, scale=0.1).to(DEV)
    y32 = ops.linear_fwd_raw(xd, Wd, None, 0)
    with ops.mlp_arith(arith):
        y3 = ops.linear_fwd_raw(xd, Wd, None, 0)
    with ops.mlp_arith("bf16x3" if arith == "bf16x6" else "bf16x6"):
        y_other = ops.linear_fwd_raw(xd, Wd, None, 0)
    y64 = xd.double() @ Wd.double().t()
    e32, e3 = (y32.double() - y64).abs().max().item(), (y3.double() - y64).abs().max().item()
    scale = y64.abs().max().item()
    assert not torch.equal(y32, y3) and not torch.equal(y3, y_other) and e32 < 2e-6 * scale and e3 < err_vs_scale * scale, (e32, e3, scale)


def _f32_catalog_sequence(ops, seed, cases=20):
    import random
    rng = random.Random(seed)
    for case in range(cases):
        D = rng.choice([8, 16, 24, 32, 64, 100, 128, 256])
        N = rng.choice([rng.randint(1, 70), rng.randint(71, 3000), rng.randint(3001, 80000)])
        R = rng.randint(1, max(1, min(600, 6_000_000 // N)))
        scale = rng.choice([0.5, 2.0, 6.0])
        rx, E = rnd(R, D, seed=seed + 11 + case, scale=scale), unit_rows(N, D, seed=seed + 12 + case)
        tgt = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(seed + 13 + case))
        msg = f"seed={seed} case={case} R={R} N={N} D={D} scale={scale}"
        nll, lse, dx = ops.catalog_ce_raw(rx.to(DEV), E.to(DEV), tgt.to(DEV))
        wn, wl, wd = co.ce(rx.numpy(), E.numpy(), tgt.numpy())
        np.testing.assert_allclose(lse.cpu().numpy(), wl, rtol=2e-6, atol=2e-6, err_msg=msg)
        np.testing.assert_allclose(nll.cpu().numpy(), wn, rtol=2e-6, atol=3e-6 + 2e-6 * float(np.abs(wl).max()), err_msg=msg)
        np.testing.assert_allclose(dx.cpu().numpy(), wd, rtol=2e-5, atol=2e-6, err_msg=msg)
        # greedy ids: bit-exact, plain f32 route and (for the widths that have it) the bf16-screened route
        wi, wb = co.argmax(rx.numpy(), E.numpy())
        idx, best = ops.catalog_argmax(rx.to(DEV), E.to(DEV), return_best=True, screened=False)
        np.testing.assert_array_equal(idx.cpu().numpy(), wi, err_msg=msg)
        np.testing.assert_array_equal(best.cpu().numpy(), wb, err_msg=msg)
        if D in ops.BF16_DIMS:
            idx2 = ops.catalog_argmax(rx.to(DEV), E.to(DEV), screened=True)
            np.testing.assert_array_equal(idx2.cpu().numpy(), wi, err_msg=msg + " (screened)")


def test_f32_catalog_random_shapes_fuzz(ops):
    """20 random (R, N, D, scale) - widths with and without a native kernel, catalogs from one item to 80 000, one row to 600 -
    through the exact-f32 CE (C oracle, 2e-6 / 2e-5) and both argmax routes (ids and winning scores bit-exact).
    PCVAE_FUZZ_SEEDS="1,2,.." runs other sequences as well (one-off campaigns; the default is the committed sequence)."""
    for seed in [int(v) for v in os.environ.get("PCVAE_FUZZ_SEEDS", "4242").split(",")]:
        _f32_catalog_sequence(ops, seed)


def test_catalog_ce_sparse_random_shapes_fuzz(ops):
    """ten random (R, N, D, keep probability, seed, row offset) per sequence through the exactness check of
    test_catalog_ce_sparse_is_the_documented_stream (kept sets rebuilt on the host from the documented Philox stream).
    PCVAE_FUZZ_SEEDS="1,2,.." runs other sequences as well (one-off campaigns; the default is the committed sequence)."""
    import random
    for sd in [int(v) for v in os.environ.get("PCVAE_FUZZ_SEEDS", "31337").split(",")]:
        rng = random.Random(sd)
        for case in range(10):
            D = rng.choice([8, 16, 24, 32, 64, 128, 256])
            N = rng.choice([rng.randint(1, 200), rng.randint(201, 5000), rng.randint(5001, 60000)])
            R = rng.randint(1, 48)
            p = rng.choice([0.5, 0.1, 0.02, min(0.9, 300.0 / N), min(0.9, 3000.0 / N)])
            seed, off = rng.randint(0, 2 ** 40), rng.randint(0, 2 ** 33)
            rx, E = rnd(R, D, seed=sd + case, scale=2.0), unit_rows(N, D, seed=sd + 100 + case)
            tgt = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(sd + 200 + case))
            msg = f"seed={sd} case={case} R={R} N={N} D={D} p={p:.4f}"
            nll, lse, dx = ops.catalog_ce_sparse_raw(rx.to(DEV), E.to(DEV), tgt.to(DEV), p, seed=seed, row_offset=off)
            keep = philox_ref.sparse_keep_mask(R, N, p, seed, off)
            wn, wl, wd = co.ce(rx.numpy(), E.numpy(), tgt.numpy(), keep)
            np.testing.assert_allclose(lse.cpu().numpy(), wl, rtol=2e-6, atol=2e-6, err_msg=msg)
            np.testing.assert_allclose(nll.cpu().numpy(), wn, rtol=2e-6, atol=3e-6 + 2e-6 * float(np.abs(wl).max()), err_msg=msg)
            np.testing.assert_allclose(dx.cpu().numpy(), wd, rtol=2e-5, atol=2e-6, err_msg=msg)


def test_candidate_ce_and_sampler_random_shapes_fuzz(ops):
    """ten random (R, S-less rows, N, D, Cn, seed, row offset, precision) per sequence through the fused candidate kernel - sets drawn
    in-kernel == the documented stream (host restatement + the oracle's first-hit rule), loss / lse / gradient == the oracle's
    bmm + CE + autograd in fp64 (on the bf16-rounded table in bf16 mode), the materialised route, a shard - and through the rejection
    sampler (ids == the host restatement on decision-safe rows).  PCVAE_FUZZ_SEEDS="1,2,.." runs other sequences as well."""
    import random
    from pivotcvae_amd._hip import PREC_BF16, PREC_F32
    for sd in [int(v) for v in os.environ.get("PCVAE_FUZZ_SEEDS", "2718").split(",")]:
        rng = random.Random(sd)
        for case in range(10):
            D = rng.choice([8, 16, 20, 32, 64, 128, 256])
            N = rng.choice([rng.randint(1, 40), rng.randint(41, 3000), rng.randint(3001, 70000)])
            R = rng.randint(1, 40)
            Cn = rng.choice([1, rng.randint(2, 64), rng.randint(65, 1200), rng.randint(2049, 4200)])
            seed, off = rng.randint(0, 2 ** 40), rng.randint(0, 2 ** 33)
            bf16 = D in (64, 128, 256) and rng.random() < 0.4
            prec = PREC_BF16 if bf16 else PREC_F32
            # (round 6) the draw's id range: the table's rows (None) or a dataset range n_items <= N (data_loader.py:23, :46)
            n_items = rng.choice([None, None, N, rng.randint(1, N)])
            n_draw = N if n_items is None else n_items
            msg = f"seed={sd} case={case} R={R} N={N} D={D} Cn={Cn} bf16={bf16} n_items={n_items}"
            rx, E = rnd(R, D, seed=sd + case, scale=3.0), unit_rows(N, D, seed=sd + 100 + case)
            feat = torch.randint(0, n_draw, (R,), generator=torch.Generator().manual_seed(sd + 200 + case))
            table = ops.CatalogTable(E.to(DEV))
            raw = torch.from_numpy(philox_ref.candidate_raw(R, Cn, n_draw, seed, off)).view(R, 1, Cn)
            wc, wt = orc.candidate_targets(feat.view(R, 1), raw)
            wc, wt = wc.view(R, Cn), wt.view(R)
            nll, lse, dx, tcol = ops.candidate_ce_raw(rx.to(DEV), table, Cn, feat.to(DEV), seed, off, want_target=True, prec=prec,
                                                      n_items=n_items)
            assert torch.equal(tcol.cpu(), wt), msg
            Eo = E.to(torch.bfloat16).float() if bf16 else E
            wn, wl, wd = orc.candidate_ce(rx, Eo, wc, wt)
            tol = max(2e-6, 1e-7 * Cn ** 0.5)
            np.testing.assert_allclose(lse.cpu().double().numpy(), wl.numpy(), rtol=2e-6, atol=2e-6, err_msg=msg)
            np.testing.assert_allclose(nll.cpu().double().numpy(), wn.numpy(), rtol=2e-6, atol=3e-6 + 2e-6 * float(wl.abs().max()), err_msg=msg)
            assert (dx.cpu().double() - wd).abs().max() <= tol * max(1.0, float(wd.abs().max())), msg
            g = ops.candidate_ce_raw(rx.to(DEV), table, cand=wc.to(DEV), cand_target=wt.to(DEV), prec=prec)
            assert torch.equal(g[0], nll) and torch.equal(g[2], dx), msg
            if not bf16:
                p = ops.candidate_scores(rx.to(DEV), E.to(DEV), wc.to(DEV))
                want_p = torch.bmm(E[wc].double(), rx.double().reshape(R, D, 1)).reshape(R, Cn)
                assert (p.cpu().double() - want_p).abs().max() <= 3e-6 * max(1.0, float(want_p.abs().max())), msg
            h = R // 2
            if h:
                part = ops.candidate_ce_raw(rx[h:].contiguous().to(DEV), table, Cn, feat[h:].contiguous().to(DEV), seed, off + h, prec=prec,
                                            n_items=n_items)
                assert torch.equal(part[0], nll[h:]) and torch.equal(part[2], dx[h:]), msg
            idx = ops.catalog_sample(rx.to(DEV), table, seed=seed, row_offset=off).cpu().numpy()
            want, _k, safe = philox_ref.sample_reject(rx.numpy(), E.numpy(), seed, off)
            ok = safe & (want >= 0)
            np.testing.assert_array_equal(idx[ok], want[ok], err_msg=msg)
            assert ((idx >= 0) & (idx < N)).all(), msg
