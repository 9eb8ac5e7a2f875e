"""-m gpu: the bf16-MFMA catalog CE kernel (PCVAE_PREC_BF16).

Two references:
  * an EMULATION of the kernel's own arithmetic in torch on the CPU (bf16-rounded operands, fp32 products and
    sums, log2-domain softmax): pins indexing / swizzle / split-merge logic tightly (lse, nll 2e-5;
    gradient 4e-3 of its scale, the only un-modelled step being the bf16 rounding of the numerators);
  * the fp32 C oracle: states what bf16 costs against the reference arithmetic (per-row nll |err| < 2.5e-2 at |rx| ~ 13,
    batch-mean relative error < 1e-4 at R >= 2048, gradient direction within 2e-2 of its scale).
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
LOG2E = 1.4426950408889634
LN2 = 0.6931471805599453


@pytest.fixture(scope="module")
def ops():
    from pivotcvae_amd import ops as _ops
    return _ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def bf16r(x):
    return x.to(torch.bfloat16).to(torch.float32)


def emulate_fast(rx, E, tgt):
    """arithmetic of the max-free fast kernel (no mask): raw exp2 of the log2-domain logits, numerators rounded to bf16,
    and BOTH the row sum and the gradient accumulated from those rounded numerators (the sum is an MFMA against ones)"""
    xs = bf16r(rx * np.float32(LOG2E))
    Eh = bf16r(E)
    s2 = (xs.double() @ Eh.double().t()).float()
    R = s2.shape[0]
    eb = bf16r(torch.exp2(s2)).double()
    L = eb.sum(1, keepdim=True)
    lse = (torch.log2(L) * LN2).squeeze(1)
    nll = lse - s2.double()[torch.arange(R), tgt] * LN2
    dx = (eb @ Eh.double()) / L - Eh.double()[tgt]
    return nll.float(), lse.float(), dx.float()


def emulate(rx, E, tgt, keep=None):
    """kernel arithmetic, minus the bf16 rounding of the softmax numerators"""
    xs = bf16r(rx * np.float32(LOG2E))
    Eh = bf16r(E)
    s2 = (xs.double() @ Eh.double().t()).float()           # log2-domain logits (products exact in fp32)
    R, N = s2.shape
    k = torch.ones(R, N, dtype=torch.bool) if keep is None else keep.bool().clone()
    k[torch.arange(R), tgt] = True
    z2 = torch.where(k, s2, torch.zeros_like(s2)).double()
    m = z2.max(1, keepdim=True)[0]
    e = torch.exp2(z2 - m)
    L = e.sum(1, keepdim=True)
    lse = ((m + torch.log2(L)) * LN2).squeeze(1)
    nll = lse - z2[torch.arange(R), tgt] * LN2
    pk = torch.where(k, e, torch.zeros_like(e)) / L
    dx = pk @ Eh.double() - Eh.double()[tgt]
    return nll.float(), lse.float(), dx.float()


SHAPES = [(70, 1000, 64), (300, 4099, 128), (257, 9000, 128), (64, 333, 128), (600, 20000, 64), (300, 4099, 256),
          (520, 9000, 256), (33, 100, 256), (130, 50000, 64), (257, 70000, 128), (260, 40001, 256), (256, 32, 128),
          (256, 2048 + 96, 128), (100, 300000, 128), (100, 300000, 64)]


@pytest.mark.parametrize("pipelined", [False, True])
@pytest.mark.parametrize("R,N,D", SHAPES)
def test_bf16_ce_matches_its_own_arithmetic(ops, R, N, D, pipelined, monkeypatch):
    from pivotcvae_amd._hip import PREC_BF16
    # the software-pipelined kernels (64 rows per wave) normally take only long catalog ranges (>= 2048 tiles per
    # workgroup); PCVAE_PIPE_MIN_TILES (read at every launch) forces them onto these small shapes: fill, drain, fenced last
    # slots and the ragged tail are then most of the work
    monkeypatch.setenv("PCVAE_PIPE_MIN_TILES", "1" if pipelined else "1000000000")
    rx, E = rnd(R, D, seed=1, scale=2.0 * (128.0 / D) ** 0.5), orc.normalize_rows(rnd(N, D, seed=2))
    tgt = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(3))
    tgt[0], tgt[-1] = 0, N - 1
    nll, lse, dx = ops.catalog_ce_raw(rx.to(DEV), E.to(DEV), tgt.to(DEV), prec=PREC_BF16)
    # |rx| ~ 13: every row block passes the logit bound, so this call runs the max-free fast kernel
    wn, wl, wd = emulate_fast(rx, E, tgt)
    torch.testing.assert_close(lse.cpu(), wl, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(nll.cpu(), wn, rtol=2e-5, atol=3e-5)
    assert (dx.cpu() - wd).abs().max() < 1e-3 * wd.abs().max()
    nll2, _, none = ops.catalog_ce_raw(rx.to(DEV), E.to(DEV), tgt.to(DEV), prec=PREC_BF16, want_dx=False)
    # the loss-only call runs the lazy-max kernel (32x32x16 MFMA, fp32 row sums of the unrounded numerators), the
    # training call the max-free 16x16x32 kernel (row sums of the bf16 numerators): same logits, row sums that differ
    # by the zero-mean rounding noise of the numerators, ~2^-9 / sqrt(items that matter)
    assert none is None
    wn2, _, _ = emulate(rx, E, tgt)
    torch.testing.assert_close(nll2.cpu(), wn2, rtol=2e-5, atol=3e-5)
    # (bound: every numerator rounds by at most 2^-9 relative, so |d lse| <= log(1 + 2^-9) = 1.95e-3 - reached only by
    # tiny catalogs where a handful of items carry the whole sum)
    torch.testing.assert_close(nll2, nll, rtol=0, atol=2.5e-3)
    if N >= 4096:
        torch.testing.assert_close(nll2, nll, rtol=2e-4, atol=3e-4)


@pytest.mark.parametrize("R,N,D", [(70, 1000, 64), (300, 4099, 128)])
def test_bf16_ce_masks(ops, R, N, D):
    from pivotcvae_amd._hip import PREC_BF16
    rx, E = rnd(R, D, seed=4, scale=2.0), orc.normalize_rows(rnd(N, D, seed=5))
    tgt = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(6))
    keep = (torch.rand(R, N, generator=torch.Generator().manual_seed(7)) < 0.2).to(torch.uint8)
    nll, lse, dx = ops.catalog_ce_raw(rx.to(DEV), E.to(DEV), tgt.to(DEV), keep_mask=keep.to(DEV), prec=PREC_BF16)
    wn, wl, wd = emulate(rx, E, tgt, keep)
    torch.testing.assert_close(nll.cpu(), wn, rtol=2e-5, atol=3e-5)
    assert (dx.cpu() - wd).abs().max() < 4e-3 * wd.abs().max()
    seed, off, p = 77, 500, 0.1
    nll, lse, dx = ops.catalog_ce_raw(rx.to(DEV), E.to(DEV), tgt.to(DEV), keep_prob=p, seed=seed, row_offset=off,
                                      prec=PREC_BF16)
    wn, wl, wd = emulate(rx, E, tgt, torch.from_numpy(philox_ref.keep_mask(R, N, p, seed, off)))
    torch.testing.assert_close(nll.cpu(), wn, rtol=2e-5, atol=3e-5)
    assert (dx.cpu() - wd).abs().max() < 4e-3 * wd.abs().max()


@pytest.mark.parametrize("pipelined", [False, True])
@pytest.mark.parametrize("D", [64, 128, 256])
def test_bf16_ce_peaked_rows_both_kernels(ops, D, pipelined, monkeypatch):
    """rows whose softmax is dominated by one item, late / early / very negative logits.  With |rx| = 60 the logit bound
    (60 * log2 e = 86.6 <= 90) still admits the max-free fast kernel: exp2 spans 2^+-86 and everything stays normal
    fp32; scaled by 1.2 the bound fails and the row block runs the lazy running-max kernel (its raise branches)."""
    from pivotcvae_amd._hip import PREC_BF16
    # pipelined: at D = 128 every range is split between two kernels; for a row block that the lazy-max kernel owns the
    # second one must still hand the merge kernel a neutral partial (second half of the rows below, scale 1.2)
    monkeypatch.setenv("PCVAE_PIPE_MIN_TILES", "1" if pipelined else "1000000000")
    R, N = 512, 8192
    E = orc.normalize_rows(rnd(N, D, seed=2))
    rx = rnd(R, D, seed=1, scale=0.1)
    rx[3] = E[N - 5] * 60.0
    rx[4] = E[40] * 60.0
    rx[5] = -E[77] * 50.0
    rx[6] = E[5000] * 30.0 + E[100] * 20.0
    tgt = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(3))
    tgt[3], tgt[4] = N - 5, 40
    for scale, emu in ((1.0, emulate_fast), (1.2, emulate)):
        x = rx * scale
        nll, lse, dx = ops.catalog_ce_raw(x.to(DEV), E.to(DEV), tgt.to(DEV), prec=PREC_BF16)
        wn, wl, wd = emu(x, E, tgt)
        torch.testing.assert_close(lse.cpu(), wl, rtol=3e-5, atol=3e-5)
        torch.testing.assert_close(nll.cpu(), wn, rtol=3e-5, atol=1e-4)
        assert (dx.cpu() - wd).abs().max() < 4e-3 * wd.abs().max()
    # mixed: the first 256-row block stays within the bound (max-free kernels), the second is scaled past it (lazy-max)
    x = rx.clone()
    x[256:] = rx[:256] * 1.2
    nll, lse, dx = ops.catalog_ce_raw(x.to(DEV), E.to(DEV), tgt.to(DEV), prec=PREC_BF16)
    wn0, wl0, wd0 = emulate_fast(x[:256], E, tgt[:256])
    wn1, wl1, wd1 = emulate(x[256:], E, tgt[256:])
    torch.testing.assert_close(lse.cpu(), torch.cat([wl0, wl1]), rtol=3e-5, atol=3e-5)
    assert (dx.cpu() - torch.cat([wd0, wd1])).abs().max() < 4e-3 * wd0.abs().max()


def test_bf16_vs_fp32_reference_arithmetic(ops):
    """what bf16 costs against the fp32 oracle: zero-mean per-row noise, batch mean within 1e-4 relative"""
    from pivotcvae_amd._hip import PREC_BF16
    R, N, D = 2048, 4099, 128
    rx, E = rnd(R, D, seed=1, scale=2.0), orc.normalize_rows(rnd(N, D, seed=2))
    tgt = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(3))
    nll, lse, dx = ops.catalog_ce_raw(rx.to(DEV), E.to(DEV), tgt.to(DEV), prec=PREC_BF16)
    wn, wl, wd = co.ce(rx.numpy(), E.numpy(), tgt.numpy())
    err = nll.cpu().numpy().astype(np.float64) - wn
    assert np.abs(err).max() < 2.5e-2  # |rx| ~ 13 here: one bf16 product error is ~13 * 2^-9
    assert abs(err.mean()) / wn.mean() < 1e-4
    assert np.abs(dx.cpu().numpy() - wd).max() < 2e-2 * np.abs(wd).max()


def test_model_level_bf16_elbo(ops):
    """PivotCVAE.loss with catalog_precision=bf16 against the fp32 oracle at D=64: ELBO terms within 1e-3
    (R = 640 rows only, so the per-row bf16 noise has not averaged out as it does at the bench sizes)."""
    import pivotcvae_amd as pa
    S, D, Z, N, NU, B, H, HP = 5, 64, 8, 3001, 40, 128, 64, 32
    C = S + 1
    torch.manual_seed(0)
    e_raw, u_raw = orc.synthetic_tables(N, NU, D, seed=0)
    st = dict(enc=[S * D + C + D, H, H], psm=[Z + C + D, H, H, D], scm=[Z + C + 2 * D, H, H, (S - 1) * D],
              prior=[C + D, HP, HP])
    m = pa.PIVOTCVAE_MODELS["pivotcvae_gt_pi"](torch.nn.Embedding.from_pretrained(e_raw),
                                              torch.nn.Embedding.from_pretrained(u_raw), S, D, Z, C, st["enc"],
                                              st["psm"], st["scm"], st["prior"], False, DEV)
    m.set_catalog_precision("bf16")
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    cfg = orc.Config("pivotcvae_gt_pi", S, D, Z, False, st)
    g = torch.Generator().manual_seed(1)
    s = torch.randint(0, N, (B, S), generator=g)
    u = torch.randint(0, NU, (B, 1), generator=g)
    r = (torch.rand(B, S, generator=g) < 0.5).float()
    eps = torch.randn(B, Z, generator=g)
    loss, rec, kld = m.loss(s.to(DEV), r.to(DEV), u.to(DEV), 0.001, eps=eps.to(DEV))
    loss.backward()
    (ol, orec, okld), grads = orc.loss_and_grads(sd, cfg, s, r, u, eps, 0.001)
    np.testing.assert_allclose([loss.item(), rec.item(), kld.item()], [ol, orec, okld], rtol=1e-3)
    for k, prm in m.named_parameters():
        if grads.get(k) is not None:
            gs = grads[k].abs().max()
            assert (prm.grad.cpu() - grads[k]).abs().max() < 3e-2 * gs + 1e-6, k


def test_model_recommend_screened_ids_identical(ops, monkeypatch):
    """recommend() at D=128 over a catalog large enough for the bf16-screened argmax: pivot + slot ids are identical to
    the ones of the plain f32-MFMA argmax (itself bit-exact against oracle/catalog_oracle.c), for a pi and a pt model."""
    import pivotcvae_amd as pa
    S, D, Z, N, NU, B, H, HP = 5, 128, 8, 40000, 40, 96, 64, 32
    C = S + 1
    torch.manual_seed(0)
    e_raw, u_raw = orc.synthetic_tables(N, NU, D, seed=0)
    st = dict(enc=[S * D + C + D, H, H], psm=[Z + C + D, H, H, D], scm=[Z + C + 2 * D, H, H, (S - 1) * D],
              prior=[C + D, HP, HP])
    g = torch.Generator().manual_seed(1)
    u = torch.randint(0, NU, (B, 1), generator=g).to(DEV)
    r = (torch.rand(B, S, generator=g) < 0.5).float().to(DEV)
    eps = torch.randn(B, Z, generator=g).to(DEV)
    assert N >= ops.SCREENED_MIN_ITEMS
    for name in ("pivotcvae_gt_pi", "pivotcvae_pt_pi"):
        m = pa.PIVOTCVAE_MODELS[name](torch.nn.Embedding.from_pretrained(e_raw), torch.nn.Embedding.from_pretrained(u_raw),
                                      S, D, Z, C, st["enc"], st["psm"], st["scm"], st["prior"], False, DEV)
        m.set_catalog_precision("bf16")  # the loss precision must not leak into the ids
        with torch.no_grad():
            items_s, _ = m.recommend(r, u, return_item=True, eps=eps)
            monkeypatch.setattr(ops, "SCREENED_MIN_ITEMS", 1 << 62)
            items_f, _ = m.recommend(r, u, return_item=True, eps=eps)
            monkeypatch.undo()
        assert items_s.dtype == torch.int64 and items_s.numel() == B * S
        assert torch.equal(items_s, items_f), name


def test_bf16_ce_random_shapes_every_kernel_choice(ops, monkeypatch):
    """Range bookkeeping under stress: random (R, N, D) with catalogs that are not multiples of anything, with the pipelined
    kernels forced on (fill slot / steady trips / fenced last slots / drain / ragged tail / the D = 128 range split between
    two kernels) and forced off, against the emulation of the kernels' arithmetic; and the two choices against each other."""
    for seed in [int(v) for v in os.environ.get("PCVAE_FUZZ_SEEDS", "20261003").split(",")]:   # other sequences: one-off campaigns
        _bf16_ce_shape_sequence(ops, monkeypatch, seed)


def _bf16_ce_shape_sequence(ops, monkeypatch, seed):
    from pivotcvae_amd._hip import PREC_BF16
    rng = np.random.default_rng(seed)
    for case in range(18):
        D = int(rng.choice([64, 128, 256]))
        R = int(rng.integers(1, 700))
        N = int(rng.choice([rng.integers(1, 200), rng.integers(200, 6000), rng.integers(6000, 60000)]))
        rx, E = rnd(R, D, seed=100 + case, scale=2.0 * (128.0 / D) ** 0.5), orc.normalize_rows(rnd(N, D, seed=200 + case))
        tgt = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(300 + case))
        wn, wl, wd = emulate_fast(rx, E, tgt)
        outs = []
        for min_tiles in ("1", "1000000000"):
            monkeypatch.setenv("PCVAE_PIPE_MIN_TILES", min_tiles)
            nll, lse, dx = ops.catalog_ce_raw(rx.to(DEV), E.to(DEV), tgt.to(DEV), prec=PREC_BF16)
            msg = f"case {case}: R={R} N={N} D={D} min_tiles={min_tiles}"
            # the emulation rounds the SAME numerators to bf16, but torch.exp2 and v_exp_f32 differ in the last bit: a numerator on a
            # bf16 rounding boundary lands on the other side in one of the two - 2^-8 of ONE term, which a 65-item catalog does not
            # average away (12-seed campaign, round 3: one row of 531 off by 1.9e-4).  Such rows are counted, not excused: at most
            # 0.5 % of the rows (one row in a batch of fewer than 200), none by more than one flipped term can move an lse (2^-8).
            err = (lse.cpu() - wl).abs()
            off = err > 5e-5 + 3e-5 * wl.abs()
            assert int(off.sum()) <= max(1, int(0.005 * R)) and float(err.max()) <= 2.0 ** -8, \
                msg + f": {int(off.sum())} rows off, max {float(err.max()):.2e}"
            assert (dx.cpu() - wd).abs().max() < 1e-3 * wd.abs().max() + 1e-6 + (2.0 ** -8 if bool(off.any()) else 0.0), msg
            outs.append((lse, dx))
        # the two kernel choices round the same numerators with the same instruction: they agree wherever no flip is involved
        both = (outs[0][0] - outs[1][0]).abs()
        assert int((both > 3e-5 + 2e-5 * outs[1][0].abs()).sum()) <= max(1, int(0.005 * R)), f"case {case}: kernel choices disagree"


def _train_curve(prec, steps=60):
    """`steps` optimisation steps of a mid-size model (N = 20 000, S = 5, D = 64, B = 256) from a fixed init, with a fixed batch
    sequence and the in-kernel Philox eps stream (same seed and offsets for every arithmetic) -> (losses, final parameters)"""
    import pivotcvae_amd as pa
    from pivotcvae_amd.train_generative import Trainer
    S, D, Z, N, NU, B, H, HP = 5, 64, 8, 20000, 100, 256, 128, 64
    C = S + 1
    torch.manual_seed(0)
    e_raw, u_raw = orc.synthetic_tables(N, NU, D, seed=0)
    st = dict(enc=[S * D + C + D, H, H], psm=[Z + C + D, H, H, D], scm=[Z + C + 2 * D, H, H, (S - 1) * D], prior=[C + D, HP, HP])
    m = pa.PIVOTCVAE_MODELS["pivotcvae_gt_pi"](torch.nn.Embedding.from_pretrained(e_raw), torch.nn.Embedding.from_pretrained(u_raw),
                                              S, D, Z, C, st["enc"], st["psm"], st["scm"], st["prior"], False, DEV)
    m.set_catalog_precision(prec)
    tr = Trainer(m, lr=1e-3, beta=0.001)
    g = torch.Generator().manual_seed(1)
    losses = []
    for _ in range(steps):
        s = torch.randint(0, N, (B, S), generator=g).to(DEV)
        u = torch.randint(0, NU, (B, 1), generator=g).to(DEV)
        r = (torch.rand(B, S, generator=g) < 0.5).float().to(DEV)
        loss, rec, kld = tr.step(s, r, u)
        losses.append([loss.item(), rec.item(), kld.item()])
    return np.array(losses), {k: v.detach().clone() for k, v in m.state_dict().items()}


def test_bf16_training_tracks_f32_training():
    """bf16 is a reported variant of the headline, so it has to TRAIN like the reference arithmetic, not just agree on one step:
    60 Adam steps (lr 1e-3: the KL term falls from 56 to 2.4, the loss by 0.1 nat) from the same init, batches and eps in f32
    and in bf16 catalog arithmetic.  Stated bounds: loss and reconstruction term of EVERY step within 1e-4 relative - the
    north_star's ELBO tolerance - (measured 2.2e-5), the KL term within 5e-3 (measured 1.2e-3: it is a sum of small per-slate
    terms that the noisy gradients move around), final parameters within 6 % of the distance training moved them (measured
    3.3 %).  The same run in bf16x3 stays within 1e-6 / 1e-4 / 0.2 %."""
    # (every kernel of the step sums in a fixed order: two runs of the same arithmetic are bitwise equal, so the bounds below measure
    # the arithmetics.  With the fp32 atomics of round 1 this test failed once: 2e-5 between two f32 runs.)
    lf, pf = _train_curve("f32")
    _, p0 = _train_curve("f32", steps=0)
    assert lf[-1, 0] < lf[0, 0] - 0.05 and lf[-1, 2] < 0.1 * lf[0, 2]       # the run trains
    dist = sum(float((pf[k] - p0[k]).pow(2).sum()) for k in pf) ** 0.5
    for prec, tol_loss, tol_kld, tol_param in (("bf16", 1e-4, 5e-3, 0.06), ("bf16x3", 1e-6, 1e-4, 0.002)):
        lb, pb = _train_curve(prec)
        rel = np.abs(lb - lf) / np.abs(lf)
        assert rel[:, :2].max() < tol_loss, (prec, rel[:, :2].max())
        assert rel[:, 2].max() < tol_kld, (prec, rel[:, 2].max())
        moved = sum(float((pf[k] - pb[k]).pow(2).sum()) for k in pf) ** 0.5
        assert moved < tol_param * dist, (prec, moved, dist)
