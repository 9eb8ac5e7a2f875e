"""-m gpu: the data-parallel step ON THE HIP COMPUTE PATH with W simulated ranks on one GPU ("a multi-rank
test that runs DP on 1 GPU with W simulated ranks (split batch, sum grads on host)").

One replica plays every rank in turn: ``Trainer(world_size=W)`` runs the PRODUCT's ``local_phase`` on shard w exactly as
``Trainer.step`` would on rank w (``row_offset`` = index of the shard's first slate, ``inv_count = 1 / (B_local S W)`` folded
into the catalog merge kernels, ``eps_offset = (step B_global + row_offset) Z``, ``mask_seed = step``); the W flat gradient
buffers (+ their statistics tails) are summed - exactly what ``all_reduce(SUM)`` leaves in every rank's buffer - and
``finish_phase`` applies the one Adam step.  Replicas are identical before a step and apply the same update, so one replica
playing all ranks IS the W-rank job.  Checked against the single-process step on the whole batch:

  * ELBO terms (rtol 1e-6), every parameter gradient (<= 2e-5 of its tensor's scale), parameters after 1 and 3 steps.  The strict
    gradient bound is asserted with the GEMMs pinned to ONE tile path (PCVAE_GEMM_SMALL_BELOW=0): a row's forward activations are
    then bitwise independent of the batch size, so no LeakyReLU unit changes sign between the two jobs.  With the default tile
    choice (32 x 32 K-split tiles for small shards, another summation order) an activation within rounding of 0 flips its
    LeakyReLU' from 1 to 0.01 for ONE slate - a discrete change of that slate's contribution (~1 expected flip per 8M
    activations = one config-4 batch).  One slate's contribution to a sum over B slates is ~1 / (3 sqrt(B)) of the tensor's scale,
    so that mode is held to 1e-2 of scale at B = 512 and 2e-3 at B = 8192 - it guards the structure (a wrong 1 / W, a missing
    shard: O(1)), the pinned-path mode guards the arithmetic;
  * the in-kernel Philox streams - eps, the sparse kept set, candidate draws, sampled pivots - bitwise independent of W;
  * the fused train path and the operator-by-operator path (FUSED_TRAIN_PATH = False) under sharding;
  * once at config 4's STATED size with W = 8, B_local = 1024 (the driver's 8-GPU run, one rank at a time);
  * and as two REAL processes on the one GPU, all_reduce included (gloo as the transport: RCCL refuses two ranks on one device).

Reference semantics preserved: /root/reference/train_generative.py:59-63 (recLoss is a MEAN over B S rows, KLD a SUM over B).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
Z, H, HP, NU = 16, 256, 128, 500


def make_model(N, S, D, prec, seed=0, variant="pivotcvae_gt_pi"):
    import pivotcvae_amd as pa
    torch.manual_seed(seed)
    gen = torch.Generator(device=DEV).manual_seed(seed)
    a = (2.0 / D) ** 0.5
    doc = torch.nn.Embedding(N, D, device=DEV)
    doc.weight.data = (torch.rand(N, D, device=DEV, generator=gen) * 2 - 1) * a
    usr = torch.nn.Embedding(NU, D, device=DEV)
    usr.weight.data = (torch.rand(NU, D, device=DEV, generator=gen) * 2 - 1) * a
    C = S + 1
    m = pa.PIVOTCVAE_MODELS[variant](doc, usr, S, D, Z, C, [S * D + C + D, H, H], [Z + C + D, H, H, D],
                                     [Z + C + 2 * D, H, H, (S - 1) * D], [C + D, HP, HP], False, DEV)
    return m.set_catalog_precision(prec)


def batch(N, S, B, seed=1):
    g = torch.Generator(device=DEV).manual_seed(seed)
    s = torch.randint(0, N, (B, S), device=DEV, generator=g)
    u = torch.randint(0, NU, (B, 1), device=DEV, generator=g)
    r = (torch.rand(B, S, device=DEV, generator=g) < 0.5).float()
    return s, r, u


def simulated_step(tr, W, s, r, u, eps=None):
    """one global step of the W-rank job, this trainer playing every rank in turn -> ((loss, rec, kld), summed grads, eps used)"""
    B = s.shape[0]
    per = B // W
    acc = torch.zeros_like(tr.opt.grad_ext)
    eps_used = []
    for w in range(W):
        tr.rank = w
        sl = slice(w * per, (w + 1) * per)
        tr.local_phase(s[sl], r[sl], u[sl], None if eps is None else eps[sl], global_batch=B, row_offset=w * per)
        acc += tr.opt.grad_ext              # all_reduce(SUM) over gradients + the statistics tail
        eps_used.append(tr.model._last_eps.clone())
    tr.opt.grad_ext.copy_(acc)
    grads = tr.opt.grad.clone()
    return tr.finish_phase(), grads, torch.cat(eps_used)


def single_step(tr, s, r, u, eps=None):
    tr.local_phase(s, r, u, eps)
    grads = tr.opt.grad.clone()
    eps_used = tr.model._last_eps.clone()
    tr.reduce_phase()
    return tr.finish_phase(), grads, eps_used


STRICT, FLIP_BUDGET = 2e-5, 1e-2   # FLIP_BUDGET: one slate's LeakyReLU unit on the other side of zero, at B = 512


@pytest.fixture(params=["one_tile_path", "default_tiles"])
def tile_mode(request, monkeypatch):
    """-> gradient tolerance (fraction of the tensor's scale); see the module docstring"""
    if request.param == "one_tile_path":
        monkeypatch.setenv("PCVAE_GEMM_SMALL_BELOW", "0")
        return STRICT
    monkeypatch.delenv("PCVAE_GEMM_SMALL_BELOW", raising=False)
    return FLIP_BUDGET


def grads_close(tr, got, want, tol=2e-5):
    """per parameter tensor: max |difference| <= tol * max |gradient| of that tensor"""
    off = 0
    for p in tr.opt.params:
        n = p.numel()
        a, b = got[off:off + n], want[off:off + n]
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= tol * max(scale, 1e-30) + 1e-12, (tuple(p.shape), float((a - b).abs().max()), scale)
        off += n


def grads_close_or_kinked(tr, got, want, sd0, meta, s, r, u, eps, budget):
    """default tile choice: a shard of 64 - 256 slates takes the 32 x 32 K-split GEMM tiles, the whole batch the 64 x 64 ones -
    another summation order, so a hidden unit whose pre-activation lies within rounding of zero for some slate can take the other
    LeakyReLU slope in one of the two jobs.  Every tensor must still meet the STRICT bound unless such a unit is DEMONSTRATED: the
    pre-activations are recomputed in fp64 from the parameters both jobs started from (tests/helpers.py::leaky_kinks) and the
    offending tensor must sit at or below a kinked layer of its chain, at the kinked layer in kinked rows; never beyond `budget`."""
    from tests.helpers import explain_by_kinks, leaky_kinks
    names = {id(p): k for k, p in tr.model.named_parameters()}
    off, offenders = 0, {}
    for p in tr.opt.params:
        n = p.numel()
        a, b = got[off:off + n].reshape(p.shape), want[off:off + n].reshape(p.shape)
        scale = max(float(b.abs().max()), 1e-30)
        d = (a - b).abs()
        assert float(d.max()) <= budget * scale + 1e-12, (names[id(p)], float(d.max()), scale)
        bad = d > STRICT * scale + 1e-12
        if bad.any():
            offenders[names[id(p)]] = bad.nonzero()[:, 0].tolist()
        off += n
    if offenders:
        kinks = leaky_kinks(sd0, meta, s.cpu(), r.cpu(), u.cpu(), eps.cpu(), rel=2.0 ** -19)
        for k, rows in offenders.items():
            print("[kink] " + explain_by_kinks(k, rows, kinks, "pivotcvae") + f" ({len(rows)} entries beyond {STRICT:g} of scale)")


@pytest.mark.parametrize("prec", ["f32", "bf16x6", "bf16x3"])
@pytest.mark.parametrize("W", [2, 8])
@pytest.mark.parametrize("fused", [True, False])
def test_w_simulated_ranks_equal_the_single_process_step(W, prec, fused, tile_mode):
    """mid-size D = 128 model (the width of the bf16x3 / MFMA kernels), full-catalog softmax, in-kernel Philox eps"""
    from pivotcvae_amd.train_generative import Trainer
    N, S, D, B = 20011, 10, 128, 512
    s, r, u = batch(N, S, B)
    runs = {}
    for mode in ("single", "sharded"):
        m = make_model(N, S, D, prec)
        m.FUSED_TRAIN_PATH = fused
        m.rng_seed = 4242
        sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}   # both jobs start from these parameters
        tr = Trainer(m, lr=3e-4, beta=0.001, world_size=W if mode == "sharded" else None)
        assert tr.world == (W if mode == "sharded" else 1)
        out = []
        for _ in range(3):
            st, g, e = simulated_step(tr, W, s, r, u) if mode == "sharded" else single_step(tr, s, r, u)
            out.append(([float(x) for x in st], g, e, tr.opt.flat.clone()))
        runs[mode] = (tr, out)
    tr = runs["single"][0]
    for k, ((st1, g1, e1, p1), (stW, gW, eW, pW)) in enumerate(zip(runs["single"][1], runs["sharded"][1])):
        assert torch.equal(e1, eW), f"step {k}: eps depends on the world size"      # Philox at global slate indices: BITWISE
        np.testing.assert_allclose(stW, st1, rtol=1e-6 if k == 0 else 2e-6)          # all-reduced ELBO terms == whole-batch terms
        np.testing.assert_allclose(st1[0], st1[1] + 0.001 * st1[2], rtol=1e-6)
        if k == 0:   # same parameters on both sides: the summed shard gradients ARE the whole-batch gradient
            if tile_mode == STRICT:
                grads_close(tr, gW, g1, tol=STRICT)
            else:
                grads_close_or_kinked(tr, gW, g1, sd0, dict(S=S, D=D, no_user=False, model="pivotcvae_gt_pi"), s, r, u, e1,
                                      budget=FLIP_BUDGET)
        # Adam's first steps move a weight by ~lr * sign(g): a gradient that changes sign at rounding level moves by up to 2 lr
        diff = (pW - p1).abs()
        assert float(diff.max()) <= 2.001 * 3e-4 * (k + 1)
        # (a flipped unit perturbs the lower layers' gradients densely at ~1e-3 of scale: Adam's normalised step then differs
        # for the entries with |g| below that)
        frac = float((diff > 3e-6).float().mean())
        assert frac < (2e-3 if tile_mode == STRICT else 2e-2), f"step {k}: {frac}"
    assert runs["single"][1][0][0] != runs["single"][1][1][0]   # the steps really differ (eps, parameters)


@pytest.mark.parametrize("W", [2, 8])
def test_masked_mode_and_sampled_pivots_under_sharding(W, tile_mode):
    """the reference's default n_neg = 1000 (sparse kept-rows kernel, mask keyed by GLOBAL row) and a sampled-pivot rule (sgt:
    rejection sampler keyed by GLOBAL slate index): the sharded job reproduces the single-process one"""
    from pivotcvae_amd.train_generative import Trainer
    N, S, D, B = 50021, 10, 128, 256
    s, r, u = batch(N, S, B, seed=3)
    for variant in ("pivotcvae_gt_pi", "pivotcvae_sgt_pi"):
        res = {}
        for mode in ("single", "sharded"):
            m = make_model(N, S, D, "f32", variant=variant)
            m.rng_seed = 99
            tr = Trainer(m, lr=3e-4, beta=0.001, n_neg=1000, world_size=W if mode == "sharded" else None)
            out = []
            for _ in range(2):
                st, g, e = simulated_step(tr, W, s, r, u) if mode == "sharded" else single_step(tr, s, r, u)
                out.append(([float(x) for x in st], g, e))
            res[mode] = (tr, out)
        for k in range(2):
            st1, g1, e1 = res["single"][1][k]
            stW, gW, eW = res["sharded"][1][k]
            assert torch.equal(e1, eW)
            np.testing.assert_allclose(stW, st1, rtol=2e-6)
            if k == 0:
                grads_close(res["single"][0], gW, g1, tol=tile_mode)


@pytest.mark.parametrize("W", [2, 8])
def test_candidate_mode_and_sampled_pivots_under_sharding(W, tile_mode):
    """the reference's DEFAULT mode (candidate sets, train_generative.py:52-57) through the fused kernel - sets drawn in-kernel from
    a stream keyed by (step, GLOBAL slot) - alone and together with a sampled-pivot rule (spt: a rejection-sampled pivot from the PSM output,
    keyed by GLOBAL slate index): the sharded job reproduces the single-process one, and two steps draw different sets"""
    from pivotcvae_amd.train_generative import Trainer
    N, S, D, B = 50021, 10, 128, 256
    s, r, u = batch(N, S, B, seed=3)
    for variant in ("pivotcvae_gt_pi", "pivotcvae_spt_pi"):
        res = {}
        for mode in ("single", "sharded"):
            m = make_model(N, S, D, "f32", variant=variant)
            m.rng_seed = 99
            tr = Trainer(m, lr=3e-4, beta=0.001, n_candidate=300, world_size=W if mode == "sharded" else None)
            out = []
            for _ in range(2):
                st, g, e = simulated_step(tr, W, s, r, u) if mode == "sharded" else single_step(tr, s, r, u)
                out.append(([float(x) for x in st], g, e))
            res[mode] = (tr, out)
        for k in range(2):
            st1, g1, e1 = res["single"][1][k]
            stW, gW, eW = res["sharded"][1][k]
            assert torch.equal(e1, eW)
            np.testing.assert_allclose(stW, st1, rtol=2e-6)
            assert 0.5 * np.log(300) < st1[1] < 1.5 * np.log(300)     # a CE over 300 candidates, not over the catalog
            if k == 0:
                grads_close(res["single"][0], gW, g1, tol=tile_mode)


@pytest.mark.parametrize("mode", ["n_neg", "candidates", "sgt", "spt+candidates"])
def test_graph_replay_of_the_in_kernel_draw_modes_follows_the_eager_trajectory(mode, monkeypatch):
    """Round 5: the sparse mask kernel, the fused candidate kernel and the rejection sampler read their step-dependent word (seed /
    stream position) from DEVICE memory when a step is captured, so these modes replay as a hipGraph too: four steps of a
    captured trainer (a rank in the middle of a sharded batch: row_offset > 0) equal four eager steps - the draws are the same ones
    (same seeds, same positions), so ELBO terms agree to rounding and the parameters to Adam's step noise - and consecutive steps
    really draw differently."""
    from pivotcvae_amd.train_generative import Trainer
    monkeypatch.setenv("PCVAE_GEMM_SMALL_BELOW", "0")   # one GEMM tile path in both runs
    N, S, D, B, gb, lo = 50021, 10, 128, 64, 256, 128
    s, r, u = batch(N, S, B, seed=8)
    variant = {"sgt": "pivotcvae_sgt_pi", "spt+candidates": "pivotcvae_spt_pi"}.get(mode, "pivotcvae_gt_pi")
    kw = {"n_neg": dict(n_neg=1000), "candidates": dict(n_candidate=300), "sgt": {}, "spt+candidates": dict(n_candidate=200)}[mode]
    runs = []
    for graph in (False, True):
        m = make_model(N, S, D, "f32", variant=variant)
        m.rng_seed = 77
        tr = Trainer(m, lr=3e-4, beta=0.001, capture_graph=graph, **kw)
        out = [[float(x) for x in tr.step(s, r, u, global_batch=gb, row_offset=lo)] for _ in range(4)]
        assert tr.capture_graph == graph and (tr._graph is not None) == graph and tr.capture_failed is None
        runs.append((out, tr.opt.flat.clone(), m.last_pivot.clone()))
    (e_out, e_flat, e_piv), (g_out, g_flat, g_piv) = runs
    np.testing.assert_allclose(g_out, e_out, rtol=5e-6)
    assert e_out[0][1] != e_out[1][1] and e_out[1][1] != e_out[2][1]
    assert torch.equal(e_piv, g_piv)                          # the last step's (sampled) pivots: the same stream position
    diff = (g_flat - e_flat).abs()
    assert float(diff.max()) <= 2.001 * 3e-4 * 4 and float((diff > 3e-6).float().mean()) < 2e-3


def test_in_kernel_streams_are_bitwise_independent_of_the_sharding():
    """each stream on its own, same inputs: a shard's draws == the corresponding rows of the whole batch's draws, bit for bit"""
    from pivotcvae_amd import ops
    N, S, D, B, W = 30011, 10, 128, 64, 8
    per = B // W
    s, r, u = batch(N, S, B, seed=5)
    g = torch.Generator(device=DEV).manual_seed(6)
    E = torch.nn.functional.normalize(torch.rand(N, D, device=DEV, generator=g) - 0.5, dim=1).contiguous()
    rx = (torch.rand(B * S, D, device=DEV, generator=g) - 0.5) * 4
    tgt = s.reshape(-1)
    table = ops.CatalogTable(E)
    # eps
    full = ops.philox_normal_(torch.empty(B, Z, device=DEV), seed=7, offset=3 * B * Z)
    parts = [ops.philox_normal_(torch.empty(per, Z, device=DEV), seed=7, offset=(3 * B + w * per) * Z) for w in range(W)]
    assert torch.equal(full, torch.cat(parts))
    # sparse kept set: per-row nll / lse / dx of the masked CE (one wave per row: bitwise a function of (seed, global row, rx row))
    nll, lse, dx = ops.catalog_ce_sparse_raw(rx, table, tgt, 1000.0 / N, seed=11, row_offset=0)
    for w in range(W):
        sl = slice(w * per * S, (w + 1) * per * S)
        a, b, c = ops.catalog_ce_sparse_raw(rx[sl].contiguous(), table, tgt[sl].contiguous(), 1000.0 / N, seed=11, row_offset=w * per * S)
        assert torch.equal(a, nll[sl]) and torch.equal(b, lse[sl]) and torch.equal(c, dx[sl])
    # ... and a different seed is a different kept set
    assert not torch.equal(ops.catalog_ce_sparse_raw(rx, table, tgt, 1000.0 / N, seed=12, row_offset=0)[1], lse)
    # candidate draws
    cand, ctg = ops.candidate_draw(s, N, 50, seed=13, row_offset=0)
    for w in range(W):
        sl = slice(w * per, (w + 1) * per)
        c2, t2 = ops.candidate_draw(s[sl].contiguous(), N, 50, seed=13, row_offset=w * per * S)
        assert torch.equal(c2, cand[sl]) and torch.equal(t2, ctg[sl])
    # ... and the fused candidate kernel that draws them itself: per-row nll / lse / dx bitwise a function of (seed, global slot)
    cn, cl, cx, ct = ops.candidate_ce_raw(rx, table, 50, tgt, 13, 0, want_target=True)
    assert torch.equal(ct.view(B, S), ctg)
    for w in range(W):
        sl = slice(w * per * S, (w + 1) * per * S)
        a, b, c, _ = ops.candidate_ce_raw(rx[sl].contiguous(), table, 50, tgt[sl].contiguous(), 13, w * per * S)
        assert torch.equal(a, cn[sl]) and torch.equal(b, cl[sl]) and torch.equal(c, cx[sl])
    # sampled pivots (rejection sampling from Categorical(sigmoid(scores)))
    q = rx[:B].contiguous()
    ids = ops.catalog_sample(q, table, seed=17, row_offset=40)
    for w in range(W):
        sl = slice(w * per, (w + 1) * per)
        assert torch.equal(ops.catalog_sample(q[sl].contiguous(), table, seed=17, row_offset=40 + w * per), ids[sl])


def test_config4_stated_size_eight_ranks_of_1024_slates(tile_mode):
    """the driver's 8-GPU run, one rank at a time on one GPU: N = 1M, K = 10, D = 128, global B = 8192, bf16x6 (the headline
    arithmetic since round 4; exact-f32 MLP GEMMs) - the eight summed shard gradients and the all-reduced ELBO equal the
    single-process step on the whole batch"""
    import bench
    from pivotcvae_amd.train_generative import Trainer
    cfg = bench.CONFIGS["4"]
    B, W = cfg["B"], 8
    model, _ = bench.build_model(cfg, torch.device(DEV), "bf16x6")
    model.rng_seed = 5
    s, r, u = bench.synthetic_batch(cfg, B, torch.device(DEV))
    flat0 = None
    res = {}
    for mode in ("single", "sharded"):
        tr = Trainer(model, lr=bench.LR, beta=bench.BETA, world_size=W if mode == "sharded" else None)
        if flat0 is None:
            flat0 = tr.opt.flat.clone()
        else:
            tr.opt.flat.copy_(flat0)    # the same replica again, parameters rewound to the start
        st, g, e = simulated_step(tr, W, s, r, u) if mode == "sharded" else single_step(tr, s, r, u)
        res[mode] = (tr, [float(x) for x in st], g, e, tr.opt.flat.clone())
        del tr
    tr, st1, g1, e1, p1 = res["single"]
    _, stW, gW, eW, pW = res["sharded"]
    assert torch.equal(e1, eW)
    np.testing.assert_allclose(stW, st1, rtol=1e-6)
    # a weight gradient is an fp32 sum over 8192 slates; one process adds them in batch-split order, eight ranks add 1024 each and
    # the all-reduce adds the eight: rounding ~ 6e-8 x |partial sums| x sqrt(adds), a few 1e-5 of a tensor's scale at this size
    grads_close(tr, gW, g1, tol=5e-5 if tile_mode == STRICT else 2e-3)
    diff = (pW - p1).abs()
    assert float(diff.max()) <= 2.001 * bench.LR and float((diff > 3e-6).float().mean()) < (2e-3 if tile_mode == STRICT else 2e-2)


# ---------------------------------------------------------------------------------------------------------------------------------
# two REAL processes on one GPU: the product's Trainer.step end to end - shard, HIP compute, the one all_reduce over the flat
# gradient buffer + statistics tail, Adam - with gloo as the transport (RCCL refuses two ranks on one device; the collective call
# and everything around it are the ones the 8-GPU run makes)
def _two_proc_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    from pivotcvae_amd.train_generative import Trainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N, S, D, B = 20011, 10, 128, 512
        s, r, u = batch(N, S, B)
        m = make_model(N, S, D, "bf16x3")
        m.rng_seed = 4242
        tr = Trainer(m, lr=3e-4, beta=0.001)
        assert tr.world == world and tr.rank == rank and tr.dist is not None
        (ss, rr, uu), lo = tr.shard(s, r, u)
        stats = [[float(x) for x in tr.step(ss.contiguous(), rr.contiguous(), uu.contiguous(), global_batch=B, row_offset=lo)]
                 for _ in range(3)]
        torch.cuda.synchronize()
        out[rank] = (stats, tr.opt.flat.detach().cpu(), lo)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_processes_on_one_gpu_equal_the_single_process_step():
    import socket
    import torch.multiprocessing as mp
    from pivotcvae_amd.train_generative import Trainer
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_two_proc_worker, args=(2, port, out), nprocs=2, join=True)
    (st0, p0, lo0), (st1, p1, lo1) = out[0], out[1]
    assert (lo0, lo1) == (0, 256)
    assert st0 == st1 and torch.equal(p0, p1)            # replicas bit-identical after three steps
    N, S, D, B = 20011, 10, 128, 512
    s, r, u = batch(N, S, B)
    m = make_model(N, S, D, "bf16x3")
    m.rng_seed = 4242
    tr = Trainer(m, lr=3e-4, beta=0.001)
    want = [[float(x) for x in tr.step(s, r, u)] for _ in range(3)]
    np.testing.assert_allclose(st0, want, rtol=2e-6)
    diff = (p0 - tr.opt.flat.cpu()).abs()
    assert float(diff.max()) <= 2.001 * 3e-4 * 3 and float((diff > 3e-6).float().mean()) < 2e-3
