"""-m gpu: the product modules (pivotcvae_amd.models.*) against the goldens minted from the reference.

fp32 path.  Tolerances: forward tensors rtol 1e-5 / atol 1e-5; ELBO terms 1e-4 relative (the north-star
bound; observed ~1e-6); gradients rtol 2e-4; parameters after Adam steps rtol 1e-4 + atol 3e-6 (Adam divides
by sqrt(v) ~ |g|, so an early step amplifies gradient rounding); greedy item / pivot ids BIT-EXACT.
"""
import numpy as np
import pytest
import torch

from tests.gpu_util import DEV, build_from_golden, close, dev
from oracle import pivotcvae_oracle as orc
from tests.helpers import load, model_cases

pytestmark = pytest.mark.gpu
CASES = model_cases()


def _override(model, g, key):
    if hasattr(model, "pivot_override"):
        model.pivot_override = dev(g.t(key)) if g.has(key) else None


@pytest.mark.parametrize("name", CASES)
def test_forward_six_tuple_and_prior(name):
    g = load(name)
    m = build_from_golden(g)
    _override(m, g, "fwd/pivot_sample")
    with torch.no_grad():
        p, rx, z, emb, mu, lv = m.forward(dev(g.t("s")), dev(g.t("r")), u=dev(g.t("u")), eps=dev(g.t("fwd/eps")))
        pmu, plv = m.get_prior(dev(g.t("r")), dev(g.t("u")))
        cond = m.get_condition(dev(g.t("r")))
    assert torch.equal(emb.cpu(), g.t("fwd/emb"))      # a gather is a copy: exact
    assert torch.equal(cond.cpu(), g.t("fwd/cond"))
    for got, key in ((p, "p"), (rx, "rx"), (z, "z"), (mu, "z_mu"), (lv, "z_logvar"), (pmu, "pMu"), (plv, "pLogvar")):
        want = g.t("fwd/" + key)
        assert tuple(got.shape) == tuple(want.shape), key
        close(got, want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("mode", ["full", "part"])
def test_fused_loss_and_gradients(name, mode):
    g = load(name)
    m = build_from_golden(g)
    _override(m, g, f"{mode}/pivot_sample")
    keep = None
    if mode == "part":
        keep = dev(g.t("part/neg_sample"))
    loss, rec, kld = m.loss(dev(g.t("s")), dev(g.t("r")), dev(g.t("u")), g.meta["beta"],
                            eps=dev(g.t(f"{mode}/eps")), keep_mask=keep)
    loss.backward()
    np.testing.assert_allclose([loss.item(), rec.item(), kld.item()], g.a[f"{mode}/loss"], rtol=1e-4)
    prefix = "grad" if mode == "full" else "part/grad"
    want = g.sub(prefix)
    for k, prm in m.named_parameters():
        if k in want:
            close(prm.grad, want[k], rtol=2e-4, atol=2e-6)
        else:  # frozen tables and the PSM stack never get a gradient
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, k
    assert sorted(k for k in g.meta["none_grads"]) == sorted(
        k for k, prm in m.named_parameters() if k not in want)


@pytest.mark.parametrize("name", CASES)
def test_trainer_three_adam_steps(name):
    from pivotcvae_amd.train_generative import Trainer
    g = load(name)
    m = build_from_golden(g)
    tr = Trainer(m, lr=g.meta["lr"], beta=g.meta["beta"], n_neg=None)
    s, r, u = dev(g.t("s")), dev(g.t("r")), dev(g.t("u"))
    for step in range(3):
        _override(m, g, f"adam/pivot_sample{step}")
        loss, rec, kld = tr.step(s, r, u, eps=dev(g.t(f"adam/eps{step}")))
        np.testing.assert_allclose([loss.item(), rec.item(), kld.item()], g.a[f"adam/loss{step}"], rtol=1e-4)
        if step in (0, 2):
            sd = m.state_dict()
            for k, v in g.sub(f"adam/step{step + 1}").items():
                close(sd[k], v, rtol=1e-4, atol=3e-6)
    for k in g.meta["none_grads"]:  # PSM + tables bit-identical after training
        assert torch.equal(m.state_dict()[k].cpu(), g.sd[k]), k


@pytest.mark.parametrize("name", CASES)
def test_recommend_greedy_ids_bit_exact(name):
    g = load(name)
    m = build_from_golden(g)
    _override(m, g, "rec/pivot_sample")
    u = None if g.meta["no_user"] else dev(g.t("u"))
    with torch.no_grad():
        items, mu = m.recommend(dev(g.t("rec/r")), u, return_item=True, eps=dev(g.t("rec/eps")))
        rx, _ = m.recommend(dev(g.t("rec/r")), u, return_item=False, eps=dev(g.t("rec/eps")))
    assert g.a["rec/item_margin"].min() > 1e-5  # these rows are not near-ties: ids must match exactly
    np.testing.assert_array_equal(items.cpu().numpy(), g.a["rec/items"])
    close(mu, g.t("rec/z_mu"), rtol=1e-5, atol=1e-6)
    assert tuple(rx.shape) == tuple(g.a["rec/rx"].shape)
    close(rx, g.t("rec/rx"), rtol=1e-5, atol=1e-5)
    if g.has("rec/pivot"):
        np.testing.assert_array_equal(m.last_pivot.cpu().numpy(), g.a["rec/pivot"])


@pytest.mark.parametrize("name", CASES)
def test_candidate_path(name):
    from pivotcvae_amd.train_generative import get_gen_loss
    g = load(name)
    m = build_from_golden(g)
    _override(m, g, "cand/pivot_sample")
    m.candidateFlag = True
    batch = {"slates": g.a["s"], "users": g.a["u"], "responses": g.a["r"],
             "sample_candidates": g.a["cand/candidates"], "sample_targets": g.a["cand/targets"]}
    loss, rec, kld = get_gen_loss(batch, m, torch.nn.CrossEntropyLoss(), g.meta["beta"], eps=dev(g.t("cand/eps")))
    loss.backward()
    np.testing.assert_allclose([loss.item(), rec.item(), kld.item()], g.a["cand/loss"], rtol=1e-4)
    for k, v in g.sub("cand/grad").items():
        close(dict(m.named_parameters())[k].grad, v, rtol=2e-4, atol=2e-6)
    with torch.no_grad():
        p = m.forward(dev(g.t("s")), dev(g.t("r")), candidates=dev(g.t("cand/candidates")), u=dev(g.t("u")),
                      eps=dev(g.t("cand/eps")))[0]
    close(p, g.t("cand/p"), rtol=1e-5, atol=1e-5)


def test_reference_style_get_gen_loss_dense_path():
    """The reference's own get_gen_loss recipe (dense p + downsample + CrossEntropyLoss) on our forward()."""
    from pivotcvae_amd.train_generative import downsample
    g = load("pivotcvae_gt_pi_user")
    m = build_from_golden(g)
    s, r, u = dev(g.t("s")), dev(g.t("r")), dev(g.t("u"))
    pmu, plv = m.get_prior(r, u)
    pred, _, _, _, mu, lv = m.forward(s, r, u=u, eps=dev(g.t("full/eps")))
    rec = torch.nn.CrossEntropyLoss()(downsample(pred, s, n_neg=g.meta["N"]), s.reshape(-1))
    KLD = -0.5 * torch.sum(1 + lv - plv - (lv.exp() + (mu - pmu).pow(2)) / plv.exp())
    loss = rec + g.meta["beta"] * KLD
    loss.backward()
    np.testing.assert_allclose([loss.item(), rec.item(), KLD.item()], g.a["full/loss"], rtol=1e-4)
    want = g.sub("grad")
    for k, prm in m.named_parameters():
        if k in want:
            close(prm.grad, want[k], rtol=2e-4, atol=2e-6)
    with pytest.raises(RuntimeError):
        downsample(pred, s, n_neg=g.meta["N"] + 1)


def test_pickle_round_trip_and_device_attr(tmp_path):
    g = load("pivotcvae_gt_pi_user")
    m = build_from_golden(g)
    path = tmp_path / "model.pt"
    torch.save(m, open(path, "wb"))
    m2 = torch.load(open(path, "rb"), weights_only=False)
    m2.to("cpu")
    m2.device = "cpu"  # reference train_generative.py:211-212
    assert m2.docEmbed.weight.device.type == "cpu"
    for k, v in m.state_dict().items():
        assert torch.equal(v.cpu(), m2.state_dict()[k])


def test_graph_captured_step_equals_eager_step():
    """Trainer(capture_graph=True): zero-grad + forward + backward replayed from a hipGraph, eps drawn outside it
    from the same Philox stream -> same ELBO terms and parameters as the eager trainer (the per-batch tile size of the
    GEMMs may differ between the two, so 'same' is to rounding)."""
    from pivotcvae_amd.train_generative import Trainer
    g = load("pivotcvae_gt_pi_s10")
    s, r, u = dev(g.t("s")), dev(g.t("r")), dev(g.t("u"))

    def run(capture, resident, raw_write):
        m = build_from_golden(g)
        m.rng_seed = 1234
        tr = Trainer(m, lr=1e-3, beta=g.meta["beta"], capture_graph=capture, resident_batch=resident)
        s2, r2 = s.clone(), r.clone()
        stats = [[float(x) for x in tr.step(s2, r2, u)] for _ in range(4)]
        assert tr.capture_graph == capture  # capture really happened (no silent fallback)
        # an IN-PLACE change of the caller's tensors must reach the replayed graph (by the per-step copy, or - resident_batch - by
        # the tensor version counters), and so must a new tensor object
        s2[0] = s2[1]
        r2[0] = 1.0 - r2[0]
        stats.append([float(x) for x in tr.step(s2, r2, u)])
        stats.append([float(x) for x in tr.step(torch.flip(s2, [0]).contiguous(), torch.flip(r2, [0]).contiguous(), torch.flip(u, [0]).contiguous())])
        if raw_write:
            # a write the version counter cannot see (.data here; this library's own out= kernels write through raw pointers the
            # same way): the DEFAULT trainer copies its inputs every step, so a refilled persistent batch buffer is what it trains on
            stats.append([float(x) for x in tr.step(s2, r2, u)])
            v = s2._version
            s2.data[2] = s2.data[3]
            r2.data[2] = 1.0 - r2.data[2]
            assert s2._version == v
            stats.append([float(x) for x in tr.step(s2, r2, u)])
        return stats, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}

    for resident, raw_write in ((False, True), (True, False)):
        outs = [run(False, False, raw_write), run(True, resident, raw_write)]
        np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=2e-5)
        assert outs[0][0][0] != outs[0][0][1]  # eps (and the parameters) really changed from step to step
        for k in outs[0][1]:
            close(outs[1][1][k], outs[0][1][k], rtol=1e-4, atol=2e-6)


def test_trainer_reports_the_loss_an_injected_loss_fn_returned():
    """Trainer(loss_fn=...) on the device: the loss that is back-propagated is the one that is reported - also when it carries a term
    beyond rec + beta * KLD (a regulariser).  (Round 3 logged rec + beta * KLD whatever the function had returned.)"""
    from pivotcvae_amd.train_generative import Trainer
    g = load("pivotcvae_gt_pi_s10")
    s, r, u = dev(g.t("s")), dev(g.t("r")), dev(g.t("u"))
    beta = g.meta["beta"]

    def with_regulariser(m, s, r, u, **kw):
        loss, rec, kld = m.loss(s, r, u, **{k: v for k, v in kw.items() if k != "mask_seed"})
        return loss + 0.25 * rec, rec, kld

    m = build_from_golden(g)
    tr = Trainer(m, lr=1e-3, beta=beta, loss_fn=with_regulariser)
    loss, rec, kld = (float(x) for x in tr.step(s, r, u, eps=dev(g.t("full/eps"))))
    np.testing.assert_allclose(loss, 1.25 * rec + beta * kld, rtol=1e-6)
    np.testing.assert_allclose([rec, kld], g.a["full/loss"][1:], rtol=1e-4)


def test_trainer_with_a_one_rank_rccl_group_equals_the_plain_trainer():
    """The data-parallel path of Trainer.step on the GPU (one all-reduce over the flat gradient buffer, whose tail carries the
    logged ELBO terms) with a 1-rank RCCL group: bitwise the same parameters and statistics as the trainer without a group."""
    import socket
    import torch.distributed as dist
    from pivotcvae_amd.train_generative import Trainer
    g = load("pivotcvae_gt_pi_s10")
    s, r, u = dev(g.t("s")), dev(g.t("r")), dev(g.t("u"))

    def run():
        m = build_from_golden(g)
        m.rng_seed = 77
        tr = Trainer(m, lr=1e-3, beta=g.meta["beta"])
        stats = [[float(x) for x in tr.step(s, r, u)] for _ in range(3)]
        return tr, stats, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}

    _, plain_stats, plain_sd = run()
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        tr, dp_stats, dp_sd = run()
        assert tr.dist is not None and tr.world == 1
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    np.testing.assert_allclose(dp_stats, plain_stats, rtol=2e-5)
    for k in plain_sd:
        close(dp_sd[k], plain_sd[k], rtol=1e-4, atol=2e-6)


def test_g7_response_model_and_click_stats():
    """UserResponseModel_MLP.forward against the golden logits of the reference; sigmoid-sum statistics."""
    from pivotcvae_amd import ops
    from pivotcvae_amd.env.response_model import UserResponseModel_MLP, sample_users
    g = load("response_mlp")
    m = g.meta
    rm = UserResponseModel_MLP(m["N"] - 1, m["NU"] - 1, m["D"], m["S"], [(m["S"] + 1) * m["D"], m["H"], m["H"], m["S"]],
                               DEV, False)
    rm.load_state_dict(g.sd)
    rm.to(DEV)
    logits = rm(dev(g.t("s")), dev(g.t("u")))
    close(logits, g.t("logits"), rtol=1e-5, atol=1e-5)
    nc, stats = ops.click_stats(logits)
    want = torch.sigmoid(g.t("logits")).sum(1)
    close(nc, want, rtol=1e-5, atol=1e-6)
    close(stats, torch.stack([want.min(), want.mean(), want.max()]), rtol=1e-5, atol=1e-6)
    users = sample_users(rm, 20000, seed=3).cpu()
    assert users.min() >= 0 and users.max() <= m["NU"] - 1
    freq = torch.bincount(users, minlength=m["NU"]).float() / 20000
    assert (freq - 1.0 / m["NU"]).abs().max() < 0.01
    assert torch.equal(users, sample_users(rm, 20000, seed=3).cpu())


def test_g10_urm_simulators_as_evaluators():
    """URM / URM_P / URM_P_MR.forward on the device (one fused kernel) against the reference's outputs, and as the evaluator of
    the in-loop recommendation test (train_generative.py:185 applies a sigmoid on top of their probabilities)."""
    from pivotcvae_amd.env.response_model import URM, URM_P, URM_P_MR
    from pivotcvae_amd.train_generative import recommendation_test
    g = load("response_urm")
    m = g.meta
    s, u = dev(g.t("s")), dev(g.t("u"))
    models = {"p_urm": URM(m["N"] - 1, m["NU"] - 1, m["S"], m["D"], "cpu", False),
              "p_urm_p": URM_P(m["N"] - 1, m["NU"] - 1, m["S"], m["D"], "cpu", False, m["p_bias_max"], m["p_bias_min"]),
              "p_urm_p_mr": URM_P_MR(m["N"] - 1, m["NU"] - 1, m["S"], m["D"], "cpu", False, m["p_bias_max"], m["p_bias_min"],
                                     m["mr_factor"])}
    assert sorted(models["p_urm"].state_dict()) == sorted(g.sd)     # same state_dict keys as the reference classes
    for key, rm in models.items():
        rm.load_state_dict(g.sd)
        if key != "p_urm":
            close(rm.posBias, g.t("posBias"), rtol=1e-6, atol=1e-7)   # the constructor's own positional constants
            rm.posDependentBias = g.t("posDependentBias").clone()
        rm = rm.to(DEV)
        assert rm.device == DEV
        close(rm(s, u), g.t(key), rtol=1e-5, atol=1e-6)
        close(rm(s, u.reshape(-1, 1)), g.t(key), rtol=1e-5, atol=1e-6)
    # as evaluator of a generative model over the same catalog
    gm = load("pivotcvae_gt_pi_user")
    model = build_from_golden(gm)
    rm = models["p_urm_p_mr"].to(DEV)
    out = recommendation_test(model, rm, bs=32, n_test_trial=2, seed=5).cpu()
    assert tuple(out.shape) == (5, 3) and torch.all(out[:, 0] <= out[:, 1]) and torch.all(out[:, 1] <= out[:, 2])
    assert torch.all(out > 0) and torch.all(out < 5)


def test_g11_candidate_draw():
    """On-device candidate sets (data_loader.py:46-58): the rule on the reference's recorded draw (golden G11), the documented
    Philox stream (host restatement), and the properties the loss relies on."""
    from pivotcvae_amd import ops
    from tests import philox_ref
    g = load("candidate_sets")
    slates, raw = dev(g.t("slates")), dev(g.t("raw"))
    cand, tgt = ops.candidate_draw(slates, int(g.a["max_iid"]) + 1, raw.shape[-1], raw=raw)
    assert torch.equal(cand.cpu(), g.t("candidates")) and torch.equal(tgt.cpu(), g.t("targets"))
    # the in-kernel stream
    B, S, Cn, N = 37, 5, 50, 60
    gen = torch.Generator().manual_seed(3)
    sl = torch.randint(0, N, (B, S), generator=gen)
    cand, tgt = ops.candidate_draw(dev(sl), N, Cn, seed=77, row_offset=10)
    want_raw = torch.from_numpy(philox_ref.candidate_raw(B * S, Cn, N, 77, 10)).view(B, S, Cn)
    wc, wt = orc.candidate_targets(sl, want_raw)
    assert torch.equal(cand.cpu(), wc) and torch.equal(tgt.cpu(), wt)
    assert int((wt > 0).sum()) > 10 and int((wt == 0).sum()) > 10     # both branches of the rule
    assert torch.equal(torch.gather(cand.cpu(), 2, tgt.cpu()[..., None])[..., 0], sl)   # the true item sits at its target
    # a shard draws what the whole batch drew for those slots
    c2, t2 = ops.candidate_draw(dev(sl[20:]), N, Cn, seed=77, row_offset=10 + 20 * S)
    assert torch.equal(c2, cand[20:]) and torch.equal(t2, tgt[20:])
    # uniform marginals
    big, _ = ops.candidate_draw(dev(torch.zeros(400, 5, dtype=torch.long)), 100, 200, seed=5)
    freq = torch.bincount(big[:, :, 1:].reshape(-1).cpu(), minlength=100).float()
    assert (freq / freq.sum() - 0.01).abs().max() < 0.0015


def test_candidate_path_draws_its_own_sets():
    """get_gen_loss on the candidate path when the batch carries no candidate sets (the reference builds them per item in a
    Python loop, data_loader.py:46-58): drawn on the device, loss finite and differentiable, equal to feeding the same sets in."""
    from pivotcvae_amd import ops
    from pivotcvae_amd.train_generative import get_gen_loss
    g = load("pivotcvae_gt_pi_user")
    model = build_from_golden(g)
    model.candidateFlag = True
    model.nCandidate = 20
    batch = {"slates": g.a["s"], "users": g.a["u"], "responses": g.a["r"]}
    eps = dev(g.t("fwd/eps"))
    loss, rec, kld = get_gen_loss(batch, model, torch.nn.CrossEntropyLoss(), 0.001, eps=eps, seed=9)
    loss.backward()
    assert torch.isfinite(loss) and model.scm_1.weight.grad.abs().max() > 0
    cand, tgt = ops.candidate_draw(dev(g.t("s")), model.docEmbed.weight.shape[0], 20, seed=9)
    batch2 = dict(batch, sample_candidates=cand.cpu().numpy(), sample_targets=tgt.cpu().numpy())
    loss2, rec2, kld2 = get_gen_loss(batch2, model, torch.nn.CrossEntropyLoss(), 0.001, eps=eps)
    assert torch.equal(rec, rec2) and torch.equal(kld, kld2)


def test_g8_response_model_training_steps():
    """pivotcvae_amd.pretrain_env.ResponseTrainer (gather + scatter-add backward, whole-vector normalisation and its
    backward, ReLU MLP, BCE of the sigmoid, Adam with weight decay over one flat buffer incl. the tables) against the
    golden steps of the reference's loop body: logits, losses, gradients, parameters after 1 and 3 steps."""
    from pivotcvae_amd.env.response_model import UserResponseModel_MLP
    from pivotcvae_amd.pretrain_env import ResponseTrainer
    g = load("response_training")
    m = g.meta
    rm = UserResponseModel_MLP(m["N"] - 1, m["NU"] - 1, m["D"], m["S"], [(m["S"] + 1) * m["D"], m["H"], m["H"], m["S"]],
                               DEV, False)
    rm.load_state_dict(g.sd)
    rm.to(DEV)
    s, u, r = dev(g.t("s")), dev(g.t("u")), dev(g.t("r"))
    close(rm(s, u), g.t("logits0"), rtol=1e-5, atol=1e-5)              # the no-grad forward: same [B, 1] user quirk
    tr = ResponseTrainer(rm, m["lr"], m["decay"])
    tr.opt.zero_grad()
    loss = tr.loss(s, u, r)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g.a["losses"][0], rtol=1e-5)
    for k, p in rm.named_parameters():
        close(p.grad, g.t("grad/" + k), rtol=2e-4, atol=2e-7)
    assert float(rm.userEmbed.weight.grad.abs().max()) < 1e-7   # zero up to the rounding of x * (1 / |x|)
    for t in range(m["steps"]):
        loss = tr.step(s, u, r)
        np.testing.assert_allclose(loss.item(), g.a["losses"][t], rtol=2e-5)
        if t in (0, m["steps"] - 1):
            for k, v in rm.state_dict().items():
                # user table: its gradient is rounding noise + weight_decay * p (see tests/test_oracle_golden.py g8)
                if k == "userEmbed.weight":
                    close(v, g.t(f"after{t + 1}/" + k), rtol=0, atol=0.05 * m["lr"])
                else:
                    close(v, g.t(f"after{t + 1}/" + k), rtol=1e-4, atol=2e-6)


def test_response_trainer_l2_users_and_validation():
    """users of shape [B] (L2-normalised user rows, non-zero user-table gradient) against the oracle; validation loss."""
    from pivotcvae_amd.env.response_model import UserResponseModel_MLP
    from pivotcvae_amd.pretrain_env import ResponseTrainer
    g = load("response_training")
    m = g.meta
    rm = UserResponseModel_MLP(m["N"] - 1, m["NU"] - 1, m["D"], m["S"], [(m["S"] + 1) * m["D"], m["H"], m["H"], m["S"]],
                               DEV, False)
    rm.load_state_dict(g.sd)
    rm.to(DEV)
    s, u, r = g.t("s"), g.t("u").reshape(-1), g.t("r")
    wl, wg = orc.response_loss_and_grads(g.sd, s, u, r)
    tr = ResponseTrainer(rm, m["lr"], m["decay"])
    tr.opt.zero_grad()
    loss = tr.loss(dev(s), dev(u), dev(r))
    loss.backward()
    np.testing.assert_allclose(loss.item(), wl, rtol=1e-5)
    for k, p in rm.named_parameters():
        close(p.grad, wg[k], rtol=2e-4, atol=2e-7)
    assert float(rm.userEmbed.weight.grad.abs().max()) > 0
    np.testing.assert_allclose(tr.validation_loss(dev(s), dev(u), dev(r)).item(), wl, rtol=1e-5)


def test_train_response_model_loop(tmp_path):
    """pretrain_env.train_response_model end to end on a small synthetic click log: same arguments as the reference function,
    the loss falls, the best model is pickled and loads back with the reference's state_dict keys."""
    from pivotcvae_amd.pretrain_env import train_response_model

    class Log:
        def __init__(self): self.lines = []
        def log(self, s): self.lines.append(s)

    class Clicks(torch.utils.data.Dataset):      # the fields of data_loader.UserSlateResponseDataset the function reads
        def __init__(self, n, seed):
            gen = torch.Generator().manual_seed(seed)
            self.slates = torch.randint(0, 50, (n, 5), generator=gen).numpy()
            self.users = torch.randint(0, 7, (n, 1), generator=gen).numpy()
            self.resp = (torch.from_numpy(self.slates) % 3 == 0).float().numpy()   # learnable: a click iff item id % 3 == 0
            self.max_iid, self.max_uid, self.noUser = 49, 6, False
        def __len__(self): return len(self.slates)
        def __getitem__(self, i): return {"slates": self.slates[i], "users": self.users[i], "responses": self.resp[i]}

    path = str(tmp_path / "resp_model")
    log = Log()
    torch.manual_seed(0)
    model, th, vh = train_response_model(Clicks(2048, 1), Clicks(256, 2), 8, 5, [48, 32, 5], 128, 6, 1e-2, 1e-5, DEV, path, log)
    assert th[-1] < 0.6 * th[0] and vh[-1] < vh[0]
    assert any("Save best model" in l for l in log.lines)
    back = torch.load(open(path, "rb"), weights_only=False)
    assert sorted(back.state_dict()) == ["docEmbed.weight", "mlp_1.bias", "mlp_1.weight", "mlp_2.bias", "mlp_2.weight",
                                         "userEmbed.weight"]
    s = torch.from_numpy(Clicks(64, 3).slates).to(DEV)
    u = torch.zeros(64, dtype=torch.long, device=DEV)
    pred = torch.sigmoid(back.to(DEV)(s, u)) > 0.5
    assert (pred.cpu() == (s.cpu() % 3 == 0)).float().mean() > 0.9


def test_g9_offline_metrics_on_device():
    """analysis.get_coverage / get_ILS (analysis.py:5-30) against the reference's values; S != 5 against the oracle."""
    from pivotcvae_amd import analysis
    g = load("response_analysis")
    sl, E = dev(g.t("slates")), dev(g.t("E"))
    assert analysis.get_coverage(sl, g.meta["N"]) == float(g.a["coverage"])
    close(analysis.get_ILS(sl, torch.nn.Embedding.from_pretrained(E)), g.t("ils"), rtol=1e-5, atol=2e-6)
    gen = torch.Generator().manual_seed(5)
    E2 = torch.randn(5000, 128, generator=gen)
    s2 = torch.randint(0, 5000, (300, 10), generator=gen)
    close(analysis.get_ILS(dev(s2), dev(E2)), orc.ils(s2, E2), rtol=1e-5, atol=2e-6)
    assert analysis.get_coverage(dev(s2), 5000) == orc.coverage(s2, 5000)


def test_recommendation_test_matches_oracle_composition():
    """The device-side in-loop evaluation == oracle recommend + oracle response model on the same users and eps."""
    from oracle import pivotcvae_oracle as orc
    from pivotcvae_amd.env.response_model import UserResponseModel_MLP, sample_users
    from pivotcvae_amd import ops
    g = load("pivotcvae_gt_pi_user")
    gm = load("response_mlp")
    model = build_from_golden(g)
    m = gm.meta
    rm = UserResponseModel_MLP(m["N"] - 1, m["NU"] - 1, m["D"], m["S"], [(m["S"] + 1) * m["D"], m["H"], m["H"], m["S"]],
                               DEV, False)
    rm.load_state_dict(gm.sd)
    rm.to(DEV)
    bs = 16
    users = sample_users(rm, bs, seed=1)
    ctx = torch.zeros(bs, 5, device=DEV)
    ctx[:, :2] = 1
    eps = torch.randn(bs, g.meta["Z"], generator=torch.Generator().manual_seed(0))
    items, _ = model.recommend(ctx, users, return_item=True, eps=dev(eps))
    _nc, stats = ops.click_stats(rm(items.view(bs, -1), users))
    o = orc.recommend(g.sd, g.cfg(), ctx.cpu(), users.cpu().reshape(-1, 1), eps)
    want = torch.sigmoid(orc.response_mlp(gm.sd, o["items"].reshape(bs, -1), users.cpu())).sum(1)
    assert torch.equal(items.cpu(), o["items"])
    close(stats, torch.stack([want.min(), want.mean(), want.max()]), rtol=1e-5, atol=1e-6)
    # and the full loop runs (Philox users / eps inside): shapes and ranges only
    from pivotcvae_amd.train_generative import recommendation_test
    out = recommendation_test(model, rm, bs=32, n_test_trial=3, seed=5)
    assert tuple(out.shape) == (5, 3)
    o3 = out.cpu()
    assert torch.all(o3[:, 0] <= o3[:, 1]) and torch.all(o3[:, 1] <= o3[:, 2]) and torch.all(o3 >= 0) and torch.all(o3 <= 5)


@pytest.mark.parametrize("candidate", [False, True])
def test_train_on_dataset_on_the_hip_path(tmp_path, candidate):
    """the epoch loop (train_generative.py:67-214 counterpart) end to end on the device: mask-train and candidate mode, validation,
    in-loop evaluation against a URM simulator, best-model pickle moved to the CPU; the loss goes down."""
    from pivotcvae_amd.env.response_model import URM_P_MR
    from pivotcvae_amd.train_generative import train_on_dataset
    g = load("pivotcvae_gt_pi_user")
    gu = load("response_urm")
    model = build_from_golden(g)
    model.candidateFlag = candidate
    m = gu.meta
    rm = URM_P_MR(m["N"] - 1, m["NU"] - 1, m["S"], m["D"], "cpu", False, 0.2, -0.1, 0.35)
    rm.load_state_dict(gu.sd)
    rm = rm.to(DEV)
    gen = torch.Generator().manual_seed(0)
    L, N, S = 96, g.meta["N"], g.meta["S"]
    train = {"slates": torch.randint(0, N, (L, S), generator=gen).numpy(), "users": torch.randint(0, g.meta["NU"], (L, 1), generator=gen).numpy(),
             "responses": (torch.rand(L, S, generator=gen) < 0.5).float().numpy(), "nCandidate": 40}
    val = {k: v[:32] for k, v in train.items() if k != "nCandidate"}

    class Log:
        lines = []

        def log(self, msg):
            self.lines.append(msg)

    path = str(tmp_path / "gen.pkl")
    hist = train_on_dataset(train, val, model, path, Log(), rm, bs=32, epochs=3, lr=3e-3, decay=0.0, beta=0.001, n_neg=N,
                            n_test_trial=2)
    assert len(hist["train"]) == 3 and hist["train"][-1] < hist["train"][0] and all(np.isfinite(hist["val"]))
    text = "\n".join(Log.lines)
    assert "Expected response (5): " in text and "Save best model" in text and "validation Loss: " in text
    best = torch.load(open(path, "rb"), weights_only=False)
    assert best.device == "cpu" and best.candidateFlag == candidate
    assert not torch.equal(best.state_dict()["scm_1.weight"], g.sd["scm_1.weight"])
    assert torch.equal(best.state_dict()["psm_1.weight"], g.sd["psm_1.weight"])       # the PSM never trains


@pytest.mark.parametrize("name", ["pivotcvae_gt_pi_user", "pivotcvae_gt_pi_nouser", "pivotcvae_gt_pi_s10", "pivotcvae_gt_spi_user"])
def test_fused_train_path_equals_the_operator_by_operator_path(name):
    """With a trainer's flat buffers attached, loss() of the ground-truth pivot rule runs one assemble kernel, one N = 2 Z GEMM per
    pair of heads, one latent kernel and writes the slate-completion output straight into rx (models/pivotcvae.py:_loss_fused).
    Same numbers as the operator-by-operator route (FUSED_TRAIN_PATH = False): ELBO terms, every gradient, and the goldens."""
    from pivotcvae_amd import ops
    from pivotcvae_amd.optim import FlatAdam
    g = load(name)
    s, r, eps = dev(g.t("s")), dev(g.t("r")), dev(g.t("full/eps"))
    u = None if g.meta["no_user"] else dev(g.t("u"))
    res = {}
    for fused in (True, False):
        m = build_from_golden(g)
        opt = FlatAdam(m, g.meta["lr"])
        assert ops.heads_adjacent(m.encmu, m.enclogvar) and ops.heads_adjacent(m.priorMu, m.priorLogvar)
        m.FUSED_TRAIN_PATH = fused
        opt.zero_grad()
        loss, rec, kld = m.loss(s, r, u, g.meta["beta"], eps=eps)
        loss.backward()
        res[fused] = ([loss.item(), rec.item(), kld.item()], {k: p.grad.clone() for k, p in m.named_parameters() if p.requires_grad})
        # FlatAdam re-homed the parameters (heads adjacent): names, shapes and values are untouched
        for k, v in m.state_dict().items():
            assert torch.equal(v.cpu(), g.sd[k]), k
    np.testing.assert_allclose(res[True][0], res[False][0], rtol=1e-6)
    np.testing.assert_allclose(res[True][0], g.a["full/loss"], rtol=1e-4)
    want = g.sub("grad")
    for k, gf in res[False][1].items():
        close(res[True][1][k], gf, rtol=1e-5, atol=1e-8)
        if k in want:
            close(res[True][1][k], want[k], rtol=2e-4, atol=2e-6)


def test_terms_only_loss_applies_any_upstream_gradient():
    """``loss(..., terms_only=True)`` lets the catalog kernel pre-scale the gradient direction for a backward seeded
    with the Trainer's registered constant 1; ANY other upstream gradient (a scaled loss, loss / k accumulation) must still be
    applied - half the seed gives half the gradient."""
    g = load("pivotcvae_gt_pi_s10")
    s, r, u, eps = dev(g.t("s")), dev(g.t("r")), dev(g.t("u")), dev(g.t("full/eps"))
    grads = []
    for mode in ("plain", "terms_only"):
        m = build_from_golden(g)
        if mode == "plain":
            _, rec, _ = m.loss(s, r, u, g.meta["beta"], eps=eps)
            (0.5 * rec).backward()
        else:
            _, rec, kld = m.loss(s, r, u, g.meta["beta"], eps=eps, terms_only=True)
            torch.autograd.backward([rec], [torch.full((), 0.5, device=DEV)])
        grads.append({k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
    assert grads[0].keys() == grads[1].keys() and any(k.startswith("scm_") for k in grads[0])
    for k in grads[0]:
        close(grads[1][k], grads[0][k], rtol=1e-5, atol=1e-8)


def test_train_on_dataset_graph_replay_equals_eager_in_candidate_mode(tmp_path):
    """the same in the reference's DEFAULT mode (model.candidateFlag: candidate sets drawn in the fused kernel, their seed read from
    a device word by the replayed graph): histories and final parameters of the replayed and the eager epoch loop agree"""
    from pivotcvae_amd.train_generative import train_on_dataset
    g = load("pivotcvae_gt_pi_user")
    gen = torch.Generator().manual_seed(3)
    L, N, S = 256, g.meta["N"], g.meta["S"]
    train = {"slates": torch.randint(0, N, (L, S), generator=gen).numpy(), "users": torch.randint(0, g.meta["NU"], (L, 1), generator=gen).numpy(),
             "responses": (torch.rand(L, S, generator=gen) < 0.5).float().numpy(), "nCandidate": 40}
    val = {k: v[:64] for k, v in train.items() if k != "nCandidate"}

    class Log:
        lines = []

        def log(self, msg):
            self.lines.append(msg)

    res = {}
    for graph in (False, True):
        model = build_from_golden(g)
        model.candidateFlag = True
        hist = train_on_dataset(train, val, model, str(tmp_path / f"cand{int(graph)}.pkl"), Log(), None, bs=64, epochs=3, lr=3e-3,
                                decay=0.0, beta=0.001, n_neg=N, seed=11, capture_graph=graph)
        res[graph] = (hist, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    assert not any("capture failed" in l for l in Log.lines)
    np.testing.assert_allclose(res[True][0]["train"], res[False][0]["train"], rtol=2e-6)
    np.testing.assert_allclose(res[True][0]["val"], res[False][0]["val"], rtol=2e-6)
    assert res[False][0]["train"][-1] < res[False][0]["train"][0]
    for k, v in res[False][1].items():
        assert float((res[True][1][k] - v).abs().max()) <= 1e-6, k


def test_train_on_dataset_graph_replay_equals_eager(tmp_path):
    """the epoch loop with hipGraph replay of the step against eager launches: same seed, same permutations, nothing read back inside
    an epoch - histories and final parameters must agree to rounding (round 3: a memset node in the captured graph raced with the
    eager Adam launch in front of it; the step-level tests, which read every step back, never saw it)."""
    from pivotcvae_amd.train_generative import Trainer, train_on_dataset
    g = load("pivotcvae_gt_pi_user")
    gen = torch.Generator().manual_seed(3)
    L, N, S = 256, g.meta["N"], g.meta["S"]
    train = {"slates": torch.randint(0, N, (L, S), generator=gen).numpy(), "users": torch.randint(0, g.meta["NU"], (L, 1), generator=gen).numpy(),
             "responses": (torch.rand(L, S, generator=gen) < 0.5).float().numpy(), "nCandidate": 40}
    val = {k: v[:64] for k, v in train.items() if k != "nCandidate"}

    class Log:
        lines = []

        def log(self, msg):
            self.lines.append(msg)

    res = {}
    for graph in (False, True):
        model = build_from_golden(g)
        trainer = Trainer(model, lr=3e-3, beta=0.001, n_neg=None, capture_graph=graph)   # n_neg = None: the full softmax, capturable
        hist = train_on_dataset(train, val, model, str(tmp_path / f"gen{int(graph)}.pkl"), Log(), None, bs=64, epochs=4, lr=3e-3, decay=0.0,
                                beta=0.001, n_neg=N, seed=11, trainer=trainer)   # (no simulator, no eval_fn: the recommendation test is skipped)
        assert trainer.capture_failed is None and (trainer._graph is not None) == graph
        res[graph] = (hist, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    np.testing.assert_allclose(res[True][0]["train"], res[False][0]["train"], rtol=2e-6)
    np.testing.assert_allclose(res[True][0]["val"], res[False][0]["val"], rtol=2e-6)
    for k, v in res[False][1].items():
        assert float((res[True][1][k] - v).abs().max()) <= 1e-6, k
