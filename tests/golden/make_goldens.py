#!/usr/bin/env python3
"""Mint golden vectors by importing and running the REAL reference (/root/reference).

Runs only in the build container (the reference never travels to the GPU box); the
resulting .npz files under tests/golden/ are data: inputs + expected outputs.

    python tests/golden/make_goldens.py

What is recorded per case (sets G1..G7):
  G1 raw tables -> normalised docEmbed/userEmbed          (models/cvae.py:26-41)
  G2 forward 6-tuple + cond + prior                        (models/pivotcvae.py:229-276)
  G3 loss terms at n_neg=N and with a recorded mask        (train_generative.py:36-65)
  G4 every .grad (and which are None), params after 1 and 3 Adam steps
                                                           (train_generative.py:103,124-134)
  G5 recommend(): pivot ids, item ids, z_mu, score margins (models/pivotcvae.py:278-296)
  G6 candidate path p[R,Cn] + recLoss                      (models/pivotcvae.py:265-271)
  G7 UserResponseModel_MLP click logits                    (env/response_model.py:76-87)
  G10 URM / URM_P / URM_P_MR click scores                  (env/response_model.py:129-154, 286-295, 315-323)
  G11 candidate sets: recorded uniform draw -> first-hit / overwrite rule   (data_loader.py:46-58)

The reference draws eps / Bernoulli masks / Categorical samples from torch's global
generator.  We do not try to replay that stream on the device: the draws are RECORDED
here by wrapping the three torch entry points while the reference runs, and fed back to
the oracle / HIP path as explicit inputs.
"""
import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

np.float = float  # reference uses the removed alias (train_generative.py:110)
sys.path.insert(0, REF)
with contextlib.redirect_stdout(io.StringIO()):
    import models.pivotcvae as ref_pivot
    import models.listcvae as ref_list
    import env.response_model as ref_env
    import train_generative as ref_tg
from torch.distributions.categorical import Categorical


# ----------------------------------------------------------------------------- recorders
class Recorder:
    """Wrap normal_/bernoulli/Categorical.sample so every draw the reference makes is kept."""

    def __init__(self):
        self.eps, self.masks, self.cats = [], [], []

    def __enter__(self):
        self._normal = torch.Tensor.normal_
        self._bern = torch.bernoulli
        self._sample = Categorical.sample
        rec = self

        def normal_(t, *a, **k):
            out = rec._normal(t, *a, **k)
            rec.eps.append(out.detach().clone())
            return out

        def bernoulli(t, *a, **k):
            out = rec._bern(t, *a, **k)
            rec.masks.append(out.detach().clone())
            return out

        def sample(d, *a, **k):
            out = rec._sample(d, *a, **k)
            rec.cats.append(out.detach().clone())
            return out

        torch.Tensor.normal_ = normal_
        torch.bernoulli = bernoulli
        Categorical.sample = sample
        return self

    def __exit__(self, *exc):
        torch.Tensor.normal_ = self._normal
        torch.bernoulli = self._bern
        Categorical.sample = self._sample


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def structs(model, S, D, Z, H, HP, no_user):
    C = S + 1
    u = 0 if no_user else D
    enc = [S * D + C + u, H, H]
    prior = [C + u, HP, HP]
    if model == "listcvae":
        return dict(enc=enc, dec=[Z + C + u, H, H, S * D], prior=prior)
    return dict(enc=enc, psm=[Z + C + u, H, H, D], scm=[Z + C + D + u, H, H, (S - 1) * D], prior=prior)


def build(model, st, raw_doc, raw_user, S, D, Z, no_user):
    C = S + 1
    if model == "listcvae":
        return quiet(ref_list.UserListCVAEWithPrior, raw_doc, None if no_user else raw_user, S, D, Z, C,
                     st["enc"], st["dec"], st["prior"], no_user, "cpu")
    return quiet(ref_pivot.PIVOTCVAE_MODELS[model], raw_doc, None if no_user else raw_user, S, D, Z, C,
                 st["enc"], st["psm"], st["scm"], st["prior"], no_user, "cpu")


def top2_margin(scores):
    """scores [R, N] -> (best - second best) per row; used to qualify bit-exact id claims."""
    v, _ = torch.topk(scores, 2, dim=1)
    return (v[:, 0] - v[:, 1]).numpy()


def make_case(name, model, S, D, Z, N, NU, B, H, HP, no_user, seed, beta=0.001, lr=3e-4, n_neg_part=None):
    torch.manual_seed(seed)
    a = (2.0 / D) ** 0.5
    raw_doc = torch.nn.Embedding(N, D)
    raw_doc.weight.data.uniform_(-a, a)
    raw_user = torch.nn.Embedding(NU, D)
    raw_user.weight.data.uniform_(-a, a)
    st = structs(model, S, D, Z, H, HP, no_user)
    m = build(model, st, raw_doc, raw_user, S, D, Z, no_user)

    g = torch.Generator().manual_seed(seed + 1000)
    s = torch.randint(0, N, (B, S), generator=g)
    u = torch.randint(0, NU, (B, 1), generator=g)
    r = (torch.rand(B, S, generator=g) < 0.5).float()
    # make sure both the all-zero and the all-one response rows are exercised
    r[0] = 0.0
    r[-1] = 1.0

    out = {}
    out["raw_doc"] = raw_doc.weight.detach().numpy().copy()
    out["raw_user"] = raw_user.weight.detach().numpy().copy()
    for k, v in m.state_dict().items():
        out["sd/" + k] = v.detach().numpy().copy()
    out["s"], out["r"], out["u"] = s.numpy(), r.numpy(), u.numpy()

    # ---- G2 forward (training mode: true pivot given) + prior
    torch.manual_seed(seed + 1)
    with Recorder() as rec, torch.no_grad():
        p, rx, z, emb, z_mu, z_logvar = m.forward(s, r, u=u)
        pMu, pLogvar = m.get_prior(r, u)
        cond = m.get_condition(r)
    out["fwd/eps"] = rec.eps[0].numpy()
    if rec.cats:
        out["fwd/pivot_sample"] = rec.cats[0].numpy()
    for k, v in dict(p=p, rx=rx, z=z, emb=emb, z_mu=z_mu, z_logvar=z_logvar, pMu=pMu, pLogvar=pLogvar,
                     cond=cond).items():
        out["fwd/" + k] = v.numpy().copy()

    batch = {"slates": s.numpy(), "users": u.numpy(), "responses": r.numpy()}
    CEL = torch.nn.CrossEntropyLoss()

    # ---- G3 loss, full-catalog softmax (n_neg = N: the mask is all ones)
    torch.manual_seed(seed + 2)
    m.zero_grad()
    with Recorder() as rec:
        loss, recLoss, KLD = ref_tg.get_gen_loss(batch, m, CEL, beta, n_neg=N)
        loss.backward()
    out["full/eps"] = rec.eps[0].numpy()
    if rec.cats:
        out["full/pivot_sample"] = rec.cats[0].numpy()
    out["full/loss"] = np.array([loss.item(), recLoss.item(), KLD.item()], dtype=np.float64)
    # ---- G4 gradients (which are None is part of the contract: SURVEY 0.7)
    none_grads = []
    for k, prm in m.named_parameters():
        if prm.grad is None:
            none_grads.append(k)
        else:
            out["grad/" + k] = prm.grad.detach().numpy().copy()

    # ---- G3 loss with a recorded Bernoulli mask (n_neg < N)
    n_neg_part = n_neg_part or max(8, N // 4)
    torch.manual_seed(seed + 3)
    m.zero_grad()
    with Recorder() as rec:
        loss, recLoss, KLD = ref_tg.get_gen_loss(batch, m, CEL, beta, n_neg=n_neg_part)
        loss.backward()
    out["part/eps"] = rec.eps[0].numpy()
    if rec.cats:
        out["part/pivot_sample"] = rec.cats[0].numpy()
    out["part/neg_sample"] = rec.masks[0].numpy().astype(np.uint8)  # the raw Bernoulli draw [R, N]
    out["part/loss"] = np.array([loss.item(), recLoss.item(), KLD.item()], dtype=np.float64)
    for k, prm in m.named_parameters():
        if prm.grad is not None:
            out["part/grad/" + k] = prm.grad.detach().numpy().copy()

    # ---- G4 Adam: 3 steps on the same batch, n_neg = N, eps recorded per step
    m2 = build(model, st, raw_doc, raw_user, S, D, Z, no_user)
    m2.load_state_dict(m.state_dict())
    opt = torch.optim.Adam(m2.parameters(), lr=lr)
    torch.manual_seed(seed + 4)
    for step in range(3):
        opt.zero_grad()
        with Recorder() as rec:
            loss, recLoss, KLD = ref_tg.get_gen_loss(batch, m2, CEL, beta, n_neg=N)
        loss.backward()
        opt.step()
        out[f"adam/eps{step}"] = rec.eps[0].numpy()
        if rec.cats:
            out[f"adam/pivot_sample{step}"] = rec.cats[0].numpy()
        out[f"adam/loss{step}"] = np.array([loss.item(), recLoss.item(), KLD.item()], dtype=np.float64)
        if step in (0, 2):
            for k, v in m2.state_dict().items():
                out[f"adam/step{step + 1}/" + k] = v.detach().numpy().copy()

    # ---- G5 recommend (inference: pivot chosen by the model)
    ctx = torch.zeros(B, S)
    ctx[:, : min(3, S)] = 1.0
    ctx[0] = 0.0
    uu = None if no_user else u
    pivots = []
    hook = m.docEmbed.register_forward_hook(lambda mod, inp, o: pivots.append(inp[0].detach().clone()))
    torch.manual_seed(seed + 5)
    with Recorder() as rec, torch.no_grad():
        items, rec_mu = m.recommend(ctx, uu, return_item=True)
    hook.remove()
    out["rec/r"] = ctx.numpy()
    out["rec/eps"] = rec.eps[0].numpy()
    out["rec/items"] = items.numpy()
    out["rec/z_mu"] = rec_mu.numpy()
    if model != "listcvae":
        out["rec/pivot"] = pivots[0].numpy()
        if rec.cats:
            out["rec/pivot_sample"] = rec.cats[0].numpy()
    # rx and margins from a replay with the same eps (recommend(return_item=False))
    torch.manual_seed(seed + 5)
    with torch.no_grad():
        rx_rec, _ = m.recommend(ctx, uu, return_item=False)
        E = m.docEmbed.weight
        out["rec/rx"] = rx_rec.numpy().copy()
        out["rec/item_margin"] = top2_margin(rx_rec.reshape(-1, D) @ E.t())

    # ---- G6 candidate path (the reference's default when --mask_train is absent)
    Cn = 13
    gc = torch.Generator().manual_seed(seed + 2000)
    cand = torch.randint(0, N, (B, S, Cn), generator=gc)
    tgt = torch.zeros(B, S, dtype=torch.long)
    for b in range(B):
        for i in range(S):  # data_loader.py:46-55 semantics
            hit = (cand[b, i] == s[b, i]).nonzero()
            if len(hit) > 0:
                tgt[b, i] = hit[0, 0]
            else:
                cand[b, i, 0] = s[b, i]
    m.candidateFlag = True
    cb = dict(batch, sample_candidates=cand.numpy(), sample_targets=tgt.numpy())
    torch.manual_seed(seed + 6)
    m.zero_grad()
    with Recorder() as rec:
        loss, recLoss, KLD = ref_tg.get_gen_loss(cb, m, CEL, beta)
        loss.backward()
    with torch.no_grad():
        torch.manual_seed(seed + 6)
        pc = m.forward(s, r, candidates=cand, u=u)[0]
    m.candidateFlag = False
    out["cand/eps"] = rec.eps[0].numpy()
    if rec.cats:
        out["cand/pivot_sample"] = rec.cats[0].numpy()
    out["cand/candidates"] = cand.numpy()
    out["cand/targets"] = tgt.numpy()
    out["cand/p"] = pc.numpy().copy()
    out["cand/loss"] = np.array([loss.item(), recLoss.item(), KLD.item()], dtype=np.float64)
    for k, prm in m.named_parameters():
        if prm.grad is not None:
            out["cand/grad/" + k] = prm.grad.detach().numpy().copy()

    meta = dict(name=name, model=model, S=S, D=D, Z=Z, N=N, NU=NU, B=B, no_user=no_user, beta=beta, lr=lr,
                n_neg_part=n_neg_part, structs=st, none_grads=none_grads, seed=seed,
                torch=torch.__version__)
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: {len(out)} arrays, none_grads={len(none_grads)}")


def make_stated_case(name, model, S, D, Z, N, NU, B, H, HP, seed, beta=0.001, lr=3e-4, n_neg_part=None, tables_from_seed=False):
    """Round 3: the reference at a BASELINE config's STATED size (configs 1 and 2 of SURVEY.md 8d) and one D = 128 case - the
    width that takes the MFMA bf16 / bf16x3 catalog kernels and the fused train path.  Same recipe as make_case but without the
    dense [R, N] logits: raw tables, state, inputs, recorded eps, the three loss terms at n_neg = N, every .grad, the trained
    parameters after 1 and 3 Adam steps, recommend() ids + pivots + top-2 margins (train_generative.py:44-65, 103, 124-134;
    models/pivotcvae.py:242-296; models/listcvae.py:134-188).  A recorded Bernoulli mask (n_neg < N) only where it is small."""
    torch.manual_seed(seed)
    a = (2.0 / D) ** 0.5
    raw_doc = torch.nn.Embedding(N, D)
    raw_doc.weight.data.uniform_(-a, a)
    raw_user = torch.nn.Embedding(NU, D)
    raw_user.weight.data.uniform_(-a, a)
    st = structs(model, S, D, Z, H, HP, False)
    m = build(model, st, raw_doc, raw_user, S, D, Z, False)
    g = torch.Generator().manual_seed(seed + 1000)
    s = torch.randint(0, N, (B, S), generator=g)
    u = torch.randint(0, NU, (B, 1), generator=g)
    r = (torch.rand(B, S, generator=g) < 0.5).float()
    r[0] = 0.0
    r[-1] = 1.0
    frozen = ("docEmbed.weight", "userEmbed.weight")
    out = {"raw_doc": raw_doc.weight.detach().numpy().copy(), "raw_user": raw_user.weight.detach().numpy().copy()}
    for k, v in m.state_dict().items():
        out["sd/" + k] = v.detach().numpy().copy()
    if tables_from_seed:
        # a catalog of 10^6 rows is 0.5 GB: the fixture keeps the SEED of the tables and fingerprints of what it produced; the loader
        # (tests/helpers.py) redraws them with the same torch calls and checks the fingerprints before anything is compared
        def table_fingerprint(a):   # the same function as in tests/helpers.py
            sub = np.asarray(a[::97], dtype=np.float64)
            mid = a.shape[0] // 2
            return [float(sub.sum()), float(np.abs(sub).sum())] + [float(v) for v in a[0, :3]] + [float(v) for v in a[-1, -3:]] + \
                   [float(v) for v in a[mid, 1:4]]
        fp = {key: table_fingerprint(out.pop(key)) for key in ("raw_doc", "raw_user", "sd/docEmbed.weight", "sd/userEmbed.weight")}
        out["tables/fingerprints"] = np.array(json.dumps(fp))
    out["s"], out["r"], out["u"] = s.numpy(), r.numpy(), u.numpy()
    batch = {"slates": s.numpy(), "users": u.numpy(), "responses": r.numpy()}
    CEL = torch.nn.CrossEntropyLoss()

    # forward pieces that are small: z_mu / z_logvar / rx / prior (the dense p is NOT kept)
    torch.manual_seed(seed + 1)
    with Recorder() as rec, torch.no_grad():
        _p, rx, z, _emb, z_mu, z_logvar = m.forward(s, r, u=u)
        pMu, pLogvar = m.get_prior(r, u)
    out["fwd/eps"] = rec.eps[0].numpy()
    for k, v in dict(rx=rx, z=z, z_mu=z_mu, z_logvar=z_logvar, pMu=pMu, pLogvar=pLogvar).items():
        out["fwd/" + k] = v.numpy().copy()
    del _p

    # loss + gradients, full-catalog softmax
    torch.manual_seed(seed + 2)
    m.zero_grad()
    with Recorder() as rec:
        loss, recLoss, KLD = ref_tg.get_gen_loss(batch, m, CEL, beta, n_neg=N)
        loss.backward()
    out["full/eps"] = rec.eps[0].numpy()
    out["full/loss"] = np.array([loss.item(), recLoss.item(), KLD.item()], dtype=np.float64)
    none_grads = []
    for k, prm in m.named_parameters():
        if prm.grad is None:
            none_grads.append(k)
        else:
            out["grad/" + k] = prm.grad.detach().numpy().copy()

    # the reference's masked mode with the draw recorded, where [R, N] is small enough to keep
    if n_neg_part and B * S * N <= 2_000_000:
        torch.manual_seed(seed + 3)
        m.zero_grad()
        with Recorder() as rec:
            loss, recLoss, KLD = ref_tg.get_gen_loss(batch, m, CEL, beta, n_neg=n_neg_part)
            loss.backward()
        out["part/eps"] = rec.eps[0].numpy()
        out["part/neg_sample"] = rec.masks[0].numpy().astype(np.uint8)
        out["part/loss"] = np.array([loss.item(), recLoss.item(), KLD.item()], dtype=np.float64)
        for k, prm in m.named_parameters():
            if prm.grad is not None:
                out["part/grad/" + k] = prm.grad.detach().numpy().copy()

    # ... and where it is not: the draw kept SPARSELY (row, column of every kept entry; ~n_neg per row) - the reference's default
    # training mode (train_generative.py:44, n_neg = 1000) at a 10^6-item catalog
    if n_neg_part and B * S * N > 2_000_000:
        torch.manual_seed(seed + 3)
        m.zero_grad()
        with Recorder() as rec:
            loss, recLoss, KLD = ref_tg.get_gen_loss(batch, m, CEL, beta, n_neg=n_neg_part)
            loss.backward()
        nz = rec.masks[0].nonzero()
        out["part/eps"] = rec.eps[0].numpy()
        out["part/neg_rows"], out["part/neg_cols"] = nz[:, 0].numpy().astype(np.int32), nz[:, 1].numpy().astype(np.int32)
        out["part/neg_shape"] = np.array([B * S, N], dtype=np.int64)
        out["part/loss"] = np.array([loss.item(), recLoss.item(), KLD.item()], dtype=np.float64)
        for k, prm in m.named_parameters():
            if prm.grad is not None:
                out["part/grad/" + k] = prm.grad.detach().numpy().copy()

    # three Adam steps
    m2 = build(model, st, raw_doc, raw_user, S, D, Z, False)
    m2.load_state_dict(m.state_dict())
    opt = torch.optim.Adam(m2.parameters(), lr=lr)
    torch.manual_seed(seed + 4)
    for step in range(3):
        opt.zero_grad()
        with Recorder() as rec:
            loss, recLoss, KLD = ref_tg.get_gen_loss(batch, m2, CEL, beta, n_neg=N)
        loss.backward()
        opt.step()
        out[f"adam/eps{step}"] = rec.eps[0].numpy()
        out[f"adam/loss{step}"] = np.array([loss.item(), recLoss.item(), KLD.item()], dtype=np.float64)
        if step in (0, 2):
            for k, v in m2.state_dict().items():
                if k not in frozen and k not in none_grads:   # frozen tables and the PSM stack are bit-unchanged (asserted below)
                    out[f"adam/step{step + 1}/" + k] = v.detach().numpy().copy()
    for k, v in m2.state_dict().items():
        if k in frozen or k in none_grads:
            assert torch.equal(v, m.state_dict()[k]), k

    # recommend
    ctx = torch.zeros(B, S)
    ctx[:, : min(3, S)] = 1.0
    ctx[0] = 0.0
    pivots = []
    hook = m.docEmbed.register_forward_hook(lambda mod, inp, o: pivots.append(inp[0].detach().clone()))
    torch.manual_seed(seed + 5)
    with Recorder() as rec, torch.no_grad():
        items, rec_mu = m.recommend(ctx, u, return_item=True)
    hook.remove()
    out["rec/r"], out["rec/eps"], out["rec/items"], out["rec/z_mu"] = ctx.numpy(), rec.eps[0].numpy(), items.numpy(), rec_mu.numpy()
    torch.manual_seed(seed + 5)
    with torch.no_grad():
        rx_rec, _ = m.recommend(ctx, u, return_item=False)
        E = m.docEmbed.weight
        out["rec/rx"] = rx_rec.numpy().copy()
        out["rec/item_margin"] = top2_margin(rx_rec.reshape(-1, D) @ E.t())
    if model != "listcvae":
        out["rec/pivot"] = pivots[0].numpy()

    meta = dict(name=name, model=model, S=S, D=D, Z=Z, N=N, NU=NU, B=B, no_user=False, beta=beta, lr=lr,
                n_neg_part=n_neg_part, structs=st, none_grads=none_grads, seed=seed, torch=torch.__version__,
                tables_from_seed=bool(tables_from_seed))
    out["meta"] = np.array(json.dumps(meta))
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {len(out)} arrays, {os.path.getsize(path) / 1e6:.1f} MB, loss {out['full/loss']}")


def make_stated():
    """round 3: configs 1 and 2 at their stated sizes + one case per width with MFMA bf16 / bf16x3 kernels (D = 128, 64, 256)"""
    make_stated_case("stated_config1_listcvae", "listcvae", S=5, D=16, Z=16, N=1000, NU=100, B=64, H=256, HP=128, seed=801,
                     n_neg_part=100)
    make_stated_case("stated_config2_gt_pi", "pivotcvae_gt_pi", S=5, D=32, Z=16, N=10000, NU=200, B=1024, H=256, HP=128, seed=802)
    make_stated_case("stated_d128_gt_pi", "pivotcvae_gt_pi", S=10, D=128, Z=16, N=5003, NU=50, B=64, H=128, HP=64, seed=803)
    # the other two widths with MFMA bf16 kernels (config 3's D = 64, config 5's D = 256 and its slate size), catalogs kept small
    make_stated_case("stated_d64_gt_pi", "pivotcvae_gt_pi", S=10, D=64, Z=16, N=2003, NU=50, B=64, H=64, HP=32, seed=804)
    make_stated_case("stated_d256_gt_pi", "pivotcvae_gt_pi", S=6, D=256, Z=16, N=1501, NU=40, B=48, H=64, HP=32, seed=805)
    # config 4's catalog, slate and width AS STATED (N = 10^6, S = 10, D = 128) with 16 slates and narrow hidden layers: the size at
    # which the headline kernels run their real plans (catalog ranges, ring trips, screened argmax) meets the reference itself
    make_stated_case("stated_config4_catalog_gt_pi", "pivotcvae_gt_pi", S=10, D=128, Z=16, N=1_000_000, NU=50, B=16, H=64, HP=32,
                     seed=806, tables_from_seed=True, n_neg_part=1000)
    # ... and config 3's (N = 10^5, S = 10, D = 64: the width and size of the bf16 pipelined kernel), 64 slates
    make_stated_case("stated_config3_catalog_gt_pi", "pivotcvae_gt_pi", S=10, D=64, Z=16, N=100_000, NU=50, B=64, H=64, HP=32,
                     seed=807, tables_from_seed=True)
    # ... config 5's slate and width (S = 20, D = 256) over a catalog long enough for the D = 256 kernels' steady-state trips
    make_stated_case("stated_config5_width_gt_pi", "pivotcvae_gt_pi", S=20, D=256, Z=16, N=200_000, NU=50, B=8, H=32, HP=32,
                     seed=809, tables_from_seed=True)
    # ... and the argmax-pivot rule in training (pt: the PSM stack trains, the pivot is a greedy id over the 10^6 items inside the loss)
    make_stated_case("stated_config4_catalog_pt_pi", "pivotcvae_pt_pi", S=10, D=128, Z=16, N=1_000_000, NU=50, B=16, H=64, HP=32,
                     seed=808, tables_from_seed=True)


def make_candidate_mode(name, like, n_candidate, np_seed):
    """Round 5: the reference's DEFAULT training mode (train_generative.py:270-274: candidate sets unless --mask_train; my_utils.py:169
    --nneg 1000) at a stated size.  The model, its state and the batch are those of the stated case `like` (rebuilt with the same
    calls and CHECKED against that fixture: this file only adds what is new).  The candidate sets are the ones the reference's own
    dataset class makes (data_loader.UserSlateResponseDataset.__getitem__ with init_sampling(n_candidate), :46-58) under
    np.random.seed(np_seed + k) for draw k: the fixture keeps the seeds, max_iid and checksums - the test redraws the uniform ids
    with numpy and applies the first-hit / overwrite rule (oracle) - then the loss terms and every .grad of
    get_gen_loss(batch with sample_candidates / sample_targets) (:52-57), and three Adam steps with a fresh draw per step."""
    import data_loader as ref_dl
    ref = np.load(os.path.join(OUT, like + ".npz"), allow_pickle=False)
    meta = json.loads(str(ref["meta"]))
    model, S, D, Z, N, NU, B, seed = (meta[k] for k in ("model", "S", "D", "Z", "N", "NU", "B", "seed"))
    beta, lr, st = meta["beta"], meta["lr"], meta["structs"]
    torch.manual_seed(seed)
    a = (2.0 / D) ** 0.5
    raw_doc = torch.nn.Embedding(N, D)
    raw_doc.weight.data.uniform_(-a, a)
    raw_user = torch.nn.Embedding(NU, D)
    raw_user.weight.data.uniform_(-a, a)
    m = build(model, st, raw_doc, raw_user, S, D, Z, False)
    g = torch.Generator().manual_seed(seed + 1000)
    s = torch.randint(0, N, (B, S), generator=g)
    u = torch.randint(0, NU, (B, 1), generator=g)
    r = (torch.rand(B, S, generator=g) < 0.5).float()
    r[0] = 0.0
    r[-1] = 1.0
    assert np.array_equal(s.numpy(), ref["s"]) and np.array_equal(r.numpy(), ref["r"]) and np.array_equal(u.numpy(), ref["u"])
    for k, v in m.state_dict().items():
        if "sd/" + k in ref.files:
            assert np.array_equal(v.numpy(), ref["sd/" + k]), k       # the same model as the stated case, bit for bit
    ds = quiet(ref_dl.UserSlateResponseDataset, s.numpy(), u.numpy(), r.numpy(), False)
    quiet(ds.init_sampling, n_candidate)

    def draw(k):   # what a DataLoader batch of the reference carries: per item a [S, Cn] draw + S target columns
        np.random.seed(np_seed + k)
        items = [ds[i] for i in range(B)]
        return (np.stack([np.asarray(it["sample_candidates"]) for it in items]).astype(np.int64),
                np.stack([np.asarray(it["sample_targets"]) for it in items]).astype(np.int64))

    def checksum(c, t):
        w = (np.arange(c.size, dtype=np.int64) % 1000003 + 1).reshape(c.shape)
        return [int(c.sum()), int((c * w).sum() % (1 << 61)), int(t.sum()), int((t > 0).sum())]

    out = {"np_seed": np.array(np_seed), "max_iid": np.array(int(ds.max_iid)), "n_candidate": np.array(n_candidate)}
    CEL = torch.nn.CrossEntropyLoss()
    m.candidateFlag = True
    cand, tgt = draw(0)
    out["cand/checksum"] = np.array(checksum(cand, tgt), dtype=np.int64)
    batch = {"slates": s.numpy(), "users": u.numpy(), "responses": r.numpy(), "sample_candidates": cand, "sample_targets": tgt}
    torch.manual_seed(seed + 12)
    m.zero_grad()
    with Recorder() as rec:
        loss, recLoss, KLD = ref_tg.get_gen_loss(batch, m, CEL, beta)
        loss.backward()
    out["cand/eps"] = rec.eps[0].numpy()
    out["cand/loss"] = np.array([loss.item(), recLoss.item(), KLD.item()], dtype=np.float64)
    none_grads = []
    for k, prm in m.named_parameters():
        if prm.grad is None:
            none_grads.append(k)
        else:
            out["cand/grad/" + k] = prm.grad.detach().numpy().copy()
    # three optimisation steps, a fresh draw per step (every epoch's __getitem__ draws anew)
    m2 = build(model, st, raw_doc, raw_user, S, D, Z, False)
    m2.load_state_dict(m.state_dict())
    m2.candidateFlag = True
    opt = torch.optim.Adam(m2.parameters(), lr=lr)
    torch.manual_seed(seed + 14)
    frozen = ("docEmbed.weight", "userEmbed.weight")
    for step in range(3):
        cand, tgt = draw(1 + step)
        out[f"adam/checksum{step}"] = np.array(checksum(cand, tgt), dtype=np.int64)
        b2 = dict(batch, sample_candidates=cand, sample_targets=tgt)
        opt.zero_grad()
        with Recorder() as rec:
            loss, recLoss, KLD = ref_tg.get_gen_loss(b2, m2, CEL, beta)
        loss.backward()
        opt.step()
        out[f"adam/eps{step}"] = rec.eps[0].numpy()
        out[f"adam/loss{step}"] = np.array([loss.item(), recLoss.item(), KLD.item()], dtype=np.float64)
        if step in (0, 2):
            for k, v in m2.state_dict().items():
                if k not in frozen and k not in none_grads:
                    out[f"adam/step{step + 1}/" + k] = v.detach().numpy().copy()
    out["meta"] = np.array(json.dumps(dict(name=name, like=like, n_candidate=n_candidate, np_seed=np_seed, none_grads=none_grads,
                                            max_iid=int(ds.max_iid), torch=torch.__version__, numpy=np.__version__)))
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {len(out)} arrays, {os.path.getsize(path) / 1e6:.1f} MB, loss {out['cand/loss']}, "
          f"{out['cand/checksum'][3]} of {B * S} slots hit their own item")


def make_candidate_modes():
    make_candidate_mode("candidate_mode_config2", "stated_config2_gt_pi", n_candidate=1000, np_seed=9021)
    make_candidate_mode("candidate_mode_config4_catalog", "stated_config4_catalog_gt_pi", n_candidate=1000, np_seed=9041)


def make_response_model(name, N, NU, D, S, B, H, seed):
    """G7: UserResponseModel_MLP (env/response_model.py:46-87)."""
    torch.manual_seed(seed)
    rm = quiet(ref_env.UserResponseModel_MLP, N - 1, NU - 1, D, S, [(S + 1) * D, H, H, S], "cpu", False)
    g = torch.Generator().manual_seed(seed + 1)
    s = torch.randint(0, N, (B, S), generator=g)
    u = torch.randint(0, NU, (B,), generator=g)
    with torch.no_grad():
        logits = rm(s, u)
    out = {"sd/" + k: v.numpy().copy() for k, v in rm.state_dict().items()}
    out.update(s=s.numpy(), u=u.numpy(), logits=logits.numpy())
    out["meta"] = np.array(json.dumps(dict(name=name, N=N, NU=NU, D=D, S=S, B=B, H=H, seed=seed)))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: {len(out)} arrays")


def make_response_training(name, N, NU, D, S, B, H, seed, lr, decay, steps=3):
    """G8: three optimisation steps of pretrain_env.train_response_model's loop body (pretrain_env.py:76-92) on one fixed
    batch with users of shape [B, 1] (what data_loader.UserSlateResponseDataset yields): BCELoss of the sigmoid of the
    click logits, torch.optim.Adam(lr, weight_decay) over every parameter (tables included)."""
    torch.manual_seed(seed)
    rm = quiet(ref_env.UserResponseModel_MLP, N - 1, NU - 1, D, S, [(S + 1) * D, H, H, S], "cpu", False)
    g = torch.Generator().manual_seed(seed + 1)
    s = torch.randint(0, N, (B, S), generator=g)
    s[1] = s[0]                                   # repeated item rows: their gradients must add up
    u = torch.randint(0, NU, (B, 1), generator=g)
    r = (torch.rand(B, S, generator=g) < 0.4).float()
    out = {"sd/" + k: v.numpy().copy() for k, v in rm.state_dict().items()}
    out.update(s=s.numpy(), u=u.numpy(), r=r.numpy())
    bce, sig = torch.nn.BCELoss(), torch.nn.Sigmoid()
    opt = torch.optim.Adam(rm.parameters(), lr=lr, weight_decay=decay)
    losses = []
    for t in range(steps):
        opt.zero_grad()
        pred = rm.forward(s, u)
        loss = bce(sig(pred.reshape(-1)), r.reshape(-1))
        loss.backward()
        if t == 0:
            out["logits0"] = pred.detach().numpy().copy()
            for k, p in rm.named_parameters():
                out["grad/" + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
        opt.step()
        losses.append(loss.item())
        if t in (0, steps - 1):
            for k, v in rm.state_dict().items():
                out[f"after{t + 1}/" + k] = v.numpy().copy()
    out["losses"] = np.array(losses, dtype=np.float64)
    out["meta"] = np.array(json.dumps(dict(name=name, N=N, NU=NU, D=D, S=S, B=B, H=H, seed=seed, lr=lr, decay=decay, steps=steps,
                                           torch=torch.__version__)))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: {len(out)} arrays, losses {losses}")


def make_urm(name, N, NU, D, S, B, seed):
    """G10: the simulators as evaluators - URM / URM_P / URM_P_MR.forward (env/response_model.py:97-154, 264-323) on the same
    tables, biases, slates and users (non-zero biases: the constructors' zeros would hide them)."""
    torch.manual_seed(seed)
    urm = quiet(ref_env.URM, N - 1, NU - 1, S, D, "cpu", False)
    urm.itemBias.weight.data.uniform_(-0.5, 0.5)
    urm.userBias.weight.data.uniform_(-0.5, 0.5)
    sd = {k: v.clone() for k, v in urm.state_dict().items()}
    urm_p = quiet(ref_env.URM_P, N - 1, NU - 1, S, D, "cpu", False, 0.2, -0.1)
    urm_p.load_state_dict(sd)
    urm_mr = quiet(ref_env.URM_P_MR, N - 1, NU - 1, S, D, "cpu", False, 0.2, -0.1, 0.35)
    urm_mr.load_state_dict(sd)
    urm_mr.posBias, urm_mr.posDependentBias = urm_p.posBias.clone(), urm_p.posDependentBias.clone()
    g = torch.Generator().manual_seed(seed + 1)
    s = torch.randint(0, N, (B, S), generator=g)
    s[2] = s[2, 0]                                    # a slate of one repeated item
    u = torch.randint(0, NU, (B,), generator=g)
    with torch.no_grad():
        out = {"sd/" + k: v.numpy().copy() for k, v in sd.items()}
        out.update(s=s.numpy(), u=u.numpy(), p_urm=urm(s, u).numpy(), p_urm_p=urm_p(s, u).numpy(), p_urm_p_mr=urm_mr(s, u).numpy(),
                   posBias=urm_p.posBias.numpy().copy(), posDependentBias=urm_p.posDependentBias.numpy().copy())
    out["meta"] = np.array(json.dumps(dict(name=name, N=N, NU=NU, D=D, S=S, B=B, seed=seed, p_bias_max=0.2, p_bias_min=-0.1,
                                           mr_factor=0.35, torch=torch.__version__)))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: {len(out)} arrays")


def make_candidates(name, N, S, Cn, L, seed):
    """G11: data_loader.UserSlateResponseDataset.__getitem__ with sampling (data_loader.py:46-58): the uniform draw it makes
    (re-derived by seeding numpy the same way right before) and what the first-hit / overwrite rule turns it into.  A small N
    makes hits (also repeated ones) common."""
    import data_loader as ref_dl
    g = np.random.RandomState(seed)
    slates = g.randint(0, N, size=(L, S))
    users = g.randint(0, 5, size=(L,))
    resp = (g.rand(L, S) < 0.5).astype(np.float64)
    ds = quiet(ref_dl.UserSlateResponseDataset, slates, users, resp, False)
    quiet(ds.init_sampling, Cn)
    raws, cands, tgts = [], [], []
    for i in range(L):
        np.random.seed(seed + 100 + i)
        raws.append(np.random.randint(ds.max_iid + 1, size=(S, Cn)))
        np.random.seed(seed + 100 + i)
        item = ds[i]
        cands.append(np.asarray(item["sample_candidates"]).copy())
        tgts.append(np.asarray(item["sample_targets"]).copy())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), slates=slates, raw=np.stack(raws), candidates=np.stack(cands),
                        targets=np.stack(tgts), max_iid=np.array(ds.max_iid),
                        meta=np.array(json.dumps(dict(name=name, N=N, S=S, Cn=Cn, L=L, seed=seed))))
    print(f"{name}: hits in {int((np.stack(tgts) > 0).sum())} of {L * S} slots")


def make_analysis(name="response_analysis"):
    """G9: analysis.get_coverage / get_ILS (analysis.py:5-30) on slates with repeated items."""
    import analysis as ref_analysis
    torch.manual_seed(7)
    N, D, B, S = 300, 24, 40, 5
    emb = torch.nn.Embedding(N, D)
    g = torch.Generator().manual_seed(8)
    slates = torch.randint(0, 60, (B, S), generator=g)      # few distinct ids: coverage < 1, repeated items inside slates
    slates[3] = slates[3, 0]                                  # a slate of one repeated item: ILS = 1
    cov = ref_analysis.get_coverage(slates, N)
    ils = ref_analysis.get_ILS(slates, emb)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), E=emb.weight.detach().numpy(), slates=slates.numpy(),
                        coverage=np.array(cov), ils=ils.detach().numpy(),
                        meta=np.array(json.dumps(dict(name=name, N=N, D=D, B=B, S=S))))
    print(f"{name}: coverage {cov}")


def make_pickles():
    """round 4, SURVEY 8(f)4: two WHOLE-MODULE pickles written the way the reference writes its checkpoints
    (train_generative.py:198-213: ``torch.save(model, open(path, 'wb'))``; the click model is loaded the same way, :259): the gt_pi
    model of case ``pivotcvae_gt_pi_user`` in its initial state and the click model of case ``response_mlp`` - rebuilt from those
    cases' seeds and checked against their committed ``sd/`` arrays before anything is written.  A pickle is data: tensors, the
    hyper-parameter attributes and the CLASS PATHS models.pivotcvae.UserPivotCVAE / env.response_model.UserResponseModel_MLP."""
    keys = list(ref_pivot.PIVOTCVAE_MODELS)
    seed = 100 + keys.index("pivotcvae_gt_pi")
    S, D, Z, N, NU, H, HP = 5, 16, 4, 203, 11, 24, 12
    torch.manual_seed(seed)
    a = (2.0 / D) ** 0.5
    raw_doc = torch.nn.Embedding(N, D)
    raw_doc.weight.data.uniform_(-a, a)
    raw_user = torch.nn.Embedding(NU, D)
    raw_user.weight.data.uniform_(-a, a)
    m = build("pivotcvae_gt_pi", structs("pivotcvae_gt_pi", S, D, Z, H, HP, False), raw_doc, raw_user, S, D, Z, False)
    gold = np.load(os.path.join(OUT, "pivotcvae_gt_pi_user.npz"))
    for k, v in m.state_dict().items():
        assert np.array_equal(v.numpy(), gold["sd/" + k]), k
    torch.save(m, open(os.path.join(OUT, "ref_pickle_pivotcvae_gt_pi_user.pt"), "wb"))
    torch.manual_seed(401)
    rm = quiet(ref_env.UserResponseModel_MLP, 203 - 1, 11 - 1, 16, 5, [(5 + 1) * 16, 24, 24, 5], "cpu", False)
    gold = np.load(os.path.join(OUT, "response_mlp.npz"))
    for k, v in rm.state_dict().items():
        assert np.array_equal(v.numpy(), gold["sd/" + k]), k
    torch.save(rm, open(os.path.join(OUT, "ref_pickle_response_mlp.pt"), "wb"))
    print("ref_pickle_pivotcvae_gt_pi_user.pt, ref_pickle_response_mlp.pt written (states equal the committed goldens')")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "pickles":   # round 4 (every earlier case stays byte-identical)
        make_pickles()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "response_analysis":
        make_analysis()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "candidate_modes":   # round 5 (every earlier case stays byte-identical)
        make_candidate_modes()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "stated":   # round 3 (every earlier case stays byte-identical)
        make_stated()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "round2":   # G10 / G11 only (every earlier case stays byte-identical)
        make_urm("response_urm", N=203, NU=11, D=16, S=5, B=9, seed=601)
        make_candidates("candidate_sets", N=40, S=5, Cn=12, L=16, seed=701)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "response_training":   # mint only the new case (the others stay byte-identical)
        make_response_training("response_training", N=203, NU=11, D=16, S=5, B=12, H=24, seed=501, lr=1e-2, decay=1e-3)
        return
    keys = list(ref_pivot.PIVOTCVAE_MODELS)
    # all 8 pivot variants, with user, S=5 D=16 (N=203: seven 32-row catalog tiles, ragged tail)
    for i, k in enumerate(keys):
        make_case(k + "_user", k, S=5, D=16, Z=4, N=203, NU=11, B=7, H=24, HP=12, no_user=False, seed=100 + i)
    # no-user variants
    make_case("pivotcvae_gt_pi_nouser", "pivotcvae_gt_pi", S=5, D=16, Z=4, N=203, NU=11, B=7, H=24, HP=12,
              no_user=True, seed=201)
    # a second shape: S=10 D=32 Z=8, N just over a tile multiple
    make_case("pivotcvae_gt_pi_s10", "pivotcvae_gt_pi", S=10, D=32, Z=8, N=321, NU=9, B=5, H=32, HP=16,
              no_user=False, seed=202)
    # List-CVAE baseline (config 1 plumbing)
    make_case("listcvae_user", "listcvae", S=5, D=16, Z=4, N=203, NU=11, B=7, H=24, HP=12, no_user=False, seed=301)
    make_case("listcvae_nouser", "listcvae", S=5, D=16, Z=4, N=203, NU=11, B=7, H=24, HP=12, no_user=True, seed=302)
    make_response_model("response_mlp", N=203, NU=11, D=16, S=5, B=9, H=24, seed=401)
    make_response_training("response_training", N=203, NU=11, D=16, S=5, B=12, H=24, seed=501, lr=1e-2, decay=1e-3)
    make_analysis()
    make_urm("response_urm", N=203, NU=11, D=16, S=5, B=9, seed=601)
    make_candidates("candidate_sets", N=40, S=5, Cn=12, L=16, seed=701)
    make_stated()
    make_pickles()


if __name__ == "__main__":
    main()
