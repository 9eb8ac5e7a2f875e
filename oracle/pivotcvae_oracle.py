"""CPU oracle for the PivotCVAE slate-generation hot path.  TEST INFRASTRUCTURE ONLY.

This is an independent restatement (plain torch fp32 ops on the CPU, functional style, state
passed as a ``state_dict``-shaped mapping) of what the reference computes on the path named
by BASELINE.json:north_star.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package ``pivotcvae_amd`` never
does (it fails loudly when the HIP library is missing instead of falling back to this).

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function below against
golden vectors minted by running the real reference in the build container
(``tests/golden/make_goldens.py``), for all 8 pivot variants, List-CVAE, user/no-user.

Reference citations (relative to /root/reference):
  normalize_rows        models/cvae.py:31,39          (F.normalize p=2 dim=1, eps 1e-12)
  condition             models/cvae.py:85-92
  encode                models/pivotcvae.py:159-174   (listcvae.py:91-104)
  reparametrize         models/cvae.py:79-83
  prior                 models/pivotcvae.py:229-240
  decode / pick_pivot   models/pivotcvae.py:186-227, 321-455   (listcvae.py:106-119)
  forward               models/pivotcvae.py:242-276   (listcvae.py:134-168)
  recommend             models/pivotcvae.py:278-296 + models/cvae.py:97-101
  downsample            train_generative.py:36-42
  gen_loss              train_generative.py:44-65
  adam_step             train_generative.py:103,124-134 (torch.optim.Adam defaults, no decay)
  response_mlp          env/response_model.py:76-87
  urm_forward           env/response_model.py:129-154, 286-295, 315-323   (URM / URM_P / URM_P_MR as evaluators; golden G10)
  candidate_targets     data_loader.py:46-58          (first-hit / overwrite rule on a recorded draw; golden G11)
  candidate_ce          models/pivotcvae.py:265-271 + train_generative.py:56   (per-row nll and d rx of the candidate branch; the
                        same ops as forward(candidates=..) + gen_loss, which golden G6 pins, applied to a bare rx)
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

LEAKY_SLOPE = 0.01  # nn.LeakyReLU() default, models/cvae.py:43

# training-time / inference-time pivot rule per registry key (models/pivotcvae.py:458-461)
PIVOT_RULES = {
    "pivotcvae_gt_pi": ("gt", "pi"),
    "pivotcvae_pt_pi": ("pt", "pi"),
    "pivotcvae_spt_pi": ("spt", "pi"),
    "pivotcvae_sgt_pi": ("sgt", "pi"),
    "pivotcvae_gt_spi": ("gt", "spi"),
    "pivotcvae_pt_spi": ("pt", "spi"),
    "pivotcvae_spt_spi": ("spt", "spi"),
    "pivotcvae_sgt_spi": ("sgt", "spi"),
}


@dataclass
class Config:
    model: str  # "listcvae" or a key of PIVOT_RULES
    S: int
    D: int
    Z: int
    no_user: bool
    structs: Dict[str, List[int]] = field(default_factory=dict)

    @property
    def C(self):
        return self.S + 1


def normalize_rows(w: torch.Tensor) -> torch.Tensor:
    n = w.pow(2).sum(dim=1, keepdim=True).sqrt().clamp_min(1e-12)
    return w / n


def condition(r: torch.Tensor, S: int) -> torch.Tensor:
    """one-hot of the click count: cond[b, sum_s r[b, s]] = 1."""
    cnt = r.sum(dim=1).to(torch.long)
    c = torch.zeros(r.shape[0], S + 1, dtype=torch.float32)
    c[torch.arange(r.shape[0]), cnt] = 1.0
    return c


def _linear(sd, name, x):
    return x @ sd[name + ".weight"].t() + sd[name + ".bias"]


def _n_layers(sd, prefix):
    i = 0
    while f"{prefix}_{i + 1}.weight" in sd:
        i += 1
    return i


def _mlp(sd, prefix, x, last_linear: bool):
    """LeakyReLU after every layer, except the last one when ``last_linear``."""
    n = _n_layers(sd, prefix)
    for i in range(1, n + 1):
        x = _linear(sd, f"{prefix}_{i}", x)
        if not (last_linear and i == n):
            x = F.leaky_relu(x, LEAKY_SLOPE)
    return x


def encode(sd, cfg: Config, emb, cond, u_emb):
    x = torch.cat([emb, cond] if cfg.no_user else [emb, cond, u_emb], 1)
    h = _mlp(sd, "enc", x, last_linear=False)
    return _linear(sd, "encmu", h), _linear(sd, "enclogvar", h)


def prior(sd, cfg: Config, r, u):
    cond = condition(r, cfg.S)
    x = cond if cfg.no_user else torch.cat([cond, sd["userEmbed.weight"][u.reshape(-1)]], 1)
    h = _mlp(sd, "prior", x, last_linear=False)
    return _linear(sd, "priorMu", h), _linear(sd, "priorLogvar", h)


def reparametrize(mu, logvar, eps):
    return eps * torch.exp(0.5 * logvar) + mu


def catalog_argmax(E, x):
    """first-occurrence argmax_n <x_r, E_n> (torch CPU max returns the first maximal index)."""
    return (x @ E.t()).max(1)[1]


def pick_pivot(sd, cfg: Config, pivot_output, true_pivot, pivot_sample=None):
    """Returns (pivot index [B], how) - the embedding is E[index]."""
    E = sd["docEmbed.weight"]
    train_rule, infer_rule = PIVOT_RULES[cfg.model]
    rule = infer_rule if true_pivot is None else train_rule
    if rule == "gt":
        return true_pivot
    if rule in ("pi", "pt"):
        return (E @ pivot_output.t()).max(0)[1]
    # sampled rules: Categorical(sigmoid(scores)); the draw itself is an input (recorded)
    if rule in ("spi", "spt"):
        probs = torch.sigmoid(pivot_output @ E.t())
    else:  # "sgt"
        probs = torch.sigmoid(E[true_pivot] @ E.t())
    if pivot_sample is not None:
        return pivot_sample
    return torch.multinomial(probs / probs.sum(1, keepdim=True), 1).reshape(-1)


def decode(sd, cfg: Config, z, cond, u_emb, true_pivot=None, pivot_sample=None):
    """-> (rx, pivot index or None); rx is [B,S,D] for pivot models and the flat [B,S*D]
    decoder output for List-CVAE (models/listcvae.py:119 returns it un-reshaped)."""
    B = z.shape[0]
    if cfg.model == "listcvae":
        x = torch.cat([z, cond] if cfg.no_user else [z, cond, u_emb], 1)
        return _mlp(sd, "dec", x, last_linear=True), None
    x = torch.cat([z, cond] if cfg.no_user else [z, cond, u_emb], 1)
    pivot_output = _mlp(sd, "psm", x, last_linear=True)
    pidx = pick_pivot(sd, cfg, pivot_output, true_pivot, pivot_sample)
    pivot_emb = sd["docEmbed.weight"][pidx]
    x = torch.cat([z, cond, pivot_emb] if cfg.no_user else [z, cond, pivot_emb, u_emb], 1)
    rest = _mlp(sd, "scm", x, last_linear=True).reshape(B, cfg.S - 1, cfg.D)
    return torch.cat([pivot_emb.reshape(B, 1, cfg.D), rest], 1), pidx


def forward(sd, cfg: Config, s, r, u, eps, pivot_sample=None, candidates=None):
    """The reference forward(): returns dict with the 6-tuple fields + cond."""
    B = s.shape[0]
    E = sd["docEmbed.weight"]
    cond = condition(r, cfg.S)
    emb = E[s.reshape(-1)].reshape(B, -1)
    u_emb = None if cfg.no_user else sd["userEmbed.weight"][u.reshape(-1)].reshape(B, -1)
    z_mu, z_logvar = encode(sd, cfg, emb, cond, u_emb)
    z = reparametrize(z_mu, z_logvar, eps)
    true_pivot = None if cfg.model == "listcvae" else s[:, 0]
    rx, pidx = decode(sd, cfg, z, cond, u_emb, true_pivot=true_pivot, pivot_sample=pivot_sample)
    prox = rx.reshape(-1, cfg.D)
    if candidates is not None:
        Cn = candidates.shape[-1]
        ce = E[candidates].reshape(-1, Cn, cfg.D)
        p = torch.bmm(ce, prox.reshape(-1, cfg.D, 1)).reshape(-1, Cn)
    else:
        p = prox @ E.t()
    return dict(p=p, rx=rx, z=z, emb=emb, z_mu=z_mu, z_logvar=z_logvar, cond=cond, pivot=pidx)


def downsample(pred, slate, neg_sample):
    """pred * mask with mask = onehot(target) OR neg_sample; masked-out logits become 0.0."""
    mask = neg_sample.to(pred.dtype).clone()
    mask[torch.arange(pred.shape[0]), slate.reshape(-1)] = 1.0
    return pred * mask


def kld(mu, logvar, pmu, plogvar):
    return -0.5 * torch.sum(1 + logvar - plogvar - (logvar.exp() + (mu - pmu).pow(2)) / plogvar.exp())


def gen_loss(sd, cfg: Config, s, r, u, eps, beta, neg_sample=None, pivot_sample=None,
             candidates=None, cand_targets=None):
    """-> (loss, recLoss, KLD).  ``neg_sample`` None means n_neg = N (mask of ones)."""
    pmu, plv = prior(sd, cfg, r, u)
    f = forward(sd, cfg, s, r, u, eps, pivot_sample=pivot_sample, candidates=candidates)
    if candidates is not None:
        rec = F.cross_entropy(f["p"], cand_targets.reshape(-1))
    else:
        p = f["p"] if neg_sample is None else downsample(f["p"], s, neg_sample)
        rec = F.cross_entropy(p, s.reshape(-1))
    k = kld(f["z_mu"], f["z_logvar"], pmu, plv)
    return rec + beta * k, rec, k


def trainable(sd):
    """Parameters the reference optimiser can ever touch: everything but the frozen tables."""
    return [k for k in sd if not k.startswith(("docEmbed", "userEmbed"))]


def loss_and_grads(sd, cfg: Config, s, r, u, eps, beta, **kw):
    """-> ((loss, rec, kld) floats, {name: grad or None})."""
    leaf = {k: v.clone().requires_grad_(k in trainable(sd)) for k, v in sd.items()}
    loss, rec, k = gen_loss(leaf, cfg, s, r, u, eps, beta, **kw)
    loss.backward()
    grads = {n: leaf[n].grad for n in trainable(sd)}
    return (loss.item(), rec.item(), k.item()), grads


def adam_step(sd, grads, state, lr, b1=0.9, b2=0.999, eps=1e-8):
    """One torch.optim.Adam step (no weight decay, bias-corrected); params with grad None skipped."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    new = dict(sd)
    for n, g in grads.items():
        if g is None:
            continue
        m = state.setdefault("m/" + n, torch.zeros_like(g))
        v = state.setdefault("v/" + n, torch.zeros_like(g))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** t
        bc2 = 1 - b2 ** t
        denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
        new[n] = sd[n] - (lr / bc1) * (m / denom)
    return new


def recommend(sd, cfg: Config, r, u, eps, pivot_sample=None):
    """-> dict(items [R], z_mu, rx [B,S,D], pivot [B] or None)."""
    cond = condition(r, cfg.S)
    u_emb = None if cfg.no_user else sd["userEmbed.weight"][u.reshape(-1)]
    x = cond if cfg.no_user else torch.cat([cond, u_emb], 1)
    h = _mlp(sd, "prior", x, last_linear=False)
    z_mu, z_lv = _linear(sd, "priorMu", h), _linear(sd, "priorLogvar", h)
    z = reparametrize(z_mu, z_lv, eps)
    rx, pidx = decode(sd, cfg, z, cond, u_emb, true_pivot=None, pivot_sample=pivot_sample)
    items = catalog_argmax(sd["docEmbed.weight"], rx.reshape(-1, cfg.D))
    return dict(items=items, z_mu=z_mu, rx=rx, pivot=pidx)


def response_mlp(sd, slates, users, no_user=False):
    """UserResponseModel_MLP.forward (env/response_model.py:76-87): the WHOLE concatenated slate vector is normalised.
    ``F.normalize(userEmbed(users), p=2, dim=1)``: with users [B] the user rows are L2-normalised; with users [B, 1] (the
    training batches of pretrain_env.py) the lookup is [B, 1, D] and dim=1 is its singleton axis, i.e. every component
    becomes x / max(|x|, 1e-12) - restated as written."""
    B = slates.shape[0]
    d = F.normalize(sd["docEmbed.weight"][slates].reshape(B, -1), p=2, dim=1)
    x = d if no_user else torch.cat([d, F.normalize(sd["userEmbed.weight"][users], p=2, dim=1).reshape(B, -1)], 1)
    n = _n_layers(sd, "mlp")
    for i in range(1, n + 1):
        x = _linear(sd, f"mlp_{i}", x)
        if i < n:
            x = F.relu(x)
    return x


def urm_forward(sd, slates, users, S, pos_bias=None, pos_dep=None, mr_factor=None):
    """URM / URM_P / URM_P_MR.core_forward as written (env/response_model.py:129-150, 286-295, 315-323): the item rows are
    L2-normalised per item, the user row is the RAW one (its normalised lookup is overwritten at :141-142), positional term
    with posDependentBias [S, D] read as [D, S] (``.view``, not a transpose), relation term against sigmoid(mean item)."""
    B = slates.shape[0]
    D = sd["docEmbed.weight"].shape[1]
    d = F.normalize(sd["docEmbed.weight"][slates], p=2, dim=-1).view(B, S, -1)
    d_bias = sd["itemBias.weight"][slates].view(B, S)
    u = sd["userEmbed.weight"][users.view(B)]
    u_bias = sd["userBias.weight"][users.view(B)]
    out = torch.bmm(d, u.view(B, D, 1)).view(B, S) + d_bias
    out = (out.transpose(0, 1) + u_bias.view(-1)).transpose(0, 1)
    p = torch.sigmoid(out)
    if pos_dep is not None:
        p = p.reshape(-1, S) + torch.mm(u.view(-1, D), pos_dep.view(D, S)) + pos_bias.view(-1)
    if mr_factor is not None:
        att = torch.sigmoid(torch.mean(d.view(-1, S, D), dim=1))
        p = p + torch.bmm(d.view(-1, S, D), att.view(-1, D, 1)).view(-1, S) * mr_factor
    return p


def candidate_targets(features, raw):
    """data_loader.UserSlateResponseDataset.__getitem__ (:46-58) applied to a recorded uniform draw ``raw`` [.., S, Cn]: if the
    slot's true item is among its candidates the target is the first column holding it, else column 0 becomes the item.
    -> (candidates, targets)"""
    cand = raw.clone()
    flat_c, flat_f = cand.view(-1, cand.shape[-1]), features.reshape(-1)
    tgt = torch.zeros(flat_f.shape[0], dtype=torch.long)
    for i in range(flat_f.shape[0]):
        hit = (flat_c[i] == flat_f[i]).nonzero()
        if len(hit):
            tgt[i] = hit[0, 0]
        else:
            flat_c[i, 0] = flat_f[i]
    return cand, tgt.view(features.shape)


def candidate_ce(rx, E, cand, tgt, dtype=torch.float64):
    """The candidate branch's tail on a bare rx [R, D]: candidateEmb = E[cand] [R, Cn, D]; p = bmm(candidateEmb, rx)
    (models/pivotcvae.py:268-270); per-row CrossEntropyLoss(p, tgt) (train_generative.py:56, reduction left to the caller) and its
    gradient with respect to rx by autograd.  -> (nll [R], lse [R], d(sum nll)/d rx [R, D]) in ``dtype``."""
    x = rx.to(dtype).clone().requires_grad_(True)
    Cn = cand.shape[-1]
    emb = E.to(dtype)[cand.reshape(-1, Cn)]
    p = torch.bmm(emb, x.view(-1, x.shape[1], 1)).view(-1, Cn)
    nll = F.cross_entropy(p, tgt.reshape(-1), reduction="none")
    nll.sum().backward()
    return nll.detach(), torch.logsumexp(p.detach(), dim=1), x.grad


def response_loss_and_grads(sd, slates, users, targets, no_user=False):
    """pretrain_env.py:82-90: BCELoss(sigmoid(logits), targets) (mean) and the gradient of every parameter."""
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    pred = response_mlp(leaf, slates, users, no_user)
    loss = F.binary_cross_entropy(torch.sigmoid(pred.reshape(-1)), targets.reshape(-1).float())
    loss.backward()
    return loss.item(), {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}


def adam_l2_step(sd, grads, state, lr, weight_decay, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam(weight_decay=...) as pretrain_env.py:59 uses it: the L2 term joins the gradient."""
    return adam_step(sd, {k: g + weight_decay * sd[k] for k, g in grads.items()}, state, lr, b1, b2, eps)


def coverage(slates, N):
    """analysis.get_coverage (analysis.py:5-12): distinct generated items / N"""
    return len(torch.unique(slates)) * 1.0 / N


def ils(slates, E):
    """analysis.get_ILS (analysis.py:14-30): mean pairwise cosine similarity of the items of a slate (self pairs removed)"""
    S = slates.shape[1]
    emb = F.normalize(E[slates], p=2, dim=2)
    sims = torch.bmm(emb, emb.transpose(1, 2)).reshape(slates.shape[0], -1)
    return (sims.sum(dim=1) - S) / (S * (S - 1))


# ------------------------------------------------------------------ synthetic workload
def synthetic_tables(N, NU, D, seed=0):
    """E_raw, U_raw ~ U(-a, a), a = sqrt(2/D) (env/response_model.py:29-36)."""
    g = torch.Generator().manual_seed(seed)
    a = (2.0 / D) ** 0.5
    return (torch.rand(N, D, generator=g) * 2 - 1) * a, (torch.rand(NU, D, generator=g) * 2 - 1) * a
