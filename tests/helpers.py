"""Shared test helpers: golden loading (data only; never reads /root/reference)."""
import glob
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def table_fingerprint(a):
    """13 numbers of a [rows, D] float32 table: fp64 sum and absolute sum of every 97th row, nine sampled entries (the same function
    as in tests/golden/make_goldens.py)"""
    sub = np.asarray(a[::97], dtype=np.float64)
    mid = a.shape[0] // 2
    return [float(sub.sum()), float(np.abs(sub).sum())] + [float(v) for v in a[0, :3]] + [float(v) for v in a[-1, -3:]] + \
           [float(v) for v in a[mid, 1:4]]


_TABLES = {}   # golden name -> the four table arrays redrawn from the golden's seed (drawn once per process)


def tables_from_seed(meta, fingerprints):
    """Redraw the raw item / user tables of a golden that keeps only their SEED (a 10^6-row catalog is 0.5 GB), with the same torch
    calls as tests/golden/make_goldens.py::make_stated_case, normalise them as the reference's constructor does
    (models/cvae.py:14-36: F.normalize, eps 1e-12) and check every fingerprint the golden recorded - a torch whose CPU generator
    drew something else fails HERE, loudly, not as a parity mismatch later."""
    if meta["name"] in _TABLES:
        return _TABLES[meta["name"]]
    N, NU, D = meta["N"], meta["NU"], meta["D"]
    torch.manual_seed(meta["seed"])
    a = (2.0 / D) ** 0.5
    raw_doc = torch.nn.Embedding(N, D)
    raw_doc.weight.data.uniform_(-a, a)
    raw_user = torch.nn.Embedding(NU, D)
    raw_user.weight.data.uniform_(-a, a)
    out = {"raw_doc": raw_doc.weight.detach().numpy(), "raw_user": raw_user.weight.detach().numpy()}
    out["sd/docEmbed.weight"] = torch.nn.functional.normalize(raw_doc.weight.detach(), p=2, dim=1).numpy()
    out["sd/userEmbed.weight"] = torch.nn.functional.normalize(raw_user.weight.detach(), p=2, dim=1).numpy()
    for key, want in fingerprints.items():
        if not np.allclose(table_fingerprint(out[key]), want, rtol=1e-9, atol=0.0):
            raise RuntimeError(f"{meta['name']}: tables redrawn from seed {meta['seed']} do not match the golden's fingerprint of {key} "
                               f"(torch {torch.__version__} here, {meta.get('torch')} when it was minted)")
    _TABLES[meta["name"]] = out
    return out


class Golden:
    def __init__(self, path):
        z = np.load(path, allow_pickle=False)
        self.meta = json.loads(str(z["meta"]))
        self.a = {k: z[k] for k in z.files if k not in ("meta", "tables/fingerprints")}
        if self.meta.get("tables_from_seed"):
            self.a.update(tables_from_seed(self.meta, json.loads(str(z["tables/fingerprints"]))))
        if "part/neg_rows" in self.a:   # a recorded Bernoulli draw kept sparsely: the dense [R, N] mask the reference drew
            R, N = (int(v) for v in self.a.pop("part/neg_shape"))
            dense = np.zeros((R, N), dtype=np.uint8)
            dense[self.a.pop("part/neg_rows"), self.a.pop("part/neg_cols")] = 1
            self.a["part/neg_sample"] = dense

    def t(self, key):
        return torch.from_numpy(np.ascontiguousarray(self.a[key]))

    def has(self, key):
        return key in self.a

    def sub(self, prefix):
        """{name: tensor} for every key under ``prefix/``."""
        p = prefix + "/"
        return {k[len(p):]: self.t(k) for k in self.a if k.startswith(p)}

    @property
    def sd(self):
        return self.sub("sd")

    def cfg(self):
        from oracle.pivotcvae_oracle import Config
        m = self.meta
        return Config(model=m["model"], S=m["S"], D=m["D"], Z=m["Z"], no_user=m["no_user"], structs=m["structs"])


def model_cases():
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))
    return [n for n in names if not n.startswith(("response_", "candidate_", "stated_"))]


def stated_cases():
    """goldens minted from the reference at a BASELINE config's STATED size (no dense logits inside): tests/test_*stated*"""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "stated_*.npz")))


def load(name):
    return Golden(os.path.join(GOLDEN, name + ".npz"))


def candidate_mode_cases():
    """goldens of the reference's DEFAULT training mode (candidate sets) at a stated size (round 5): tests/test_*stated*"""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "candidate_mode_*.npz")))


class CandidateMode:
    """A candidate-mode golden: model / state / batch of the stated case it names (`like`), plus the loss terms, gradients and Adam
    steps the reference produced on candidate sets its own dataset class drew.  The sets themselves are not stored: ``draw(k)``
    redraws draw k's uniform ids with numpy exactly as data_loader.py:46 does (np.random.seed(np_seed + k), one
    randint(max_iid + 1, size=(S, Cn)) per slate), applies the first-hit / overwrite rule (the oracle's candidate_targets) and checks
    the checksums the golden recorded - a numpy whose legacy stream drew something else fails HERE."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.meta = json.loads(str(z["meta"]))
        self.a = {k: z[k] for k in z.files if k != "meta"}
        self.base = load(self.meta["like"])
        self._draws = {}

    def t(self, key):
        return torch.from_numpy(np.ascontiguousarray(self.a[key]))

    def sub(self, prefix):
        p = prefix + "/"
        return {k[len(p):]: self.t(k) for k in self.a if k.startswith(p)}

    def draw(self, k):
        """-> (sample_candidates [B, S, Cn] int64, sample_targets [B, S] int64) of draw k (0: the loss / gradient case; 1..3: the
        Adam steps)"""
        if k in self._draws:
            return self._draws[k]
        from oracle import pivotcvae_oracle as orc
        s = self.base.t("s")
        B, S = s.shape
        Cn, hi = int(self.a["n_candidate"]), int(self.a["max_iid"]) + 1
        np.random.seed(int(self.a["np_seed"]) + k)
        raw = np.stack([np.random.randint(hi, size=(S, Cn)) for _ in range(B)]).astype(np.int64)
        cand, tgt = orc.candidate_targets(s, torch.from_numpy(raw))
        c, t = cand.numpy(), tgt.numpy()
        w = (np.arange(c.size, dtype=np.int64) % 1000003 + 1).reshape(c.shape)
        got = [int(c.sum()), int((c * w).sum() % (1 << 61)), int(t.sum()), int((t > 0).sum())]
        want = [int(v) for v in self.a["cand/checksum" if k == 0 else f"adam/checksum{k - 1}"]]
        if got != want:
            raise RuntimeError(f"{self.meta['name']}: candidate sets redrawn with numpy {np.__version__} (minted with "
                               f"{self.meta.get('numpy')}) do not match the golden's checksums: {got} != {want}")
        self._draws[k] = (cand, tgt)
        return cand, tgt


# ---------------------------------------------------------------------------------------------------------------------------------
# LeakyReLU kinks: the one place where two arithmetics can differ by MORE than their rounding.  A hidden unit whose pre-activation
# lies within the arithmetic's perturbation of zero for some slate takes the other slope there; its derivative jumps 1 <-> 0.01 for
# that ONE slate, and that slate's share of (a) the unit's own row of its layer's weight gradient and bias gradient and (b) every
# gradient BELOW that layer changes discretely.  The tests that allow such entries name their cause with this.
LEAKY_STACKS = {"pivot": ("enc", "scm", "prior"), "list": ("enc", "dec", "prior")}


def leaky_kinks(sd, meta, s, r, u, eps, rel):
    """fp64 forward of the trained stacks from a state_dict (reference models/pivotcvae.py:159-227, 229-240, ground-truth pivot):
    -> {"enc_1": {unit: [slates]}, ...}: units j of LeakyReLU layer `name` with |pre[b, j]| < rel * sum_k |x[b, k]| |W[j, k]| for some
    slate b (`rel`: the relative error of one product in the arithmetic under test; the sum of absolute products bounds what a
    change of arithmetic can move the pre-activation by)."""
    tables = {k: torch.as_tensor(sd[k]) for k in ("docEmbed.weight", "userEmbed.weight") if k in sd}   # (up to 0.5 GB: rows only)
    sd = {k: torch.as_tensor(v).double() for k, v in sd.items() if k not in tables}
    S, D, no_user = meta["S"], meta["D"], meta["no_user"]
    B = s.shape[0]
    E = tables["docEmbed.weight"]
    emb = E[s.reshape(-1)].double().reshape(B, S * D)
    cond = torch.zeros(B, S + 1, dtype=torch.float64)
    cond[torch.arange(B), r.sum(1).long()] = 1.0
    parts_u = [] if no_user else [tables["userEmbed.weight"][u.reshape(-1)].double()]
    kinks = {}

    def stack(prefix, x, last_linear):
        n = 0
        while f"{prefix}_{n + 1}.weight" in sd:
            n += 1
        for i in range(1, n + 1):
            W, b = sd[f"{prefix}_{i}.weight"], sd[f"{prefix}_{i}.bias"]
            pre = x @ W.t() + b
            if not (last_linear and i == n):
                bound = rel * (x.abs() @ W.abs().t() + b.abs())
                hit = (pre.abs() < bound).nonzero()
                if hit.numel():
                    d = kinks.setdefault(f"{prefix}_{i}", {})
                    for bb, j in hit.tolist():
                        d.setdefault(j, []).append(bb)
                x = torch.where(pre > 0, pre, 0.01 * pre)
            else:
                x = pre
        return x

    h = stack("enc", torch.cat([emb, cond] + parts_u, 1), False)
    mu = h @ sd["encmu.weight"].t() + sd["encmu.bias"]
    lv = h @ sd["enclogvar.weight"].t() + sd["enclogvar.bias"]
    z = torch.as_tensor(eps).double() * torch.exp(0.5 * lv) + mu
    stack("prior", torch.cat([cond] + parts_u, 1), False)
    if meta["model"] == "listcvae":
        stack("dec", torch.cat([z, cond] + parts_u, 1), True)
    else:
        stack("scm", torch.cat([z, cond, E[s[:, 0]].double()] + parts_u, 1), True)
    return kinks


def explain_by_kinks(name, off_rows, kinks, model):
    """`off_rows`: rows (output units) of parameter `name` (e.g. "enc_2.weight") that hold out-of-tolerance gradient entries.  Assert
    that LeakyReLU kinks explain them: with L* the highest kinked layer of the chain that feeds back into this parameter (its own
    stack from its layer up; for the encoder also the whole decoder stack, which z feeds), the parameter must sit AT L* - then the
    offending rows are kinked units of that layer - or below it.  -> a description for the log."""
    stem = name.rsplit(".", 1)[0]
    heads = {"encmu": ("enc", 99), "enclogvar": ("enc", 99), "priorMu": ("prior", 99), "priorLogvar": ("prior", 99)}
    stk, layer = heads[stem] if stem in heads else (stem.rsplit("_", 1)[0], int(stem.rsplit("_", 1)[1]))
    dec = "dec" if model == "listcvae" else "scm"
    chain = [(k.rsplit("_", 1)[0], int(k.rsplit("_", 1)[1])) for k in kinks]
    above = [(a, i) for a, i in chain if (a == stk and i >= layer) or (stk == "enc" and a == dec)]
    assert above, f"{name}: {len(off_rows)} rows out of tolerance and no LeakyReLU kink at or above it (kinks: {sorted(kinks)})"
    own_top = max([i for a, i in above if a == stk], default=None)
    if own_top == layer and not any(a == dec for a, i in above if stk == "enc"):
        units = set(kinks[f"{stk}_{layer}"])
        assert set(off_rows) <= units, f"{name}: rows {sorted(set(off_rows) - units)} out of tolerance are not kinked units {sorted(units)}"
        return f"{name}: rows {sorted(set(off_rows))} = kinked units of {stk}_{layer}"
    return f"{name}: {len(set(off_rows))} rows, below the kink(s) at {sorted(above)}"
