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
