"""Shared test helpers: golden loading (data only; never reads /root/reference)."""
import glob
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self, path):
        z = np.load(path, allow_pickle=False)
        self.meta = json.loads(str(z["meta"]))
        self.a = {k: z[k] for k in z.files if k != "meta"}

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
