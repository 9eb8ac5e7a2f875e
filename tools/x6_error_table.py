#!/usr/bin/env python3
"""Error of the catalog CE kernels against fp64, per arithmetic (exact f32 MFMA, bf16x6, bf16x3) and per family of rows (GPU).

    python tools/x6_error_table.py [out.json]

Families: the model's scale (|x| ~ 8), cancelling rows (alternating +-a against near-constant table rows), large norms (|x| ~ 23,
logits to +-10), one dominant logit of 30, N = 1 (the lse IS the logit: isolates the logits chain + exp2 / log2).
Reported per family: max and rms of (lse - lse64), its MEAN (a bias would show there), max |dx - dx64| / max |dx64|."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pivotcvae_amd import ops                # noqa: E402
from pivotcvae_amd._hip import PREC_NAMES    # noqa: E402

DEV, D = "cuda:0", 128


def normalize_rows(w):
    return w / w.pow(2).sum(dim=1, keepdim=True).sqrt().clamp_min(1e-12)


def truth64(rx, E, tgt):
    lg = rx.double() @ E.double().t()
    lse = torch.logsumexp(lg, dim=1)
    return lse, torch.softmax(lg, dim=1) @ E.double() - E.double()[tgt]


def families(seed=31):
    g = torch.Generator().manual_seed(seed)
    out = {}
    N, R = 20000, 256
    E = normalize_rows(torch.rand(N, D, generator=g) * 2 - 1)
    out["model_scale_|x|=8"] = ((torch.rand(R, D, generator=g) * 2 - 1) * 1.2, E)
    out["large_norm_|x|=23"] = ((torch.rand(R, D, generator=g) * 2 - 1) * 3.5, E)
    out["dominant_logit_30"] = (torch.stack([E[(17 * i) % N] * 30.0 for i in range(R)]), E)
    Ec = normalize_rows(torch.ones(N, D) * 0.7 + (torch.rand(N, D, generator=g) * 2 - 1) * 0.3)
    alt = torch.tensor([1.0, -1.0]).repeat(D // 2)
    out["cancelling_rows_const_table"] = (torch.stack([alt * (3.0 + 0.007 * i) + (torch.rand(D, generator=g) * 2 - 1) * 0.05
                                                        for i in range(R)]), Ec)
    out["positive_table_|x|=8"] = ((torch.rand(R, D, generator=g) * 2 - 1) * 1.2, Ec)
    out["N=1_lse_is_the_logit"] = ((torch.rand(2048, D, generator=g) * 2 - 1) * 3.5, E[:1])
    return out


def main():
    res = {}
    for name, (rx, E) in families().items():
        R, N = rx.shape[0], E.shape[0]
        tgt = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(5))
        wl, wd = truth64(rx, E, tgt)
        table = ops.CatalogTable(E.to(DEV))
        row = {"R": R, "N": N, "max_abs_lse64": float(wl.abs().max()), "dx_scale": float(wd.abs().max())}
        for prec in ("f32", "bf16x6", "bf16x3"):
            _, lse, dx = ops.catalog_ce_raw(rx.to(DEV), table, tgt.to(DEV), prec=PREC_NAMES[prec])
            el = lse.double().cpu() - wl
            ed = (dx.double().cpu() - wd).abs().max() / wd.abs().max()
            row[prec] = {"lse_max": float(el.abs().max()), "lse_rms": float(el.pow(2).mean().sqrt()), "lse_mean": float(el.mean()),
                         "dx_max_over_scale": float(ed)}
        res[name] = row
        print(name, json.dumps(row), flush=True)
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r04_x6_error_table.json")
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    json.dump(res, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
