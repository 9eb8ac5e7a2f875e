"""Helpers for the -m gpu parity tests (everything here runs on the GPU box; no /root/reference access)."""
import numpy as np
import torch

import pivotcvae_amd as pa
from pivotcvae_amd.models.listcvae import UserListCVAEWithPrior

DEV = "cuda:0"


def build_from_golden(g, device=DEV):
    """Construct the product model from a golden case: raw tables in, state_dict loaded on top."""
    m = g.meta
    st = m["structs"]
    doc = torch.nn.Embedding.from_pretrained(g.t("raw_doc"), freeze=True)
    usr = torch.nn.Embedding.from_pretrained(g.t("raw_user"), freeze=True)
    C = m["S"] + 1
    if m["model"] == "listcvae":
        model = UserListCVAEWithPrior(doc, None if m["no_user"] else usr, m["S"], m["D"], m["Z"], C, st["enc"],
                                      st["dec"], st["prior"], m["no_user"], device)
    else:
        model = pa.PIVOTCVAE_MODELS[m["model"]](doc, None if m["no_user"] else usr, m["S"], m["D"], m["Z"], C,
                                               st["enc"], st["psm"], st["scm"], st["prior"], m["no_user"], device)
    # the constructor normalised the raw tables itself (G1); everything else comes from the golden state
    torch.testing.assert_close(model.docEmbed.weight.cpu(), g.t("sd/docEmbed.weight"), rtol=2e-6, atol=2e-7)
    model.load_state_dict({k: v.to(device) for k, v in g.sd.items()})
    return model


def dev(t):
    return t.to(DEV)


def close(a, b, rtol, atol):
    torch.testing.assert_close(a.detach().cpu(), b.detach().cpu() if torch.is_tensor(b) else torch.as_tensor(b),
                               rtol=rtol, atol=atol)
