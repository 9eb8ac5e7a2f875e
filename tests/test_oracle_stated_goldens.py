"""Pin the CPU oracle against the goldens minted from the real reference at STATED sizes (round 3): config 1 (List-CVAE,
N = 1000, S = 5, D = 16, B = 64), config 2 (PivotCVAE gt_pi, N = 10 000, S = 5, D = 32, B = 1024), and one case per width with MFMA
bf16 / bf16x3 catalog kernels (D = 128 / 64 / 256) -
loss terms, every gradient, three Adam steps, greedy ids.  The fixtures hold no dense [R, N] logits."""
import numpy as np
import pytest
import torch

from oracle import pivotcvae_oracle as orc
from tests.helpers import CandidateMode, candidate_mode_cases, load, stated_cases

CASES = stated_cases()


def close(a, b, rtol, atol):
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol)


def test_the_stated_cases_exist():
    assert CASES == ["stated_config1_listcvae", "stated_config2_gt_pi", "stated_config3_catalog_gt_pi", "stated_config4_catalog_gt_pi",
                     "stated_config4_catalog_pt_pi", "stated_config5_width_gt_pi", "stated_d128_gt_pi", "stated_d256_gt_pi",
                     "stated_d64_gt_pi"]
    assert sorted(load(n).meta["D"] for n in CASES[6:]) == [64, 128, 256]   # every width with MFMA bf16 / bf16x3 catalog kernels
    m5 = load("stated_config5_width_gt_pi").meta   # configs[4]'s slate and width over 200 000 items (steady-state trips at D = 256)
    assert (m5["N"], m5["S"], m5["D"], m5["tables_from_seed"]) == (200_000, 20, 256, True)
    # configs[2]'s and configs[3]'s catalog, slate and width as stated (64 / 16 slates; tables redrawn from the seed: helpers.py)
    m3, m4 = load(CASES[2]).meta, load(CASES[3]).meta
    assert (m3["N"], m3["S"], m3["D"], m3["tables_from_seed"]) == (100_000, 10, 64, True)
    assert (m4["N"], m4["S"], m4["D"], m4["tables_from_seed"]) == (1_000_000, 10, 128, True)
    m1, m2 = load(CASES[0]).meta, load(CASES[1]).meta
    assert (m1["model"], m1["N"], m1["S"], m1["D"], m1["B"]) == ("listcvae", 1000, 5, 16, 64)          # BASELINE.json configs[0]
    assert (m2["model"], m2["N"], m2["S"], m2["D"], m2["B"]) == ("pivotcvae_gt_pi", 10000, 5, 32, 1024)  # configs[1]
    assert m2["structs"]["enc"] == [198, 256, 256] and m2["structs"]["scm"] == [86, 256, 256, 128]


@pytest.mark.parametrize("name", CASES)
def test_forward_pieces_loss_and_gradients(name):
    g = load(name)
    cfg = g.cfg()
    close(orc.normalize_rows(g.t("raw_doc")), g.t("sd/docEmbed.weight"), 2e-6, 2e-7)
    f = orc.forward(g.sd, cfg, g.t("s"), g.t("r"), g.t("u"), g.t("fwd/eps"))
    for k in ("rx", "z", "z_mu", "z_logvar"):
        close(f[k], g.t("fwd/" + k), 1e-5, 1e-6)
    (loss, rec, kld), grads = orc.loss_and_grads(g.sd, cfg, g.t("s"), g.t("r"), g.t("u"), g.t("full/eps"), g.meta["beta"])
    np.testing.assert_allclose([loss, rec, kld], g.a["full/loss"], rtol=2e-6)
    assert sorted(k for k, v in grads.items() if v is None) == sorted(
        k for k in g.meta["none_grads"] if not k.startswith(("docEmbed", "userEmbed")))
    for k, v in g.sub("grad").items():
        close(grads[k], v, 5e-5, 2e-7)
    if g.has("part/neg_sample"):
        (loss, rec, kld), grads = orc.loss_and_grads(g.sd, cfg, g.t("s"), g.t("r"), g.t("u"), g.t("part/eps"), g.meta["beta"],
                                                     neg_sample=g.t("part/neg_sample"))
        np.testing.assert_allclose([loss, rec, kld], g.a["part/loss"], rtol=2e-6)
        for k, v in g.sub("part/grad").items():
            close(grads[k], v, 5e-5, 2e-7)


@pytest.mark.parametrize("name", CASES)
def test_three_adam_steps_and_greedy_ids(name):
    g = load(name)
    cfg = g.cfg()
    sd, state = g.sd, {}
    # a dense [160, 10^6] softmax costs the CPU seconds per evaluation (and this container's page allocator much more on a bad day):
    # the 10^6-item cases check the FIRST step only here; the three-step trajectory is held by the other six
    for step in range(1 if g.meta["N"] >= 500_000 else 3):
        (loss, rec, kld), grads = orc.loss_and_grads(sd, cfg, g.t("s"), g.t("r"), g.t("u"), g.t(f"adam/eps{step}"), g.meta["beta"])
        np.testing.assert_allclose([loss, rec, kld], g.a[f"adam/loss{step}"], rtol=3e-6)
        sd = orc.adam_step(sd, grads, state, g.meta["lr"])
        if step in (0, 2):
            for k, v in g.sub(f"adam/step{step + 1}").items():
                # Adam moves a weight by ~lr g / (|g| + 1e-8): where |g| is of the order of its own rounding error the normalised
                # step follows the noise (one of 45 760 elements of a D = 64 layer, 6.6e-7 off).  Counted, and bounded by the move.
                v = torch.as_tensor(v)
                diff = (sd[k] - v).abs()
                off = diff > 3e-7 + 2e-5 * v.abs()
                assert float(off.float().mean()) <= 1e-4 and float(diff.max()) <= 2.001 * g.meta["lr"] * (step + 1), \
                    (name, k, int(off.sum()), float(diff.max()))
    for k in g.meta["none_grads"]:
        assert torch.equal(sd[k], g.sd[k])
    o = orc.recommend(g.sd, cfg, g.t("rec/r"), g.t("u"), g.t("rec/eps"))
    close(o["rx"], g.t("rec/rx"), 1e-5, 1e-6)
    safe = torch.from_numpy(g.a["rec/item_margin"] > 1e-5)
    assert float(safe.float().mean()) > 0.99
    assert torch.equal(o["items"][safe], g.t("rec/items")[safe])
    if cfg.model != "listcvae":
        assert torch.equal(o["pivot"], g.t("rec/pivot"))


@pytest.mark.parametrize("name", candidate_mode_cases())
def test_candidate_mode_at_stated_sizes(name):
    """the reference's DEFAULT training mode (train_generative.py:52-57, 270-274; candidate sets of data_loader.py:46-58, 1000 ids
    per slot) at config 2 as stated and over config 4's catalog: the oracle's candidate branch against the reference's loss terms
    and gradients on the very sets the reference's dataset class drew (redrawn from the recorded numpy seeds, checksums checked)"""
    cm = CandidateMode(name)
    g = cm.base
    cand, tgt = cm.draw(0)
    assert cand.shape[-1] == 1000 and int(cand.max()) <= int(cm.a["max_iid"]) < g.meta["N"]
    (loss, rec, kld), grads = orc.loss_and_grads(g.sd, g.cfg(), g.t("s"), g.t("r"), g.t("u"), cm.t("cand/eps"), g.meta["beta"],
                                                 candidates=cand, cand_targets=tgt)
    np.testing.assert_allclose([loss, rec, kld], cm.a["cand/loss"], rtol=2e-6)
    for k, v in cm.sub("cand/grad").items():
        close(grads[k], v, 5e-5, 2e-7)
    assert sorted(k for k, v in grads.items() if v is None) == sorted(
        k for k in cm.meta["none_grads"] if not k.startswith(("docEmbed", "userEmbed")))


def test_the_candidate_mode_cases_exist():
    assert candidate_mode_cases() == ["candidate_mode_config2", "candidate_mode_config4_catalog"]
    for n, like in zip(candidate_mode_cases(), ("stated_config2_gt_pi", "stated_config4_catalog_gt_pi")):
        assert CandidateMode(n).meta["like"] == like and CandidateMode(n).meta["n_candidate"] == 1000
