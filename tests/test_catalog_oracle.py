"""Pin oracle/catalog_oracle.c (k-ordered fmaf chain) against the goldens and the torch oracle."""
import numpy as np
import pytest
import torch

from oracle import catalog_oracle as co
from oracle import pivotcvae_oracle as orc
from tests.helpers import load, model_cases


@pytest.mark.parametrize("name", model_cases())
def test_argmax_matches_reference_item_ids(name):
    """Greedy ids from the C chain == ids the reference produced (margins are >> fp32 rounding)."""
    g = load(name)
    D = g.meta["D"]
    rx = g.a["rec/rx"].reshape(-1, D)
    E = g.a["sd/docEmbed.weight"]
    assert g.a["rec/item_margin"].min() > 1e-5  # qualifies the bit-exact claim for these rows
    idx, best = co.argmax(rx, E)
    np.testing.assert_array_equal(idx, g.a["rec/items"])
    np.testing.assert_allclose(best, (rx @ E.T).max(1), rtol=1e-5, atol=1e-6)


def test_argmax_first_index_on_ties():
    E = np.zeros((70, 16), np.float32)
    E[[5, 40, 69], 0] = 1.0  # three identical best rows -> index 5 wins
    x = np.zeros((3, 16), np.float32)
    x[:, 0] = [1.0, 2.0, -1.0]
    idx, _ = co.argmax(x, E)
    np.testing.assert_array_equal(idx, [5, 5, 0])


@pytest.mark.parametrize("name", ["pivotcvae_gt_pi_user", "pivotcvae_gt_pi_s10", "listcvae_user"])
@pytest.mark.parametrize("masked", [False, True])
def test_ce_matches_torch_oracle(name, masked):
    g = load(name)
    D, N = g.meta["D"], g.meta["N"]
    rx = g.t("fwd/rx").reshape(-1, D).clone().requires_grad_(True)
    E = g.t("sd/docEmbed.weight")
    tgt = g.t("s").reshape(-1)
    neg = g.t("part/neg_sample") if masked else None
    p = rx @ E.t()
    if masked:
        p = orc.downsample(p, g.t("s"), neg)
    rows = torch.nn.functional.cross_entropy(p, tgt, reduction="none")
    rows.sum().backward()
    nll, lse, dx = co.ce(rx.detach().numpy(), E.numpy(), tgt.numpy(), None if neg is None else neg.numpy())
    np.testing.assert_allclose(nll, rows.detach().numpy(), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(dx, rx.grad.numpy(), rtol=2e-5, atol=2e-6)
