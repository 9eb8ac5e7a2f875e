"""world_size-2 data-parallel step on CPU (gloo): shard -> local loss with the 1/W rule -> ONE all-reduce of the
flat gradient buffer -> identical Adam on every rank  ==  the single-process step on the whole batch.

The product's compute needs the GPU, so the test injects the CPU oracle as ``loss_fn`` and a CPU Adam over
the same flat buffers; what is under test is the Trainer's collective logic (pivotcvae_amd/train_generative.py).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pivotcvae_oracle as orc
from tests.helpers import load


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cpu_model(g):
    import pivotcvae_amd as pa
    m, st = g.meta, g.meta["structs"]
    doc = torch.nn.Embedding.from_pretrained(g.t("raw_doc"))
    usr = torch.nn.Embedding.from_pretrained(g.t("raw_user"))
    model = pa.PIVOTCVAE_MODELS[m["model"]](doc, usr, m["S"], m["D"], m["Z"], m["S"] + 1, st["enc"], st["psm"],
                                           st["scm"], st["prior"], False, "cpu")
    model.load_state_dict(g.sd)
    return model


def _oracle_loss_fn(cfg, eps_full):
    def fn(model, s, r, u, beta, n_neg, eps, row_offset, inv_count, eps_offset, mask_seed):
        sd = dict(model.named_parameters())
        B = s.shape[0]
        e = eps_full[row_offset:row_offset + B]
        pmu, plv = orc.prior(sd, cfg, r, u)
        f = orc.forward(sd, cfg, s, r, u, e)
        nll = torch.nn.functional.cross_entropy(f["p"], s.reshape(-1), reduction="sum")
        rec = nll * inv_count  # the rank's share of the global MEAN
        k = orc.kld(f["z_mu"], f["z_logvar"], pmu, plv)  # SUM: no rescaling
        return rec + beta * k, rec, k
    return fn


def _make_trainer(model, g, eps_full):
    from pivotcvae_amd.optim import FlatAdam
    from pivotcvae_amd.train_generative import Trainer

    class CpuFlatAdam(FlatAdam):
        def step(self, grad_scale=1.0):
            self.t += 1
            st = {"t": self.t - 1, "m/w": self.m, "v/w": self.v}
            new = orc.adam_step({"w": self.flat}, {"w": self.grad * grad_scale}, st, self.lr)
            self.flat.copy_(new["w"])

    return Trainer(model, lr=g.meta["lr"], beta=g.meta["beta"], loss_fn=_oracle_loss_fn(g.cfg(), eps_full),
                   optimizer=CpuFlatAdam(model, g.meta["lr"]))


def _worker(rank, world, port, name, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    g = load(name)
    B = 6  # divisible by 2
    s, r, u, eps = g.t("s")[:B], g.t("r")[:B], g.t("u")[:B], g.t("full/eps")[:B]
    model = _cpu_model(g)
    tr = _make_trainer(model, g, eps)
    (ss, rr, uu), lo = tr.shard(s, r, u)
    stats = []
    for _ in range(2):
        stats.append([float(x) for x in tr.step(ss, rr, uu, global_batch=B, row_offset=lo)])
    out[rank] = ({k: v.detach().clone() for k, v in model.state_dict().items()}, stats, lo)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_step_equals_single_process_step():
    name = "pivotcvae_gt_pi_user"
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, name, out), nprocs=world, join=True)
    g = load(name)
    B = 6
    s, r, u, eps = g.t("s")[:B], g.t("r")[:B], g.t("u")[:B], g.t("full/eps")[:B]
    # single-process reference: the oracle on the whole batch
    sd, state, want_stats = g.sd, {}, []
    for _ in range(2):
        (l, rec, k), grads = orc.loss_and_grads(sd, g.cfg(), s, r, u, eps, g.meta["beta"])
        want_stats.append([l, rec, k])
        sd = orc.adam_step(sd, grads, state, g.meta["lr"])
    (sd0, st0, lo0), (sd1, st1, lo1) = out[0], out[1]
    assert (lo0, lo1) == (0, 3)
    np.testing.assert_allclose(st0, want_stats, rtol=1e-5)  # all-reduced ELBO terms == whole-batch terms
    np.testing.assert_allclose(st1, st0, rtol=0, atol=0)
    for k in sd:
        assert torch.equal(sd0[k], sd1[k]), k                # replicas stay bit-identical
        torch.testing.assert_close(sd0[k], sd[k], rtol=2e-5, atol=1e-7)


def test_shard_rejects_indivisible_batches():
    from pivotcvae_amd.train_generative import Trainer
    g = load("pivotcvae_gt_pi_user")
    tr = _make_trainer(_cpu_model(g), g, g.t("full/eps"))
    tr.world, tr.rank = 2, 1
    (a,), lo = tr.shard(torch.arange(8).reshape(8, 1))
    assert lo == 4 and a.reshape(-1).tolist() == [4, 5, 6, 7]
    with pytest.raises(ValueError):
        tr.shard(torch.arange(7).reshape(7, 1))
