"""Host-side mirror of the hot part of reference train_generative.py (downsample :36-42, get_gen_loss :44-65,
the optimisation step :120-137) on top of the HIP path.

Two ways in:
  * ``get_gen_loss(batch_data, model, lossFun, beta, n_neg)`` - same signature and return value as the
    reference function; the mask-train branch goes through ``model.loss`` (fused full-catalog CE: no
    [B*S, N] tensor), so it also works where the reference cannot (N = 1M: 328 GB of logits).
  * ``Trainer`` - zero_grad / loss / backward / (all-reduce) / Adam as one object, single GPU or one
    process per GPU over RCCL.
"""
import weakref

import numpy as np
import torch

from . import ops
from .optim import FlatAdam


def downsample(pred, slate, n_neg=1000.0, seed=0, row_offset=0):
    """Reference semantics on a DENSE logits tensor (small catalogs only; train_generative.py:36-42): mask = onehot(target) OR
    Bernoulli(n_neg / N); masked-out logits become 0.  Kept for callers that hold a dense ``pred`` (the reference's own
    get_gen_loss recipe on forward()); one kernel, the mask drawn in it (the dense masked CE kernels' Philox stream) - the fused
    losses apply the same rule without ever forming ``pred``."""
    if n_neg > pred.shape[1]:
        raise RuntimeError(f"n_neg={n_neg} exceeds the catalog size {pred.shape[1]}")
    return ops.downsample_dense(pred, slate.reshape(-1), float(n_neg) / pred.shape[1], seed, row_offset)


def _batch_to_device(batch_data, device):
    slates = torch.as_tensor(np.asarray(batch_data["slates"]), dtype=torch.long).to(device)
    users = torch.as_tensor(np.asarray(batch_data["users"]), dtype=torch.long).to(device)
    targets = torch.as_tensor(np.asarray(batch_data["responses"])).to(torch.float).to(device)
    return slates, users, targets


def get_gen_loss(batch_data, model, lossFun, beta, n_neg=1000, eps=None, seed=0, row_offset=0):
    """-> (loss, recLoss, KLD), as reference get_gen_loss.  ``lossFun`` is only used on the candidate path.

    Candidate path (``model.candidateFlag``): the reference's dataset builds the candidate sets per item in a Python loop on the
    host (data_loader.py:46-58).  A batch that carries ``sample_candidates`` / ``sample_targets`` is used as given; one that
    does not gets them drawn on the device (``ops.candidate_draw``: ``model.nCandidate`` columns - what
    ``trainset.init_sampling(nneg)`` sets in the reference, default ``n_neg`` - Philox stream (seed, row_offset + slot))."""
    slates, users, targets = _batch_to_device(batch_data, model.device)
    if model.candidateFlag:
        given = "sample_candidates" in batch_data
        if given:
            cand = torch.as_tensor(np.asarray(batch_data["sample_candidates"]), dtype=torch.long).to(model.device)
            tgt = torch.as_tensor(np.asarray(batch_data["sample_targets"]), dtype=torch.long).to(model.device)
        # the reference passes nn.CrossEntropyLoss(): that case is ONE fused launch (ids, rows and logits never exist in memory);
        # any other lossFun is called as given on the materialised candidate logits of forward()
        plain_ce = isinstance(lossFun, torch.nn.CrossEntropyLoss) and lossFun.weight is None and \
            lossFun.reduction == "mean" and lossFun.ignore_index == -100 and getattr(lossFun, "label_smoothing", 0.0) == 0.0
        # ids are drawn from the DATASET's range [0, max_iid + 1) (data_loader.py:23, :46) when the model carries it
        # (``model.candidateIdRange``, set by train_on_dataset from ``trainset.max_iid``); the table's row count otherwise
        n_items = getattr(model, "candidateIdRange", None)
        if plain_ce and hasattr(model, "loss"):
            cands = (cand, tgt) if given else int(getattr(model, "nCandidate", n_neg))
            return model.loss(slates, targets, users, beta, eps=eps, mask_seed=seed, row_offset=row_offset, candidates=cands,
                              n_items=None if given else n_items)
        pMu, pLogvar = model.get_prior(targets, users)
        if not given:
            N = model.docEmbed.weight.shape[0] if n_items is None else int(n_items)
            cand, tgt = ops.candidate_draw(slates, N, int(getattr(model, "nCandidate", n_neg)), seed=seed,
                                           row_offset=row_offset * slates.shape[1])
        pred, _rx, _z, _emb, mu, logvar = model.forward(slates, targets, candidates=cand, u=users, eps=eps)
        recLoss = lossFun(pred, tgt.reshape(-1))
        KLD = ops.kld(mu, logvar, pMu, pLogvar)
        return recLoss + beta * KLD, recLoss, KLD
    N = model.docEmbed.weight.shape[0]
    return model.loss(slates, targets, users, beta, n_neg=None if n_neg == N else n_neg, eps=eps, mask_seed=seed,
                      row_offset=row_offset)


@torch.no_grad()
def recommendation_test(model, resp_model, bs, n_test_trial=100, seed=0, capture_graph=False):
    """The in-loop evaluation of reference train_generative.py:169-195, entirely on the device.

    For each of ``n_test_trial`` trials: sample ``bs`` users, and for the five contexts "i+1 desired clicks"
    (i = 0..4) generate greedy slates with ``model.recommend`` and score them with the response model; the expected
    number of clicks of a slate is sum_s sigmoid(logit).  Returns a [5, 3] tensor of (min, mean, max) expected clicks
    averaged over the trials - the three numbers the reference logs per context - without any host synchronisation
    inside the loop (the reference does 15 ``.cpu()`` copies per trial).

    ``capture_graph`` (round 6): the 500 generate + score calls of an evaluation are the SAME ~35 launches on the same shapes -
    they are captured once as a hipGraph (static context / users / eps buffers; eps drawn outside the graph from the model's own
    Philox stream at the positions the eager calls would use) and replayed.  Same numbers as the eager loop, bit for bit; it pays
    where the chain is launch-bound (config 2: 2.4x; tools/gen_graph_probe.py) and changes nothing where the catalog kernels
    dominate (configs 3-5).  Sampled inference rules (``spi``) keep the eager loop: their sampler's stream position is a kernel
    argument."""
    from .env.response_model import sample_users
    device = model.docEmbed.weight.device
    acc = torch.zeros(5, 3, dtype=torch.float32, device=device)
    graph = st = None
    if capture_graph and getattr(model, "INFER_RULE", "pi") == "pi" and n_test_trial > 0:
        Z = model.latent_size
        st = dict(ctx=torch.zeros(bs, 5, dtype=torch.float32, device=device), users=torch.zeros(bs, dtype=torch.int64, device=device),
                  eps=torch.zeros(bs, Z, dtype=torch.float32, device=device))
        users_shape = tuple(sample_users(resp_model, bs, seed=seed, offset=0).shape)
        st["users"] = st["users"].reshape(users_shape)

        def chain():
            slates, _mu = model.recommend(st["ctx"], st["users"], return_item=True, eps=st["eps"])
            logits = resp_model(slates.view(bs, -1), st["users"])
            return ops.click_stats(logits)[1]

        try:
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            warm = {}
            with torch.cuda.stream(side), ops.workspace_holder(warm):   # warm-up: scratch buffers, table copies, kernel attributes
                for _ in range(2):
                    chain()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            bufs = {}
            for key, buf in warm.items():
                if key[2] not in bufs or bufs[key[2]].numel() < buf.numel():
                    bufs[key[2]] = buf
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"), ops.workspace_holder(_FixedWorkspace(bufs)):
                st["stats"] = chain()
            st["ws"] = bufs
        except Exception as e:   # capture is an optimisation: fall back to the eager loop, and say so
            import warnings
            warnings.warn(f"hipGraph capture of the recommendation test failed ({e}); running eagerly")
            torch.cuda.synchronize()
            graph = None
    for k in range(n_test_trial):
        users = sample_users(resp_model, bs, seed=seed, offset=k * bs)
        context = torch.zeros(bs, 5, dtype=torch.float32, device=device)
        for i in range(5):
            context[:, i] = 1
            if graph is not None:
                st["ctx"].copy_(context)
                st["users"].copy_(users)
                # the eps an eager recommend() would draw in its reparametrisation kernel: same stream, same position
                ops.philox_normal_(st["eps"], seed=model.rng_seed, offset=model._next_offset(bs * model.latent_size))
                graph.replay()
                acc[i] += st["stats"]
                continue
            slates, _mu = model.recommend(context, users, return_item=True)
            logits = resp_model(slates.view(bs, -1), users)
            _nc, stats = ops.click_stats(logits)
            acc[i] += stats
    return acc / n_test_trial


class _FixedWorkspace(dict):
    """workspace holder for graph capture: whatever stream asks, it gets the pre-allocated buffer of that user (catalog kernels,
    grouped GEMMs) - never a new allocation inside the capture; a request the warm-up did not make is a bug and raises"""

    def __init__(self, bufs):
        super().__init__()
        self.bufs = bufs   # tag -> buffer

    def get(self, key, default=None):
        if key[2] not in self.bufs:
            raise RuntimeError(f"{key[2]} scratch requested during hipGraph capture but the warm-up used none")
        return self.bufs[key[2]]

    def __setitem__(self, key, value):
        raise RuntimeError("scratch grew during hipGraph capture (the warm-up runs the same shapes: this is a bug)")


class Trainer:
    """One optimisation step of reference train_generative.py:124-134, data-parallel aware.

    world_size > 1: every rank holds a replica, takes ``B_global / world_size`` slates, scales its reconstruction
    term by 1/world_size (it is a MEAN over rows, the KL is a SUM: SURVEY.md 8e), and the flat gradient buffer
    is summed with ONE all-reduce (RCCL over xGMI on the GPUs).  ``loss_fn(model, s, r, u, **kw)`` is
    injectable so the collective logic can be exercised on CPU with gloo in tests.
    """

    REDUCE_TIMING = None   # bench.py: (begin() -> token, end(token)) around the gradient all-reduce (HIP events on the launch stream)

    def __init__(self, model, lr, beta, n_neg=None, process_group=None, loss_fn=None, optimizer=None,
                 capture_graph=False, world_size=None, rank=0, resident_batch=False, n_candidate=None, n_items=None):
        import torch.distributed as dist
        if n_neg is not None and n_candidate is not None:
            raise ValueError("n_neg (mask-train) and n_candidate (candidate sets) are the two branches of get_gen_loss: pass one")
        self.model, self.beta, self.n_neg = model, float(beta), n_neg
        # the reference's DEFAULT mode (no --mask_train): candidate sets of n_candidate columns per slot, drawn in the fused
        # kernel from a stream keyed by (step, GLOBAL slot) - independent of the world size, like the masks and eps
        # Sets GIVEN per step (what a batch of the reference's dataset carries: sample_candidates [B, S, Cn], sample_targets [B, S])
        # are an argument of step() / local_phase(): ``candidates=(cand, tgt)``; such a step always launches eagerly.
        # (a pair passed HERE is kept as the default of steps that pass none - the round-5 interface; it is validated per step too)
        self.n_candidate = n_candidate if n_candidate is None or isinstance(n_candidate, (tuple, list)) else int(n_candidate)
        # the id range [0, n_items) of the in-kernel draw: the dataset's max_iid + 1 (data_loader.py:23, :46); None = the table's rows
        self.n_items = None if n_items is None else int(n_items)
        if self.n_items is not None and not 0 < self.n_items <= model.docEmbed.weight.shape[0]:
            raise ValueError(f"n_items={n_items} must lie in (0, {model.docEmbed.weight.shape[0]}] (the table's row count)")
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.pg = process_group
        self.world = self.dist.get_world_size(process_group) if self.dist else 1
        self.rank = self.dist.get_rank(process_group) if self.dist else 0
        # ``world_size`` WITHOUT an initialised process group: this object plays rank ``rank`` of ``world_size`` and the caller does
        # the reduction between local_phase() and finish_phase() (sum the ranks' ``opt.grad_ext`` buffers - what all_reduce(SUM)
        # leaves in every rank's buffer).  tests/test_hip_data_parallel.py runs W simulated ranks on ONE GPU this way.
        self.external_reduce = False
        if world_size is not None and self.dist is None:
            self.world, self.rank, self.external_reduce = int(world_size), int(rank), int(world_size) > 1
        self.opt = optimizer if optimizer is not None else FlatAdam(model, lr)
        self._own_loss = loss_fn is None
        self.loss_fn = loss_fn or (lambda m, s, r, u, **kw: m.loss(s, r, u, **kw))
        self._seed = None   # (1, beta) as device scalars: backward is seeded with them instead of forming rec + beta * KLD first
        self.global_step = 0
        # hipGraph capture of zero-grad + forward + backward (the ~90 small launches of a step become one graph
        # launch; matters when a rank only holds B/8 slates).  eps is drawn OUTSIDE the graph into a static buffer
        # because kernel arguments (Philox offsets) are frozen at capture; the in-kernel Bernoulli mask (n_neg < N)
        # has the same problem, so that mode stays eager - and so do the candidate-set mode (its draw is keyed the same way) and
        # the SAMPLED pivot rules (spt / sgt): their sampler takes (seed, row offset) as kernel arguments too, a replayed graph
        # would redraw the same pivots every step.
        # The all-reduce and Adam stay outside the graph.
        # Round 5: the in-kernel draws no longer block capture - the sparse mask kernel, the fused candidate kernel and the rejection
        # sampler can read their step-dependent word (seed / stream position) from DEVICE memory (``self._words``, written by one
        # tiny launch before each replay).  What still stays eager: the dense masked kernels (n_neg / N > ops.SPARSE_MAX_KEEP_PROB),
        # candidate sets handed in per step, an injected loss_fn.
        # (whether the CURRENT mode can be captured is decided when the capture is attempted: _capturable())
        self.capture_graph = bool(capture_graph) and loss_fn is None
        self._words = None   # int64 [2] on the device: (mask / candidate seed = global step, sampler stream position) of the replayed step
        self._graph = None
        self._static = None
        # hipGraph replay reads its own static input buffers.  Default: the caller's s / r / u are copied into them on EVERY step
        # (three small launches).  ``resident_batch=True`` is the caller's promise that a batch passed again as the same tensor
        # objects has not been rewritten in between - an epoch over one resident batch, the benchmark - and skips the copies then.
        # It is opt-in because tensor._version cannot see writes through raw pointers (this library's own out= kernels, .data).
        self.resident_batch = bool(resident_batch)
        self._stats, self._in_tail = None, False
        self.capture_failed = None   # the reason, if hipGraph capture was asked for and fell back to eager launches
        self.on_capture_failed = None   # callable(reason): train_on_dataset routes the fallback into its logger

    def shard(self, *tensors):
        """Contiguous shard of a global batch for this rank (+ its offset in the global batch)."""
        B = tensors[0].shape[0]
        if B % self.world:
            raise ValueError(f"global batch {B} not divisible by world size {self.world}")
        per = B // self.world
        lo = self.rank * per
        return [t[lo:lo + per] for t in tensors], lo

    def _capturable(self):
        """can a step in the CURRENT mode be replayed as a hipGraph?  Not: the dense masked kernels (n_neg / N above the sparse
        kernel's range: their keep set is keyed by a by-value seed), candidate sets handed in per step, an injected loss_fn."""
        N = self.model.docEmbed.weight.shape[0]
        masked_ok = self.n_neg is None or ops.sparse_ce_applies(float(self.n_neg) / N, N)
        return self._own_loss and masked_ok and not isinstance(self.n_candidate, (tuple, list))

    @staticmethod
    def _check_given_sets(candidates, s):
        """(sample_candidates [B, S, Cn], sample_targets [B, S]) of THIS batch - shapes are checked against the batch, so that sets
        left over from another step cannot be used silently"""
        if not isinstance(candidates, (tuple, list)) or len(candidates) != 2:
            raise ValueError("candidates: a pair (sample_candidates [B, S, Cn], sample_targets [B, S])")
        cand, tgt = candidates
        B, S = s.shape
        if cand.dim() != 3 or tuple(cand.shape[:2]) != (B, S) or tuple(tgt.shape) != (B, S):
            raise ValueError(f"candidate sets {tuple(cand.shape)} / targets {tuple(tgt.shape)} do not belong to a batch of shape "
                             f"[{B}, {S}] (expected [B, S, Cn] and [B, S])")
        return cand, tgt

    def _local(self, s, r, u, eps, row_offset, eps_offset, words=None, candidates=None):
        """zero-grad + local loss + backward (this rank's shard).  ``words``: the device words of a captured step (the seed and the
        sampler position are then read from them at run time; the by-value ones are what an eager step passes).  ``candidates``:
        this step's given sets (else the trainer's mode: ``n_candidate`` drawn in-kernel, or mask-train)."""
        B, S = s.shape
        self.opt.zero_grad()
        kw = dict(beta=self.beta, n_neg=self.n_neg, eps=eps, row_offset=row_offset, inv_count=1.0 / (B * S * self.world),
                  eps_offset=eps_offset, mask_seed=self.global_step if words is None else words[0:1])
        if candidates is None and isinstance(self.n_candidate, (tuple, list)):
            candidates = self.n_candidate
        if candidates is not None:
            kw["candidates"] = self._check_given_sets(candidates, s)
            kw["n_neg"] = None
        elif self.n_candidate is not None:
            kw["candidates"] = self.n_candidate
            if self.n_items is not None:
                kw["n_items"] = self.n_items
        if self._own_loss and getattr(self.model, "TRAIN_RULE", "gt") in ("spt", "sgt"):
            # sampled pivots: the sampler's stream position is the slate's GLOBAL index in the run, like eps - independent of how
            # the batch is sharded over ranks (captured step: this shard's offset by value + the step's base from the device word)
            kw["sample_offset"] = eps_offset // self.model.latent_size if words is None else (row_offset, words[1:2])
        if self._own_loss:
            # d(rec + beta KLD) = 1 d rec + beta d KLD: seeding backward with the two constants saves the mul / add / fill / mul
            # launches of forming the sum and differentiating it (the logged loss is formed in step(), one launch)
            if self._seed is None:
                self._seed = (ops.register_unit_seed(torch.ones((), dtype=torch.float32, device=s.device)),
                              torch.full((), float(self.beta), dtype=torch.float32, device=s.device), float(self.beta))
            elif self._seed[2] != float(self.beta):   # beta changed (annealing): in place, a captured graph keeps the address
                self._seed[1].fill_(float(self.beta))
                self._seed = (self._seed[0], self._seed[1], float(self.beta))
            _, rec, kld = self.model.loss(s, r, u, terms_only=True, **kw)
            torch.autograd.backward([rec, kld], [self._seed[0], self._seed[1]])
            return None, rec.detach(), kld.detach()
        loss, rec, kld = self.loss_fn(self.model, s, r, u, **kw)
        loss.backward()
        return loss.detach(), rec.detach(), kld.detach()

    def _capture(self, s, r, u, row_offset):
        B = s.shape[0]
        Z = self.model.latent_size
        st = dict(s=s.clone(), r=r.clone(), u=u.clone(), eps=torch.zeros(B, Z, dtype=torch.float32, device=s.device),
                  row_offset=row_offset)
        dynamic = self.n_neg is not None or self.n_candidate is not None or getattr(self.model, "TRAIN_RULE", "gt") in ("spt", "sgt")
        if dynamic and self._words is None:
            self._words = torch.zeros(2, dtype=torch.int64, device=s.device)
        st["words"] = self._words if dynamic else None
        st["mode"] = self._mode()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        # the catalog kernels' scratch pointer is baked into the graph: capture under this trainer's OWN buffer (kept in
        # self._static), not the module-wide grow-only cache that a later, larger call may reallocate
        warm = {}
        with torch.cuda.stream(side), ops.workspace_holder(warm):  # warm-up: scratch buffers, bf16 table copies, kernel attributes
            for _ in range(2):
                self._local(st["s"], st["r"], st["u"], st["eps"], row_offset, 0, st["words"])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        st["ws"] = {}
        for key, buf in warm.items():   # per user (catalog kernels, grouped GEMMs): the largest buffer the warm-up allocated
            if key[2] not in st["ws"] or st["ws"][key[2]].numel() < buf.numel():
                st["ws"][key[2]] = buf
        graph = torch.cuda.CUDAGraph()
        # thread_local: CUDA / HIP calls other threads make while this one captures (RCCL's watchdog polling its events) must not
        # invalidate the capture
        with torch.cuda.graph(graph, capture_error_mode="thread_local"), ops.workspace_holder(_FixedWorkspace(st["ws"])):
            st["out"] = self._local(st["s"], st["r"], st["u"], st["eps"], row_offset, 0, st["words"])
        self._graph, self._static = graph, st

    def _mode(self):
        """what a captured step has baked in besides shapes: the loss mode and the pivot rule (a step in another mode runs eagerly)"""
        nc = self.n_candidate
        return (self.n_neg, "given" if isinstance(nc, (tuple, list)) else nc, self.n_items, getattr(self.model, "TRAIN_RULE", "gt"),
                getattr(self.model, "catalog_precision", None), getattr(self.model, "mlp_precision", None),
                getattr(self.model, "gather_rows_bf16", None))

    def prepare_graph(self, s, r, u, row_offset=0):
        """Capture the hipGraph of a step for this batch shape now (no parameter update, no collective: two warm-up passes of
        zero-grad + loss + backward, then the captured pass).  local_phase() calls it on the first step; a caller that times steps
        calls it BEFORE its timed region (bench.py with --warmup 0).  A failed capture falls back to eager launches, loudly."""
        if not self.capture_graph or self._graph is not None:
            return
        if not self._capturable():   # evaluated NOW, from the current mode (n_neg / n_candidate may have changed since __init__)
            self.capture_graph = False
            return
        try:
            self._capture(s, r, u, row_offset)
        except Exception as e:  # capture is an optimisation: fall back to eager launches, and say so
            import warnings
            warnings.warn(f"hipGraph capture failed ({e}); running eagerly")
            if self.on_capture_failed is not None:
                self.on_capture_failed(str(e))
            torch.cuda.synchronize()
            self.capture_graph, self._graph = False, None
            self.capture_failed = str(e)

    def local_phase(self, s, r, u, eps=None, global_batch=None, row_offset=0, candidates=None):
        """Phase 1 of a step - everything a rank does on its own: zero-grad, local loss (reconstruction term scaled by
        1 / (B_local S world), eps / mask / sampler streams at GLOBAL slate indices), backward into the flat gradient buffer, and
        the rank's (loss, rec, KLD) record written behind the gradients (``opt.tail``).  ``candidates``: THIS batch's given sets
        (sample_candidates [B_local, S, Cn], sample_targets [B_local, S]); such a step is launched eagerly."""
        B, S = s.shape
        W = self.world
        gb = global_batch if global_batch is not None else B * W
        Z = self.model.latent_size
        eps_offset = (self.global_step * gb + row_offset) * Z
        if candidates is None:
            self.prepare_graph(s, r, u, row_offset)
        if candidates is None and self.capture_graph and self._graph is not None and tuple(s.shape) == tuple(self._static["s"].shape) \
                and row_offset == self._static["row_offset"] and self._static["mode"] == self._mode():
            st = self._static
            # the graph reads its own input buffers: copy the caller's batch in, unless the caller promised (resident_batch) that
            # tensors it passes again are unchanged - then only weak references to the last batch are kept, nothing is pinned
            src = (s, r, u)
            same = self.resident_batch and st.get("src") is not None and \
                all(ref() is a and a._version == v for a, (ref, v) in zip(src, st["src"]))
            if not same:
                st["s"].copy_(s)
                st["r"].copy_(r)
                st["u"].copy_(u)
                st["src"] = [(weakref.ref(t), t._version) for t in src] if self.resident_batch else None
            if eps is None:
                ops.philox_normal_(st["eps"], seed=self.model.rng_seed, offset=eps_offset)
            else:
                st["eps"].copy_(eps)
            if st["words"] is not None:   # this step's seed and the sampler's base position (eager: mask_seed, sample_offset - row_offset)
                ops.set_words_(st["words"], self.global_step, eps_offset // Z - row_offset)
            self._graph.replay()
            loss, rec, kld = st["out"]
        else:
            loss, rec, kld = self._local(s, r, u, eps, row_offset, eps_offset, candidates=candidates)
        tail = getattr(self.opt, "tail", None)
        self._stats, self._in_tail = None, False
        host_record = lambda: torch.stack([torch.add(rec, kld, alpha=float(self.beta)) if loss is None else loss, rec, kld])
        if (self.world > 1 or self.dist is not None) and tail is not None and tail.numel() >= 3:
            # (rec + beta KLD, rec, KLD) as ONE record behind the gradients: the gradient all-reduce sums the statistics with them -
            # a second, latency-bound collective for three floats would cost as much as the first
            if rec.is_cuda and loss is None:
                ops.elbo_pack_(rec, kld, self.beta, tail)
            else:   # an injected loss_fn: ITS loss is what was back-propagated and is what gets logged (it may carry extra terms)
                tail[:3].copy_(host_record())
            self._in_tail = True
        elif rec.is_cuda and loss is None:
            self._stats = ops.elbo_pack_(rec, kld, self.beta, torch.empty(3, dtype=torch.float32, device=rec.device))
        else:
            self._stats = host_record()

    def reduce_phase(self):
        """Phase 2: ONE all_reduce(SUM) over gradients + statistics (RCCL over xGMI on the GPUs).  Also with a 1-rank group: the
        collective path is then the one that is exercised.  Simulated ranks (``external_reduce``): the caller sums the buffers."""
        if self.dist is None:
            return
        timing = self.REDUCE_TIMING
        tok = timing[0]() if timing else None
        if self._in_tail:
            self.dist.all_reduce(self.opt.grad_ext, group=self.pg)
        else:
            self.dist.all_reduce(self.opt.grad, group=self.pg)
            self.dist.all_reduce(self._stats, group=self.pg)
        if timing:
            timing[1](tok)

    def finish_phase(self):
        """Phase 3: identical Adam on every rank.  -> (loss, recLoss, KLD) of the GLOBAL batch as device scalars."""
        stats = self.opt.tail[:3].clone() if self._in_tail else self._stats   # the tail is zeroed with the gradients
        self.opt.step()
        self.global_step += 1
        return stats[0], stats[1], stats[2]

    def step(self, s, r, u, eps=None, global_batch=None, row_offset=0, candidates=None):
        """s, r, u: THIS rank's shard.  Returns (loss, recLoss, KLD) as device scalars of the GLOBAL batch; nothing is
        synchronised with the host.  ``candidates``: this batch's given candidate sets (local_phase)."""
        if self.external_reduce:
            raise RuntimeError("this Trainer plays one of several simulated ranks: call local_phase(), sum the ranks' "
                               "opt.grad_ext yourself, then finish_phase()")
        self.local_phase(s, r, u, eps, global_batch, row_offset, candidates)
        self.reduce_phase()
        return self.finish_phase()


# ---------------------------------------------------------------------------------------------------------------------------
# epoch loop
# ---------------------------------------------------------------------------------------------------------------------------
def _dataset_arrays(ds):
    """(slates [L, S], users [L, 1], responses [L, S]) of a data_loader.UserSlateResponseDataset-like object or a dict"""
    get = (lambda k: ds[k]) if isinstance(ds, dict) else (lambda k: getattr(ds, k))
    slates = torch.as_tensor(np.asarray(get("slates")), dtype=torch.long)
    users = torch.as_tensor(np.asarray(get("users")), dtype=torch.long).reshape(slates.shape[0], -1)[:, :1]
    resp = torch.as_tensor(np.asarray(get("responses"))).to(torch.float)
    return slates, users, resp


def candidate_loss(model, s, r, u, beta, n_candidate, seed=0, row_offset=0, inv_count=None, eps=None, eps_offset=None, n_items=None):
    """The candidate path of get_gen_loss as a Trainer loss function: ``model.loss(candidates=n_candidate)`` - candidate sets drawn
    in the fused kernel, streams pinned to global slate indices (independent of sharding), the mean scaled by ``inv_count``.
    -> (loss, recLoss, KLD)"""
    return model.loss(s, r, u, beta, eps=eps, mask_seed=seed, row_offset=row_offset, inv_count=inv_count, eps_offset=eps_offset,
                      candidates=int(n_candidate), n_items=n_items)


def candidate_loss_materialised(model, s, r, u, beta, n_candidate, seed=0, row_offset=0, inv_count=None, eps=None, eps_offset=None,
                                n_items=None):
    """The same loss the way the reference computes it - ids [B, S, Cn] (``ops.candidate_draw``), candidate logits through
    ``forward(candidates=...)`` (K9 scores), dense CE - kept as the cross-check of the fused kernel (tests) and for callers that
    want ``forward()``'s ``p``.  Same streams, same result up to fp32 summation order."""
    B, S = s.shape
    N = model.docEmbed.weight.shape[0] if n_items is None else int(n_items)
    cand, tgt = ops.candidate_draw(s, N, n_candidate, seed=seed, row_offset=row_offset * S)
    if eps is None:
        eps = torch.empty(B, model.latent_size, dtype=torch.float32, device=s.device)
        off = model._next_offset(B * model.latent_size) if eps_offset is None else int(eps_offset)
        ops.philox_normal_(eps, seed=model.rng_seed, offset=off)
    was = model.candidateFlag
    model.candidateFlag = True
    try:
        pMu, pLogvar = model.get_prior(r, u)
        pred, _rx, _z, _emb, mu, logvar = model.forward(s, r, candidates=cand, u=u, eps=eps)
    finally:
        model.candidateFlag = was
    rec = ops.dense_ce(pred, tgt.reshape(-1), inv_count)
    kld = ops.kld(mu, logvar, pMu, pLogvar)
    return rec + beta * kld, rec, kld


def train_on_dataset(trainset, valset, model, model_path, logger, resp_model, bs, epochs, lr, decay, beta, n_neg=1000,
                     n_test_trial=100, seed=0, process_group=None, capture_graph=False, trainer=None, val_loss_fn=None,
                     eval_fn=None):
    """Counterpart of reference train_generative.train_on_dataset (:67-214) for catalogs of any size, single GPU or one process
    per GPU (data parallel over the batch, ONE all-reduce per step).

    Per epoch, like the reference: shuffled training batches -> zero_grad / get_gen_loss / backward / Adam (default
    ``n_neg = 1000`` in mask-train mode, :126 + :44; candidate sets of ``trainset.nCandidate`` columns when
    ``model.candidateFlag``); validation under no_grad at ``n_neg = trainset.nCandidate`` (:157); the in-loop recommendation
    test against ``resp_model`` (:169-195); the whole-module pickle when the validation loss improves by more than 1e-3
    (:198-202) and the final move to CPU (:209-213).  The same lines are logged.  ``decay`` is logged and NOT applied, as in
    the reference (:103).

    What differs, on purpose: the datasets' index arrays are uploaded once and stay resident in HBM; batches are slices of a
    permutation drawn ON the device (same on every rank); candidate sets and Bernoulli masks are drawn in the kernels; nothing
    is synchronised with the host inside an epoch (the reference does a ``loss.item()`` per step, :128) - the per-batch
    losses accumulate on the device and are read once per epoch.  Under data parallelism every rank takes bs / world slates of
    each batch (a last batch that does not divide is cut to a multiple of the world size), rank 0 logs and saves.

    Candidate ids are drawn from the DATASET's range, as in the reference (data_loader.py:23 ``max_iid = np.max(slates)``, :46
    ``randint(max_iid + 1, ...)``): ``trainset.max_iid + 1`` when the dataset has it (a table may have more rows than the slates
    use - the simulators build n_item + 1, env/response_model.py:30), the table's row count otherwise.
    ``model_path=None``: no pickle is written (a benchmark of the loop itself).  ``history`` also carries the seconds each
    epoch's training loop and validation pass took (measured at the host reads the loop does anyway).

    ``trainer`` / ``val_loss_fn`` / ``eval_fn`` are injection points for the CPU tests of the loop logic."""
    import time

    import torch.distributed as dist
    device = model.docEmbed.weight.device
    use_dist = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(process_group) if use_dist else 0
    world = dist.get_world_size(process_group) if use_dist else 1

    def log(msg):
        if rank == 0:
            logger.log(msg)

    log("----------------------------------------")
    log("Train user response model as simulator")
    log("\tbatch size: " + str(bs))
    log("\tnumber of epoch: " + str(epochs))
    log("\tlearning rate: " + str(lr))
    log("\tweight decay: " + str(decay))
    log("\tbeta: " + str(beta))
    log("----------------------------------------")
    if rank == 0:
        model.log(logger)
    log("----------------------------------------")

    ds_get = (lambda k, d: trainset.get(k, d)) if isinstance(trainset, dict) else (lambda k, d: getattr(trainset, k, d))
    n_cand = int(ds_get("nCandidate", n_neg))
    N = model.docEmbed.weight.shape[0]
    max_iid = ds_get("max_iid", None)
    n_items = None if max_iid is None else int(max_iid) + 1   # the id range of the candidate draw (data_loader.py:23, :46)
    if n_items is not None and not 0 < n_items <= N:
        raise ValueError(f"trainset.max_iid + 1 = {n_items} does not fit the model's table of {N} rows")
    model.candidateIdRange = n_items   # get_gen_loss reads it (a caller that evaluates batches through the reference's entry point);
    #                                      None - a dataset without max_iid - resets what an earlier dataset may have left
    S = model.slate_size
    tr_s, tr_u, tr_r = (t.to(device) for t in _dataset_arrays(trainset))
    va_s, va_u, va_r = (t.to(device) for t in _dataset_arrays(valset))
    L = tr_s.shape[0]

    if trainer is None:
        if model.candidateFlag:   # the reference's default mode: the fused candidate kernel inside the trainer's own (seeded) step
            trainer = Trainer(model, lr=lr, beta=beta, n_neg=None, process_group=process_group, n_candidate=n_cand,
                              capture_graph=capture_graph, n_items=n_items)
        else:
            trainer = Trainer(model, lr=lr, beta=beta, n_neg=None if n_neg >= N else n_neg, process_group=process_group,
                              capture_graph=capture_graph)
    if getattr(trainer, "on_capture_failed", "absent") is None:   # a capture that falls back to eager launches is a logged event
        trainer.on_capture_failed = lambda why: log(f"hipGraph capture failed ({why}): training continues with eager launches")
    if val_loss_fn is None:
        def val_loss_fn(m, s, r, u, row_offset):   # forward only, n_neg = the dataset's candidate count (reference :157)
            if m.candidateFlag:
                return candidate_loss(m, s, r, u, beta, n_cand, seed=0x5641, row_offset=row_offset, n_items=n_items)
            return m.loss(s, r, u, beta, n_neg=None if n_cand >= N else n_cand, mask_seed=0x5641, row_offset=row_offset)
    run_eval = eval_fn is not None or resp_model is not None   # neither given: the recommendation test is skipped
    if eval_fn is None:
        eval_fn = lambda m: recommendation_test(m, resp_model, bs, n_test_trial=n_test_trial, seed=seed, capture_graph=capture_graph)

    dropped = sum((min(bs, L - lo) % world) for lo in range(0, L, bs))
    if dropped:   # runs at different world sizes see the same slates only if every batch divides
        log(f"data parallel over {world} ranks: {dropped} of {L} training slates per epoch are cut (batches are trimmed to a multiple "
            f"of the world size)")
    gen = torch.Generator(device=device)
    best_val = float("inf")
    temper = 2
    history = {"train": [], "val": [], "train_seconds": [], "val_seconds": []}
    for epoch in range(epochs):
        log("Epoch " + str(epoch + 1))
        t_epoch = time.perf_counter()
        gen.manual_seed((seed << 20) + epoch)            # the same permutation on every rank
        perm = torch.randperm(L, device=device, generator=gen)
        acc = torch.zeros((), dtype=torch.float32, device=device)
        n_batches = 0
        for lo in range(0, L, bs):
            idx = perm[lo:lo + bs]
            gb = (idx.shape[0] // world) * world
            if gb == 0:
                continue
            idx = idx[:gb]
            per = gb // world
            mine = idx[rank * per:(rank + 1) * per]
            loss, _rec, _kld = trainer.step(tr_s[mine], tr_r[mine], tr_u[mine], global_batch=gb, row_offset=rank * per)
            acc += loss.to(acc.device)
            n_batches += 1
        history["train"].append(float(acc) / max(n_batches, 1))   # the epoch's one host read of the training loop: it has drained
        history["train_seconds"].append(time.perf_counter() - t_epoch)
        t_val = time.perf_counter()
        log("train loss: " + str(history["train"][-1]))
        if epoch == 0 and getattr(trainer, "capture_failed", None):
            log("hipGraph capture was asked for and failed (" + trainer.capture_failed + "): every step is launched eagerly")

        # validation: every rank evaluates its slice of each batch; the sums are combined once
        sums = torch.zeros(4, dtype=torch.float64, device=device)   # loss, rec, kld, batches
        with torch.no_grad():
            for lo in range(0, va_s.shape[0], bs):
                hi = min(lo + bs, va_s.shape[0])
                if (lo // bs) % world != rank:
                    continue
                l_, r_, k_ = val_loss_fn(model, va_s[lo:hi], va_r[lo:hi], va_u[lo:hi], lo)
                sums += torch.stack([l_.double(), r_.double(), k_.double(), torch.ones((), dtype=torch.float64, device=device)]).to(device)
        if use_dist:
            dist.all_reduce(sums, group=process_group)
        v_loss, v_rec, v_kld = (float(x) / max(float(sums[3]), 1.0) for x in sums[:3])
        history["val_seconds"].append(time.perf_counter() - t_val)
        history["val"].append(v_loss)
        log("validation Loss: " + str(v_loss) + " = " + str(v_rec) + " + " + str(beta) + " * " + str(v_kld))

        # recommendation test
        if run_eval:
            stats = eval_fn(model)
            if stats is not None:
                stats = stats.detach().cpu()
                for i in range(stats.shape[0]):
                    log("Expected response (" + str(i + 1) + "): " + str(stats[i, 0].numpy()) + "; " + str(stats[i, 1].numpy()) +
                        "; " + str(stats[i, 2].numpy()))

        # save best model (reference :198-208; early termination is commented out there too)
        if epoch == 0 or v_loss < best_val - 1e-3:
            if rank == 0 and model_path is not None:
                torch.save(model, open(model_path, "wb"))
            log("Save best model")
            temper = 3
            best_val = v_loss
        else:
            temper -= 1
            log("Temper down to " + str(temper))
    if use_dist:
        dist.barrier(group=process_group)
    if rank == 0 and model_path is not None:
        log("Move model to cpu before saving")
        best = torch.load(open(model_path, "rb"), weights_only=False)
        best.to("cpu")
        best.device = "cpu"
        torch.save(best, open(model_path, "wb"))
    return history
