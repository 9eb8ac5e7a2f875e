"""BaseCVAE: what every slate CVAE shares (reference models/cvae.py:7-118).

Same constructor arguments, attributes (``docEmbed``, ``userEmbed``, ``device``, ``candidateFlag``,
``slate_size``, ``latent_size``, ``noUser``) and methods as the reference class, but the arithmetic is
done by the HIP kernels of libpcvae_hip.so (``pivotcvae_amd.ops``); there is no eager fallback.
"""
import torch
from torch import nn

from .. import ops
from .._hip import PREC_F32, PREC_NAMES


def _in_mlp_arith(fn):
    """run a model method with the model's MLP arithmetic selected (ops.mlp_arith): exact fp32 by default, bf16x3 / bf16x6 after
    ``set_mlp_precision(...)``.  Backward passes re-select the arithmetic of their forward by themselves."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        # (a module pickled before round 6 carries the boolean ``mlp_x3`` instead of ``mlp_precision``)
        with ops.mlp_arith(self.__dict__.get("mlp_precision", self.__dict__.get("mlp_x3", False))):
            return fn(self, *a, **k)
    return wrapped


# the TRAINING loss only: generation (recommend) and the reference-shaped entry points (forward / get_prior / encode / decode)
# always run the exact fp32 stacks - greedy item ids are index work and stay bit-exact whatever the training arithmetic is
MLP_ARITH_METHODS = ("loss", "_loss_fused")


def _normalized_rows(w):
    # F.normalize(w, p=2, dim=1): x / max(||x||_2, 1e-12)   (one-off at construction, models/cvae.py:31,39)
    return w / w.pow(2).sum(dim=1, keepdim=True).sqrt().clamp_min(1e-12)


class BaseCVAE(nn.Module):
    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        for name in MLP_ARITH_METHODS:   # every entry point that launches MLP GEMMs runs them in the model's arithmetic
            if name in cls.__dict__:
                setattr(cls, name, _in_mlp_arith(cls.__dict__[name]))

    def __init__(self, embeddings, u_embeddings, slate_size, latent_size, no_user, device, fine_tune=False):
        super().__init__()
        if fine_tune:
            raise NotImplementedError("fine_tune=True (trainable item/user tables) is outside the accelerated path: "
                                      "every reference model passes fine_tune=False (models/pivotcvae.py:319)")
        self.candidateFlag = False  # forward() scores candidate sets (True) or the whole catalog (False)
        self.slate_size = slate_size
        self.latent_size = latent_size
        self.noUser = no_user
        self.device = device
        with torch.no_grad():  # fresh, row-normalised, frozen copies (built where the source table lives)
            w = _normalized_rows(embeddings.weight.detach().float())
            self.docEmbed = nn.Embedding(w.shape[0], w.shape[1], _weight=w.contiguous())
            self.docEmbed.weight.requires_grad = False
            if not no_user:
                w = _normalized_rows(u_embeddings.weight.detach().float())
                self.userEmbed = nn.Embedding(w.shape[0], w.shape[1], _weight=w.contiguous())
                self.userEmbed.weight.requires_grad = False
        # precision of the [R,D]x[D,N] catalog contraction: "f32" (exact), "bf16x3", "bf16"
        self.catalog_precision = PREC_F32
        # arithmetic of the MLP stacks' GEMMs inside the training loss: "f32" (exact fp32 MFMA), "bf16x3", "bf16x6" (ops.MLP_PRECISIONS)
        self.mlp_precision = "f32"
        # table rows the GATHER kernels (sparse mask-train kernel, fused candidate kernel) read: the fp32 table - the reference's
        # arithmetic, whatever ``catalog_precision`` is - unless bf16 rows are asked for explicitly (set_gather_rows("bf16"))
        self.gather_rows_bf16 = False
        # Philox stream for eps when the caller does not supply one
        self.rng_seed = 0
        self._rng_offset = 0
        self._table = None

    # ---- plumbing -------------------------------------------------------------------------------
    def set_catalog_precision(self, name):
        self.catalog_precision = PREC_NAMES[name] if isinstance(name, str) else int(name)
        return self

    def set_mlp_precision(self, name):
        """arithmetic of the MLP GEMMs inside ``loss()`` (forward + backward of a train step): "f32" (exact fp32 MFMA, the default),
        "bf16x3" (operands split into bf16 hi + lo in registers, three bf16 MFMAs per product, fp32 accumulate: ELBO terms agree
        with fp32 to ~1e-6, parameter gradients to ~1e-4 of each tensor's scale) or "bf16x6" (three bf16 components per operand = the
        fp32 value exactly, six MFMAs per product: fp32-exact products, gradients inside the f32 kernel's tolerances)"""
        name = {"fp32": "f32"}.get(name, name)
        if name not in ops.MLP_PRECISIONS:
            raise ValueError(f"MLP arithmetic {name!r}: one of {ops.MLP_PRECISIONS}")
        self.mlp_precision = name
        return self

    @property
    def mlp_x3(self):   # the round-3 name of the switch (a module pickled before round 6 still carries it in its __dict__)
        d = self.__dict__
        return d.get("mlp_precision", "bf16x3" if d.get("mlp_x3") else "f32") == "bf16x3"

    def set_gather_rows(self, name):
        """which table the gather kernels read: "f32" (default: the fp32 table, the reference's arithmetic) or "bf16" (rows of the
        bf16 copy, widened exactly, fp32 products and sums: half the gathered bytes - the stated arithmetic of configs 3 / 5; loss
        terms then differ from fp32 rows at the bf16 catalog kernels' tolerance, ~2e-3 relative)"""
        if name not in ("f32", "fp32", "bf16"):
            raise ValueError(f"gather rows {name!r}: 'f32' or 'bf16'")
        self.gather_rows_bf16 = name == "bf16"
        return self

    def catalog_table(self):
        """CatalogTable over docEmbed.weight (caches the bf16 copies used by the MFMA bf16 modes)."""
        w = self.docEmbed.weight
        if self._table is None or self._table.weight is not w:
            self._table = ops.CatalogTable(w)
        return self._table

    def __getstate__(self):  # keep torch.save(model) working: device scratch is not part of the model
        state = self.__dict__.copy()
        state["_table"] = None
        return state

    def _mlp_layers(self, prefix, n):
        return [(getattr(self, f"{prefix}_{i + 1}").weight, getattr(self, f"{prefix}_{i + 1}").bias) for i in range(n)]

    def _head(self, name):
        m = getattr(self, name)
        return [(m.weight, m.bias)]

    def _next_offset(self, n):
        o = self._rng_offset
        self._rng_offset += int(n)
        return o

    # ---- reference API --------------------------------------------------------------------------
    def encode(self, emb, c, u_emb=None):
        raise NotImplementedError

    def decode(self, z, c, u_emb=None):
        raise NotImplementedError

    def get_prior(self, r, u=None):
        raise NotImplementedError

    def forward(self, s, r, candidates=None, u=None):
        raise NotImplementedError

    def recommend(self, r, u=None, return_item=False):
        raise NotImplementedError

    def log(self, logger):
        raise NotImplementedError

    def reparametrize(self, mu, logvar, eps=None):
        """z = eps * exp(0.5 logvar) + mu (models/cvae.py:79-83).

        eps=None draws N(0,1) inside the kernel (Philox keyed by ``rng_seed`` and a running offset);
        pass eps to reproduce a reference run exactly."""
        off = 0 if eps is not None else self._next_offset(mu.numel())
        z, self._last_eps = ops.reparam(mu, logvar, eps, seed=self.rng_seed, offset=off)
        return z

    def get_condition(self, r):
        return ops.condition(r, self.slate_size)

    def get_recommended_item(self, embeddings):
        return ops.catalog_argmax(embeddings.reshape(-1, self.feature_size), self.catalog_table(),
                                  prec=self.catalog_precision)

    def _rec_term(self, rx, s, n_neg, keep_mask, mask_seed, row_offset, inv_count, terms_only, candidates=None, n_items=None):
        """the reconstruction term of get_gen_loss from rx [B, S, D], fused loss + gradient, no logits in memory.

        ``candidates`` None: the mask-train branch (train_generative.py:58-59): full-catalog softmax CE with the downsample rule.
        ``candidates`` = an int Cn: the candidate-set branch (:52-56) with the sets of data_loader.py:46-58 drawn in-kernel (Philox
        stream keyed by (mask_seed, global slot)); a pair (sample_candidates [B, S, Cn], sample_targets [B, S]): sets as given.
        ``n_items``: the id range [0, n_items) of the in-kernel draw - the DATASET's ``max_iid + 1`` (data_loader.py:23, :46), which
        is smaller than the table when the table has rows no slate uses; None = the table's row count."""
        S, D = s.shape[1], self.feature_size
        rows = rx.reshape(-1, D)
        if candidates is not None:
            if isinstance(candidates, (tuple, list)):
                cand, tgt = candidates
                return ops.candidate_ce(rows, self.catalog_table(), cand=cand, cand_target=tgt, inv_count=inv_count,
                                        unit_upstream=terms_only, prec=self._gather_prec())
            return ops.candidate_ce(rows, self.catalog_table(), int(candidates), s.reshape(-1), mask_seed, row_offset * S,
                                    inv_count=inv_count, unit_upstream=terms_only, prec=self._gather_prec(), n_items=n_items)
        N = self.docEmbed.weight.shape[0]
        keep_prob = 1.0 if n_neg is None else float(n_neg) / N
        if keep_prob > 1.0:
            raise RuntimeError(f"n_neg={n_neg} exceeds the catalog size {N}")
        return ops.catalog_ce(rows, self.catalog_table(), s.reshape(-1), keep_prob, mask_seed, row_offset * S, keep_mask,
                              self.catalog_precision, inv_count, unit_upstream=terms_only, gather_bf16=self.gather_rows_bf16)

    def _gather_prec(self):
        from .._hip import PREC_BF16
        return PREC_BF16 if self.__dict__.get("gather_rows_bf16", False) else PREC_F32

    def _user_rows(self, u, B):
        return None if self.noUser else ops.gather_rows(self.userEmbed.weight, u.reshape(-1)).reshape(B, -1)

    def sample_encoding(self, s, r, u=None):
        B = s.shape[0]
        cond = self.get_condition(r)
        emb = ops.gather_rows(self.docEmbed.weight, s.reshape(-1), group=s.shape[1])
        return self.encode(emb, cond, self._user_rows(u, B))

    # the north_star text calls the generation method generate(); the reference's name is recommend()
    def generate(self, r, u=None, return_item=False, **kw):
        return self.recommend(r, u, return_item=return_item, **kw)
