"""UserPivotCVAE and its 7 pivot-rule variants (reference models/pivotcvae.py:33-461).

A slate is generated as: encoder MLP -> (mu, logvar) -> z -> pivot-selection MLP (PSM) -> pivot item ->
slate-completion MLP (SCM) -> S-1 more slot vectors -> dot with the item catalog.  The 8 registry entries
differ only in how the pivot is chosen during training / inference:

    gt  ground-truth first item      pt / pi  catalog argmax of the PSM output
    sgt sampled around ground truth  spt / spi sampled from sigmoid(PSM output . E^T)

The arithmetic runs in libpcvae_hip.so (see pivotcvae_amd.ops); this file is host-side orchestration that
keeps the reference's class names, constructor, attributes, state_dict keys and method signatures.
"""
import torch
from torch import nn

from .. import ops
from .cvae import BaseCVAE
from .listcvae import _stack


class UserPivotCVAE(BaseCVAE):
    TRAIN_RULE = "gt"
    INFER_RULE = "pi"

    def __init__(self, embeddings, u_embeddings, slate_size, feature_size, latent_size, condition_size,
                 encoder_struct, psm_struct, scm_struct, prior_struct, no_user, device, fine_tune=False):
        super().__init__(embeddings, u_embeddings, slate_size, latent_size, no_user, device, fine_tune)
        u = 0 if no_user else feature_size
        assert encoder_struct[0] == slate_size * feature_size + condition_size + u
        assert psm_struct[0] == latent_size + condition_size + u
        assert psm_struct[-1] == feature_size
        assert scm_struct[0] == latent_size + condition_size + feature_size + u
        assert scm_struct[-1] == (slate_size - 1) * feature_size
        assert prior_struct[0] == condition_size + u
        self.feature_size = feature_size
        self.condition_size = condition_size
        self.encoderStruct = encoder_struct
        self.psmStruct = psm_struct
        self.scmStruct = scm_struct
        self.priorStruct = prior_struct
        self._n_enc = _stack(self, "enc", encoder_struct)
        self.encmu = nn.Linear(encoder_struct[-1], latent_size)
        self.enclogvar = nn.Linear(encoder_struct[-1], latent_size)
        self._n_psm = _stack(self, "psm", psm_struct)
        self._n_scm = _stack(self, "scm", scm_struct)
        self._n_prior = _stack(self, "prior", prior_struct)
        self.priorMu = nn.Linear(prior_struct[-1], latent_size)
        self.priorLogvar = nn.Linear(prior_struct[-1], latent_size)
        self.pivot_override = None  # test hook: feed recorded Categorical draws back in
        self.last_pivot = None      # pivot item ids chosen by the most recent decode()
        self.to(self.device)

    def flat_param_groups(self):
        """parameters that want to be adjacent in the optimiser's flat buffer: the two heads of the encoder and of the prior (their
        weights back to back are one [2 Z, K] operand: mu and logvar out of ONE GEMM, ops.mlp_heads_packed)"""
        return [[self.encmu.weight, self.enclogvar.weight], [self.encmu.bias, self.enclogvar.bias],
                [self.priorMu.weight, self.priorLogvar.weight], [self.priorMu.bias, self.priorLogvar.bias]]

    def params_without_grad(self):
        """trainable parameters that never receive a gradient under ANY pivot rule: the PSM stack (its output is
        either ignored or feeds a non-differentiable argmax / sample).  torch.optim.Adam skips them (grad is None)."""
        return [p for i in range(self._n_psm) for p in getattr(self, f"psm_{i + 1}").parameters()]

    # ---- encoder / prior -----------------------------------------------------------------------
    def encode(self, emb, c, u_emb=None):
        x = ops.concat([emb, c] if self.noUser else [emb, c, u_emb])
        return ops.mlp_heads(x, self._mlp_layers("enc", self._n_enc), self._head("encmu")[0], self._head("enclogvar")[0])

    def _prior_from(self, cond, u_emb):
        x = cond if self.noUser else ops.concat([cond, u_emb])
        return ops.mlp_heads(x, self._mlp_layers("prior", self._n_prior), self._head("priorMu")[0], self._head("priorLogvar")[0])

    def get_prior(self, r, u=None):
        return self._prior_from(self.get_condition(r), self._user_rows(u, r.shape[0]))

    # ---- pivot selection ------------------------------------------------------------------------
    def _pivot_index(self, rule, pivot_output, true_pivot, sample_offset=None, pivot_row=None):
        """Item id of the pivot for one of the rules gt / pt|pi / spt|spi / sgt (never differentiable).
        ``sample_offset``: stream position (a global slate index) of the sampled rules; None = this model's running offset.
        ``pivot_row``: the ground-truth pivot's table row [B, D] when the caller already holds it (sgt: no second gather)."""
        if rule == "gt":
            return true_pivot
        if self.pivot_override is not None and rule in ("spt", "spi", "sgt"):
            return self.pivot_override
        table = self.catalog_table()
        if rule in ("pt", "pi"):
            return ops.catalog_argmax(pivot_output, table, prec=self.catalog_precision)
        if rule in ("spt", "spi"):
            query = pivot_output.detach()
        else:  # sgt: scores of the ground-truth pivot's own embedding against the catalog
            query = pivot_row if pivot_row is not None else ops.gather_rows(self.docEmbed.weight, true_pivot)
        B = query.shape[0]
        if sample_offset is None:
            off = self._next_offset(B)
        else:   # an int, or (int, device word added to it) for a hipGraph-replayed step
            off = sample_offset if isinstance(sample_offset, (tuple, list)) else int(sample_offset)
        return ops.catalog_sample(query, table, seed=self.rng_seed ^ 0x5A17, row_offset=off)

    def pick_pivot(self, pivot_output, true_pivot, sample_offset=None, pivot_row=None):
        """-> pivot embedding [B, D]; rule = TRAIN_RULE when a true pivot is given, else INFER_RULE."""
        training = len(true_pivot) > 0
        p = self._pivot_index(self.TRAIN_RULE if training else self.INFER_RULE, pivot_output, true_pivot, sample_offset, pivot_row)
        self.last_pivot = p
        return ops.gather_rows(self.docEmbed.weight, p)

    def decode(self, z, c, u_emb=None, true_pivot=[], sample_offset=None, pivot_row=None):
        B = z.shape[0]
        x = ops.concat([z, c] if self.noUser else [z, c, u_emb])
        pivot_output = ops.mlp(x, self._mlp_layers("psm", self._n_psm), last_linear=True)
        pivot_emb = self.pick_pivot(pivot_output, true_pivot, sample_offset, pivot_row)
        return self._complete(z, c, u_emb, pivot_emb)

    def _complete(self, z, c, u_emb, pivot_emb):
        B = z.shape[0]
        x = ops.concat([z, c, pivot_emb] if self.noUser else [z, c, pivot_emb, u_emb])
        rest = ops.mlp(x, self._mlp_layers("scm", self._n_scm), last_linear=True)
        return ops.concat([pivot_emb, rest]).reshape(B, self.slate_size, self.feature_size)

    # ---- reference entry points -----------------------------------------------------------------
    def forward(self, s, r, candidates=None, u=None, eps=None):
        """-> (p, rx, z, emb, z_mu, z_logvar) exactly like the reference; p is the DENSE [B*S, N] logits (or the
        [B*S, Cn] candidate logits), so this entry point is for catalogs where that matrix fits."""
        B = s.shape[0]
        cond = self.get_condition(r)
        emb = ops.gather_rows(self.docEmbed.weight, s.reshape(-1), group=s.shape[1])
        u_emb = self._user_rows(u, B)
        z_mu, z_logvar = self.encode(emb, cond, u_emb)
        z = self.reparametrize(z_mu, z_logvar, eps)
        rx = self.decode(z, cond, u_emb=u_emb, true_pivot=s[:, 0])
        prox = rx.reshape(-1, self.feature_size)
        if self.candidateFlag:
            p = ops.candidate_scores(prox, self.docEmbed.weight, candidates.reshape(prox.shape[0], -1))
        else:
            p = ops.dense_scores(prox, self.docEmbed.weight)
        return p, rx, z, emb, z_mu, z_logvar

    def loss(self, s, r, u, beta, n_neg=None, eps=None, keep_mask=None, mask_seed=0, row_offset=0, inv_count=None,
             eps_offset=None, terms_only=False, sample_offset=None, candidates=None, n_items=None):
        """Fused counterpart of train_generative.get_gen_loss -> (loss, recLoss, KLD).  ``candidates`` None: the mask-train
        branch; an int Cn / a pair (sample_candidates, sample_targets): the candidate-set branch - the reference's default mode
        (train_generative.py:52-57, 270-274) - from ONE fused launch (BaseCVAE._rec_term, ops.candidate_ce); ``n_items`` = the id range
        of the in-kernel draw (the dataset's ``max_iid + 1``, data_loader.py:23, :46; default: the table's row count).

        The [B*S, N] logits never exist: the full-catalog softmax CE (with the reference's downsample
        semantics when n_neg < N) and its gradient come from one streaming pass over the catalog.
        ``row_offset`` = index of this shard's first slate in the global batch (data parallel), so the
        Philox mask / eps streams do not depend on the world size; ``inv_count`` overrides the 1/(B*S)
        of the 'mean' (a rank passes 1/(B_local*S*world_size)); ``sample_offset`` pins the stream position of the
        sampled pivot rules (spt / sgt) the same way (global slate index of this shard's first slate).
        With TRAIN_RULE == "gt" the PSM is skipped: its output is unused and it never receives a gradient in
        the reference either.
        """
        B, S = s.shape
        if self.TRAIN_RULE == "gt" and self.FUSED_TRAIN_PATH and r.shape[1] == S and \
                ops.heads_adjacent(self.encmu, self.enclogvar) and ops.heads_adjacent(self.priorMu, self.priorLogvar):
            return self._loss_fused(s, r, u, beta, n_neg, eps, keep_mask, mask_seed, row_offset, inv_count, eps_offset, terms_only,
                                    candidates, n_items)
        cond = self.get_condition(r)
        emb = ops.gather_rows(self.docEmbed.weight, s.reshape(-1), group=S)
        u_emb = self._user_rows(u, B)
        pmu, plv = self._prior_from(cond, u_emb)
        z_mu, z_logvar = self.encode(emb, cond, u_emb)
        # reparametrize + KLD as one autograd node (their backward is one kernel); values as ops.reparam / ops.kld
        if eps is None:  # in-kernel Philox; a data-parallel caller pins the stream position explicitly
            off = self._next_offset(B * self.latent_size) if eps_offset is None else int(eps_offset)
            z, self._last_eps, k = ops.latent(z_mu, z_logvar, pmu, plv, None, seed=self.rng_seed, offset=off)
        else:
            z, self._last_eps, k = ops.latent(z_mu, z_logvar, pmu, plv, eps)
        if self.TRAIN_RULE == "gt":
            self.last_pivot = s[:, 0]
            # the ground-truth pivot's row is the first D columns of the slate's gathered rows: no second gather
            pivot_emb = emb[:, : self.feature_size]
            rx = self._complete(z, cond, u_emb, pivot_emb)
        else:
            # no torch op on the step's path (round 5's kernel trace showed one strided-copy kernel per spt / sgt step: the
            # ``s[:, 0].contiguous()`` that stood here): pt / spt only need to know THAT a true pivot exists; sgt needs its table row,
            # which is the first D columns of the slate's gathered rows (one library copy kernel, no second gather)
            pivot_row = None
            if self.TRAIN_RULE == "sgt":
                pivot_row = torch.empty(B, self.feature_size, dtype=torch.float32, device=emb.device)
                ops.copy2d(emb[:, : self.feature_size], pivot_row)
            rx = self.decode(z, cond, u_emb=u_emb, true_pivot=s[:, 0], sample_offset=sample_offset, pivot_row=pivot_row)
        rec = self._rec_term(rx, s, n_neg, keep_mask, mask_seed, row_offset, inv_count, terms_only, candidates, n_items)
        if terms_only:   # the caller seeds backward with (1, beta) and forms the logged loss itself: no mul / add launches
            return None, rec, k
        return rec + beta * k, rec, k

    FUSED_TRAIN_PATH = True   # tests switch it off to compare the two routes

    def _loss_fused(self, s, r, u, beta, n_neg, eps, keep_mask, mask_seed, row_offset, inv_count, eps_offset, terms_only,
                    candidates=None, n_items=None):
        """loss() for the ground-truth pivot rule with a trainer's flat parameter buffer attached: the same arithmetic in fewer,
        larger launches - one kernel assembles the three stack inputs (condition, gathers, concatenations), each stack's two heads
        are one N = 2 Z GEMM, reparametrize + KL are one kernel that writes z straight into the slate-completion input, and that
        stack's last GEMM writes slots 1.. of rx next to the pivot row."""
        B, S = s.shape
        D, Z = self.feature_size, self.latent_size
        enc_in, prior_in, scm_in, rx = ops.assemble_inputs(self.docEmbed.weight, None if self.noUser else self.userEmbed.weight,
                                                           s, r, u, Z)
        # the encoder and the prior share only their inputs: layer i of both is one grouped launch, forward and backward
        y_enc, y_prior = ops.mlp_heads_packed_pair(
            (enc_in, self._mlp_layers("enc", self._n_enc), self._head("encmu")[0], self._head("enclogvar")[0]),
            (prior_in, self._mlp_layers("prior", self._n_prior), self._head("priorMu")[0], self._head("priorLogvar")[0]))
        if eps is None:
            off = self._next_offset(B * Z) if eps_offset is None else int(eps_offset)
            scm_x, self._last_eps, k = ops.latent_packed(y_enc, y_prior, scm_in, None, seed=self.rng_seed, offset=off, Z=Z)
        else:
            scm_x, self._last_eps, k = ops.latent_packed(y_enc, y_prior, scm_in, eps, Z=Z)
        self.last_pivot = s[:, 0]
        rx = ops.mlp_into(scm_x, self._mlp_layers("scm", self._n_scm), rx, D, grad_cols=Z)   # only z carries a gradient
        rec = self._rec_term(rx, s, n_neg, keep_mask, mask_seed, row_offset, inv_count, terms_only, candidates, n_items)
        if terms_only:
            return None, rec, k
        return rec + beta * k, rec, k

    def recommend(self, r, u=None, return_item=False, random_pivot=False, eps=None):
        B = r.shape[0]
        cond = self.get_condition(r)
        u_emb = self._user_rows(u, B)
        z_mu, z_logvar = self._prior_from(cond, u_emb)
        z = self.reparametrize(z_mu, z_logvar, eps)
        rx = self.decode(z, cond, u_emb)
        if return_item:
            return self.get_recommended_item(rx.reshape(-1, self.feature_size)), z_mu
        return rx, z_mu

    def log(self, logger):
        logger.log("\tfeature size: " + str(self.feature_size))
        logger.log("\tslate size: " + str(self.slate_size))
        logger.log("\tz size: " + str(self.latent_size))
        logger.log("\tcondition size: " + str(self.condition_size))
        logger.log("\tuser is ignored: " + str(self.noUser))
        logger.log("\tencoder struct: " + str(self.encoderStruct))
        logger.log("\tpsm struct: " + str(self.psmStruct))
        logger.log("\tscm struct: " + str(self.scmStruct))
        logger.log("\tprior struct: " + str(self.priorStruct))
        logger.log("\tdevice: " + str(self.device))


def _variant(name, train_rule, infer_rule, doc):
    return type(name, (UserPivotCVAE,), {"TRAIN_RULE": train_rule, "INFER_RULE": infer_rule, "__doc__": doc,
                                         "__module__": __name__})


UserPivotCVAE2 = _variant("UserPivotCVAE2", "pt", "pi", "best pivot in training and inference")
UserPivotCVAE_PrePermute = _variant("UserPivotCVAE_PrePermute", "spt", "pi", "sampled pivot in training, best at inference")
UserPivotCVAE_PrePermute2 = _variant("UserPivotCVAE_PrePermute2", "sgt", "pi", "sampled ground truth in training, best at inference")
UserPivotCVAE_PrePermute3 = _variant("UserPivotCVAE_PrePermute3", "gt", "spi", "ground truth in training, sampled at inference")
UserPivotCVAE_PrePermute4 = _variant("UserPivotCVAE_PrePermute4", "pt", "spi", "best pivot in training, sampled at inference")
UserPivotCVAE_PrePermute5 = _variant("UserPivotCVAE_PrePermute5", "spt", "spi", "sampled pivot in training and inference")
UserPivotCVAE_PrePermute6 = _variant("UserPivotCVAE_PrePermute6", "sgt", "spi", "sampled ground truth in training, sampled at inference")

PIVOTCVAE_MODELS = {
    "pivotcvae_gt_pi": UserPivotCVAE, "pivotcvae_pt_pi": UserPivotCVAE2,
    "pivotcvae_spt_pi": UserPivotCVAE_PrePermute, "pivotcvae_sgt_pi": UserPivotCVAE_PrePermute2,
    "pivotcvae_gt_spi": UserPivotCVAE_PrePermute3, "pivotcvae_pt_spi": UserPivotCVAE_PrePermute4,
    "pivotcvae_spt_spi": UserPivotCVAE_PrePermute5, "pivotcvae_sgt_spi": UserPivotCVAE_PrePermute6,
}

# BASELINE.json's north_star says "models.pivotcvae.PivotCVAE": an alias of the benchmark variant
PivotCVAE = UserPivotCVAE
