"""UserListCVAEWithPrior (reference models/listcvae.py:8-199): one decoder MLP emits all S slot vectors."""
import torch
from torch import nn

from .. import ops
from .cvae import BaseCVAE


def _stack(module, prefix, struct):
    for i in range(len(struct) - 1):
        lin = nn.Linear(struct[i], struct[i + 1])
        nn.init.kaiming_uniform_(lin.weight)
        module.add_module(f"{prefix}_{i + 1}", lin)
    return len(struct) - 1


class UserListCVAEWithPrior(BaseCVAE):
    def __init__(self, embeddings, u_embeddings, slate_size, feature_size, latent_size, condition_size,
                 encoder_struct, decoder_struct, prior_struct, no_user, device, fine_tune=False):
        super().__init__(embeddings, u_embeddings, slate_size, latent_size, no_user, device, fine_tune)
        self.feature_size = feature_size
        self.condition_size = condition_size
        self.encoderStruct = encoder_struct
        self.decoderStruct = decoder_struct
        self.priorStruct = prior_struct
        u = 0 if no_user else feature_size
        assert encoder_struct[0] == slate_size * feature_size + condition_size + u
        assert decoder_struct[0] == latent_size + condition_size + u
        assert decoder_struct[-1] == slate_size * feature_size
        assert prior_struct[0] == condition_size + u
        self._n_enc = _stack(self, "enc", encoder_struct)
        self.encmu = nn.Linear(encoder_struct[-1], latent_size)
        self.enclogvar = nn.Linear(encoder_struct[-1], latent_size)
        self._n_dec = _stack(self, "dec", decoder_struct)
        self._n_prior = _stack(self, "prior", prior_struct)
        self.priorMu = nn.Linear(prior_struct[-1], latent_size)
        self.priorLogvar = nn.Linear(prior_struct[-1], latent_size)
        self.to(self.device)

    def encode(self, emb, c, u_emb=None):
        x = ops.concat([emb, c] if self.noUser else [emb, c, u_emb])
        return ops.mlp_heads(x, self._mlp_layers("enc", self._n_enc), self._head("encmu")[0], self._head("enclogvar")[0])

    def decode(self, z, c, u_emb=None):
        x = ops.concat([z, c] if self.noUser else [z, c, u_emb])
        return ops.mlp(x, self._mlp_layers("dec", self._n_dec), last_linear=True)

    def _prior_from(self, cond, u_emb):
        x = cond if self.noUser else ops.concat([cond, u_emb])
        return ops.mlp_heads(x, self._mlp_layers("prior", self._n_prior), self._head("priorMu")[0], self._head("priorLogvar")[0])

    def get_prior(self, r, u=None):
        return self._prior_from(self.get_condition(r), self._user_rows(u, r.shape[0]))

    def forward(self, s, r, candidates=None, u=None, eps=None):
        B = s.shape[0]
        cond = self.get_condition(r)
        emb = ops.gather_rows(self.docEmbed.weight, s.reshape(-1), group=s.shape[1])
        u_emb = self._user_rows(u, B)
        z_mu, z_logvar = self.encode(emb, cond, u_emb)
        z = self.reparametrize(z_mu, z_logvar, eps)
        rx = self.decode(z, cond, u_emb)
        prox = rx.reshape(-1, self.feature_size)
        if self.candidateFlag:
            p = ops.candidate_scores(prox, self.docEmbed.weight, candidates.reshape(prox.shape[0], -1))
        else:
            p = ops.dense_scores(prox, self.docEmbed.weight)
        return p, rx, z, emb, z_mu, z_logvar

    def loss(self, s, r, u, beta, n_neg=None, eps=None, keep_mask=None, mask_seed=0, row_offset=0, inv_count=None,
             eps_offset=None, terms_only=False, candidates=None, n_items=None):
        """Fused counterpart of train_generative.get_gen_loss: neither the [R, N] logits of the mask-train branch nor the
        [R, Cn] ids / rows / logits of the candidate branch (``candidates``: BaseCVAE._rec_term) are ever materialised.
        -> (loss, recLoss, KLD)."""
        B = s.shape[0]
        cond = self.get_condition(r)
        emb = ops.gather_rows(self.docEmbed.weight, s.reshape(-1), group=s.shape[1])
        u_emb = self._user_rows(u, B)
        pmu, plv = self._prior_from(cond, u_emb)
        z_mu, z_logvar = self.encode(emb, cond, u_emb)
        if eps is None:
            off = self._next_offset(B * self.latent_size) if eps_offset is None else int(eps_offset)
            z, self._last_eps, k = ops.latent(z_mu, z_logvar, pmu, plv, None, seed=self.rng_seed, offset=off)
        else:
            z, self._last_eps, k = ops.latent(z_mu, z_logvar, pmu, plv, eps)
        rx = self.decode(z, cond, u_emb)
        rec = self._rec_term(rx, s, n_neg, keep_mask, mask_seed, row_offset, inv_count, terms_only, candidates, n_items)
        if terms_only:   # the caller seeds backward with (1, beta) and forms the logged loss itself: no mul / add launches
            return None, rec, k
        return rec + beta * k, rec, k

    def recommend(self, r, u=None, return_item=False, eps=None):
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
        logger.log("\tdecoder struct: " + str(self.decoderStruct))
        logger.log("\tprior struct: " + str(self.priorStruct))
        logger.log("\tdevice: " + str(self.device))
