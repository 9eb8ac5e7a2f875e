"""The learned click model used for in-loop evaluation (reference env/response_model.py:15-87) and the uniform user
sampler (:10-13), forward-only on the HIP path.

``Environment`` (owner of the RAW, un-normalised ``docEmbed`` / ``userEmbed`` tables that the CVAE copies and
normalises), ``UserResponseModel_MLP.forward`` (no-grad, the in-loop evaluation of ``train_generative.py:169-195``),
``forward_train`` (the same forward with hand-written backward kernels, for ``pivotcvae_amd.pretrain_env``) and
``sample_users``; and the simulators ``URM`` / ``URM_P`` / ``URM_P_MR`` (:97-154, 264-323) as in-loop evaluators - the
``resp_model`` of every ``--dataset urm*`` run (train_generative.py:247,185): forward only, one fused kernel.  The simulators'
dataset generation is out of scope.
"""
import math

import torch
from torch import nn

from .. import ops
from .._hip import ACT_NONE, ACT_RELU


def sample_users(environment, batch_size, seed=0, offset=0):
    """Uniform user ids in [0, maxUserId]; drawn on the device from a Philox stream instead of the reference's
    host-side ``torch.multinomial(ones)`` (same distribution; that stream cannot be matched)."""
    return ops.philox_randint(batch_size, environment.maxUserId + 1, environment.docEmbed.weight.device, seed, offset)


class Environment(nn.Module):
    def __init__(self, maxIID, maxUID, f_size, s_size, device, no_user):
        super().__init__()
        self.maxItemId = maxIID
        self.maxUserId = maxUID
        self.featureSize = f_size
        self.slateSize = s_size
        self.device = device
        self.noUser = no_user
        a = math.sqrt(2.0 / f_size)
        self.docEmbed = nn.Embedding(maxIID + 1, f_size)
        self.docEmbed.weight.data.uniform_(-a, a)
        if not no_user:
            self.userEmbed = nn.Embedding(maxUID + 1, f_size)
            self.userEmbed.weight.data.uniform_(-a, a)


class UserResponseModel_MLP(Environment):
    """Click logits [B, S] of a slate for a user: normalised slate vector (+ normalised user vector) -> ReLU MLP."""

    def __init__(self, maxIID, maxUID, f_size, s_size, struct, device, no_user):
        super().__init__(maxIID, maxUID, f_size, s_size, device, no_user)
        if no_user:
            assert struct[0] == s_size * f_size
        else:
            assert struct[0] == (s_size + 1) * f_size
        assert struct[-1] == s_size
        self._n = len(struct) - 1
        for i in range(self._n):
            lin = nn.Linear(struct[i], struct[i + 1])
            nn.init.kaiming_uniform_(lin.weight)
            self.add_module("mlp_" + str(i + 1), lin)

    def forward_train(self, slates, users):
        """forward() with autograd (reference pretrain_env.py:82-83).  A quirk of the reference is reproduced: the training
        batches carry users as [B, 1], so ``F.normalize(self.userEmbed(users), p=2, dim=1)`` (env/response_model.py:81)
        normalises a [B, 1, D] tensor over its size-1 axis - every component becomes x / max(|x|, 1e-12) = sign(x) and
        the user table receives a zero gradient (weight decay still moves it).  With users of shape [B] (what
        sample_users returns) the user rows are L2-normalised as expected."""
        B = slates.shape[0]
        S, D = self.slateSize, self.featureSize
        d = ops.normalize_rows(ops.embedding_rows(self.docEmbed.weight, slates, group=S))           # [B, S*D]
        if self.noUser:
            x = d
        else:
            urows = ops.embedding_rows(self.userEmbed.weight, users)                              # [B, D]
            if users.dim() == 2:   # [B, 1]: normalisation over the singleton axis
                uemb = ops.normalize_rows(urows.reshape(B * D, 1)).reshape(B, D)
            else:
                uemb = ops.normalize_rows(urows)
            x = ops.concat([d, uemb])
        layers = [(getattr(self, f"mlp_{i}").weight, getattr(self, f"mlp_{i}").bias) for i in range(1, self._n + 1)]
        return ops.mlp_relu(x, layers)

    @torch.no_grad()
    def forward(self, slates, users):
        B = slates.shape[0]
        S, D = self.slateSize, self.featureSize
        width = S * D if self.noUser else (S + 1) * D
        x = torch.empty(B, width, dtype=torch.float32, device=self.docEmbed.weight.device)
        # the WHOLE concatenated slate vector is normalised (not each item): env/response_model.py:78
        ops.gather_rows(self.docEmbed.weight, slates.reshape(-1), out=x[:, : S * D], group=S)
        ops.normalize_rows_(x[:, : S * D])
        if not self.noUser:
            if users.dim() == 2:  # [B, 1] batches: the reference normalises over the singleton axis (see forward_train)
                urows = ops.gather_rows(self.userEmbed.weight, users.reshape(-1))
                ops.normalize_rows_(urows.reshape(B * D, 1))
                ops.copy2d(urows, x[:, S * D:])
            else:
                ops.gather_rows(self.userEmbed.weight, users.reshape(-1), out=x[:, S * D:])
                ops.normalize_rows_(x[:, S * D:])
        for i in range(1, self._n + 1):
            lin = getattr(self, f"mlp_{i}")
            x = ops.linear_fwd_raw(x, lin.weight, lin.bias, ACT_RELU if i < self._n else ACT_NONE)
        return x


class URM(Environment):
    """Matrix-factorisation click simulator (env/response_model.py:97-154): p = sigmoid(<normalised item, raw user> + item bias
    + user bias) per slot.  Used as the evaluator of the in-loop recommendation test; ``forward`` returns the [B, S] scores
    ``core_forward`` computes first (train_generative.py:185 then applies a sigmoid on top - for these models a second one)."""

    def __init__(self, maxIID, maxUID, slate_size, latent_size, device, no_user):
        super().__init__(maxIID, maxUID, latent_size, slate_size, device, no_user)
        assert not no_user
        self.itemBias = nn.Embedding(maxIID + 1, 1)
        self.itemBias.weight.data = torch.zeros_like(self.itemBias.weight.data)
        self.userBias = nn.Embedding(maxUID + 1, 1)
        self.userBias.weight.data = torch.zeros_like(self.userBias.weight.data)

    def _extras(self):
        return {}

    def to(self, *args, **kwargs):
        out = super().to(*args, **kwargs)
        out.device = args[0] if args else kwargs.get("device", out.device)
        return out

    @torch.no_grad()
    def forward(self, slates, users):
        return ops.urm_forward(self.docEmbed.weight, self.itemBias.weight, self.userEmbed.weight, self.userBias.weight,
                               slates.reshape(slates.shape[0], -1), users, **self._extras())


class URM_P(URM):
    """URM + positional bias (env/response_model.py:264-303): a per-position constant and a user-dependent term
    U[u] . posDependentBias, where the [S, D] buffer is READ AS [D, S] (the reference's ``.view(featureSize, slateSize)``)."""

    def __init__(self, maxIID, maxUID, slate_size, latent_size, device, no_user, p_bias_max, p_bias_min):
        super().__init__(maxIID, maxUID, slate_size, latent_size, device, no_user)
        self.p_bias_max, self.p_bias_min = p_bias_max, p_bias_min
        self.posBias = torch.tensor([p_bias_max - i * (p_bias_max - p_bias_min) / slate_size for i in range(slate_size)],
                                    dtype=torch.float32).to(device)
        a = math.sqrt(0.5 / latent_size)
        self.posDependentBias = torch.empty(slate_size * latent_size, dtype=torch.float32).uniform_(-a, a) \
            .reshape(slate_size, latent_size).to(device)

    def _extras(self):
        return dict(pos_bias=self.posBias, pos_dep=self.posDependentBias)

    def to(self, *args, **kwargs):
        out = super().to(*args, **kwargs)
        out.posBias = out.posBias.to(*args, **kwargs)   # plain tensors, not buffers (reference :298-302): moved by hand
        out.posDependentBias = out.posDependentBias.to(*args, **kwargs)
        return out


class URM_P_MR(URM_P):
    """URM_P + an item-relation term (env/response_model.py:305-323): mr_factor * <normalised item, sigmoid(mean of the slate's
    normalised items)>."""

    def __init__(self, maxIID, maxUID, slate_size, latent_size, device, no_user, p_bias_max, p_bias_min, mr_factor):
        super().__init__(maxIID, maxUID, slate_size, latent_size, device, no_user, p_bias_max, p_bias_min)
        self.mrFactor = mr_factor

    def _extras(self):
        return dict(pos_bias=self.posBias, pos_dep=self.posDependentBias, mr_factor=self.mrFactor)
