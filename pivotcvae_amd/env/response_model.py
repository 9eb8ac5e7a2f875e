"""The learned click model used for in-loop evaluation (reference env/response_model.py:15-87) and the uniform user
sampler (:10-13), forward-only on the HIP path.

Only what ``train_generative.py:169-195`` touches is mirrored: ``Environment`` (owner of the RAW, un-normalised
``docEmbed`` / ``userEmbed`` tables that the CVAE copies and normalises), ``UserResponseModel_MLP.forward`` and
``sample_users``.  Training this model (pretrain_env.py) and the simulators' dataset generation are out of scope.
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
            ops.gather_rows(self.userEmbed.weight, users.reshape(-1), out=x[:, S * D:])
            ops.normalize_rows_(x[:, S * D:])
        for i in range(1, self._n + 1):
            lin = getattr(self, f"mlp_{i}")
            x = ops.linear_fwd_raw(x, lin.weight, lin.bias, ACT_RELU if i < self._n else ACT_NONE)
        return x
