"""pivotcvae_amd: MI355X-native PivotCVAE slate-generation hot path (hand-written gfx950 HIP kernels
behind a C ABI, driven by thin PyTorch-ROCm host code that mirrors the reference's module surface)."""
from .models.cvae import BaseCVAE  # noqa: F401
from .models.listcvae import UserListCVAEWithPrior  # noqa: F401
from .models.pivotcvae import PIVOTCVAE_MODELS, PivotCVAE, UserPivotCVAE  # noqa: F401

__all__ = ["BaseCVAE", "UserListCVAEWithPrior", "PIVOTCVAE_MODELS", "PivotCVAE", "UserPivotCVAE"]
