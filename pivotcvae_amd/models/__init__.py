"""Host-side mirror of the reference's ``models`` package for the slate-generation path.

``pivotcvae_amd.models.pivotcvae.PIVOTCVAE_MODELS`` and ``pivotcvae_amd.models.listcvae.UserListCVAEWithPrior``
are drop-ins for the reference classes of the same names (reference models/pivotcvae.py:458-461,
models/listcvae.py:8).
"""
from .cvae import BaseCVAE  # noqa: F401
from .listcvae import UserListCVAEWithPrior  # noqa: F401
from .pivotcvae import PIVOTCVAE_MODELS, PivotCVAE, UserPivotCVAE  # noqa: F401
