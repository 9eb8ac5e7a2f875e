"""Host-side mirror of the part of the reference's ``env`` package that sits on the in-loop evaluation path."""
from .response_model import Environment, UserResponseModel_MLP, sample_users  # noqa: F401
