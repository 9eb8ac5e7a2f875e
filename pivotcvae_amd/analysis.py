"""Offline metrics of generated slates on the device: mirror of reference analysis.py:5-30.

``get_coverage(slates, N)`` and ``get_ILS(slates, embeds)`` keep the reference signatures and return values; the work is
two small kernels of libpcvae_hip.so (an N-bit map + popcount; one wave per slate with ||sum_i e_i||^2 instead of the
[S, S] similarity matrix).  The reference's ``get_ILS`` asserts 5-slot slates (analysis.py:21); any S > 1 is accepted here.
"""
import torch

from ._hip import check, lib, ptr, require_device, stream

F32 = torch.float32


def get_coverage(slates, N):
    """item coverage: distinct generated items / N (analysis.py:5-12)"""
    require_device(slates)
    ids = slates.reshape(-1).to(torch.long).contiguous()
    bits = torch.empty((N + 31) // 32, dtype=torch.int32, device=ids.device)
    count = torch.empty((), dtype=torch.long, device=ids.device)
    check(lib().pcvae_coverage_count(ptr(ids), ids.numel(), N, ptr(bits), ptr(count), stream()), "coverage_count")
    return count.item() * 1.0 / N


def get_ILS(slates, embeds, normalize=False):
    """intra-list similarity per slate, diversity = 1 - ILS (analysis.py:14-30); ``embeds``: nn.Embedding or its weight"""
    w = embeds.weight if hasattr(embeds, "weight") else embeds
    require_device(slates, w)
    if slates.dim() != 2 or slates.shape[1] < 2:
        raise RuntimeError(f"get_ILS: slates must be [B, S >= 2], got {tuple(slates.shape)}")
    s = slates.to(torch.long).contiguous()
    w = w.detach().to(F32).contiguous()
    out = torch.empty(s.shape[0], dtype=F32, device=s.device)
    check(lib().pcvae_ils(ptr(w, F32), w.shape[0], w.shape[1], ptr(s), s.shape[0], s.shape[1], ptr(out, F32), stream()), "ils")
    return out
