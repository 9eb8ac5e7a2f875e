"""ctypes binding of libpcvae_hip.so (the C ABI declared in include/pcvae.h).

There is NO fallback: if the library is missing or a tensor is not on a ROCm device the call raises.
PyTorch is used only for device memory (caching allocator), the current HIP stream and autograd
bookkeeping; every kernel on the hot path is ours.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PCVAE_LIB: load an alternative build of the same ABI (kernel A/B experiments, tools/bench_catalog.py)
LIB_PATH = os.environ.get("PCVAE_LIB") or os.path.join(_HERE, "lib", "libpcvae_hip.so")

ABI_VERSION = 3   # include/pcvae.h: PCVAE_ABI_VERSION
ACT_NONE, ACT_LEAKY, ACT_RELU = 0, 1, 2
PREC_F32, PREC_BF16, PREC_BF16X3, PREC_SCREENED, PREC_BF16X6 = 0, 1, 2, 3, 4
PREC_NAMES = {"f32": PREC_F32, "fp32": PREC_F32, "bf16": PREC_BF16, "bf16x3": PREC_BF16X3, "bf16x6": PREC_BF16X6}

GEMM_FWD, GEMM_DX, GEMM_DX_ACC, GEMM_DW = 0, 1, 2, 3
GEMM_X3 = 0x100   # OR-ed into a problem's kind: bf16x3 arithmetic (include/pcvae.h: PCVAE_GEMM_X3)
GEMM_X6 = 0x200   # ... bf16x6 arithmetic: fp32-exact products on the bf16 matrix cores (PCVAE_GEMM_X6)
GEMM_GROUP_MAX = 6

_c = ctypes
_P, _I, _L, _U64, _F, _SZ = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_uint64, _c.c_float, _c.c_size_t


class GemmDesc(_c.Structure):
    """pcvae_gemm_desc of include/pcvae.h (one problem of pcvae_linear_group)."""
    _fields_ = [("kind", _c.c_int32), ("act", _c.c_int32), ("a", _P), ("lda", _L), ("b", _P), ("ldb", _L), ("c", _P), ("ldc", _L),
                ("aux", _P), ("ldaux", _L), ("aux_out", _P), ("M", _L), ("N", _L), ("K", _L)]


# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/pcvae.h one to one
SIGNATURES = {
    "pcvae_abi_version": [],
    "pcvae_last_error": [],
    "pcvae_gather_rows": [_P, _L, _I, _P, _L, _I, _P, _L, _P],
    "pcvae_gather_rows_variant": [_I, _I, _L],
    "pcvae_condition": [_P, _L, _I, _I, _P, _L, _P],
    "pcvae_copy2d": [_P, _L, _P, _L, _L, _I, _P],
    "pcvae_concat": [_P, _L, _I, _P, _L, _I, _P, _L, _I, _P, _L, _I, _P, _L, _L, _P],
    "pcvae_scale_rows": [_P, _L, _P, _L, _L, _I, _P, _F, _P],
    "pcvae_normalize_rows": [_P, _L, _L, _I, _P],
    "pcvae_click_stats": [_P, _L, _I, _P, _P, _P],
    "pcvae_philox_randint": [_P, _L, _L, _U64, _U64, _P],
    "pcvae_linear_fwd": [_P, _L, _P, _L, _P, _P, _L, _L, _L, _L, _I, _P],
    "pcvae_linear_bwd_input": [_P, _L, _P, _L, _P, _L, _P, _L, _L, _L, _L, _P],
    "pcvae_linear_bwd_input_acc": [_P, _L, _P, _L, _P, _L, _P, _L, _L, _L, _L, _P],
    "pcvae_linear_bwd_weight": [_P, _L, _P, _L, _P, _L, _P, _L, _L, _L, _P],
    "pcvae_linear_group_ws_bytes": [_P, _I],
    "pcvae_linear_group": [_P, _I, _P, _SZ, _P],
    "pcvae_leaky_bwd": [_P, _L, _P, _L, _L, _I, _P],
    "pcvae_reparam_fwd": [_P, _P, _P, _U64, _U64, _P, _L, _P, _L, _I, _P],
    "pcvae_philox_normal": [_P, _L, _U64, _U64, _P],
    "pcvae_reparam_bwd": [_P, _L, _P, _P, _P, _P, _L, _I, _P],
    "pcvae_kld_fwd": [_P, _P, _P, _P, _L, _P, _P],
    "pcvae_kld_bwd": [_P, _P, _P, _P, _L, _P, _F, _P, _P, _P, _P, _P],
    "pcvae_latent_bwd": [_P, _L, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _L, _I, _P],
    "pcvae_assemble_inputs": [_P, _L, _P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _P, _L, _P, _L, _P, _L, _P, _L, _P],
    "pcvae_latent_fwd_packed": [_P, _P, _L, _P, _U64, _U64, _P, _L, _P, _P, _L, _I, _P],
    "pcvae_latent_bwd_packed": [_P, _L, _P, _P, _P, _L, _P, _F, _P, _P, _L, _L, _I, _P],
    "pcvae_downsample_dense": [_P, _L, _P, _L, _L, _F, _U64, _U64, _P, _L, _P],
    "pcvae_sum": [_P, _L, _F, _P, _P],
    "pcvae_catalog_ws_bytes": [_L, _L, _I, _I],
    "pcvae_catalog_ce_variant": [_L, _L, _I, _I],
    "pcvae_catalog_ce": [_P, _L, _P, _P, _L, _I, _I, _F, _P, _F, _U64, _U64, _P, _P, _P, _P, _P, _SZ, _P],
    "pcvae_catalog_ce_sparse": [_P, _L, _P, _L, _I, _P, _F, _U64, _U64, _P, _P, _P, _P],
    "pcvae_catalog_ce_sparse_scaled": [_P, _L, _P, _I, _L, _I, _P, _F, _U64, _U64, _P, _P, _P, _F, _P, _P],
    "pcvae_catalog_ce_scaled": [_P, _L, _P, _P, _L, _I, _I, _F, _P, _F, _U64, _U64, _P, _P, _P, _P, _F, _P, _SZ, _P],
    "pcvae_catalog_argmax": [_P, _L, _P, _P, _L, _I, _I, _F, _P, _P, _P, _SZ, _P],
    "pcvae_catalog_sample": [_P, _L, _P, _P, _L, _I, _I, _U64, _U64, _P, _P, _SZ, _P],
    "pcvae_catalog_sample_at": [_P, _L, _P, _P, _L, _I, _I, _U64, _U64, _P, _P, _P, _SZ, _P],
    "pcvae_split_bf16": [_P, _L, _P, _P, _P],
    "pcvae_split_bf16x2": [_P, _L, _I, _P, _P],
    "pcvae_split_bf16x3": [_P, _L, _I, _P, _P],
    "pcvae_urm_forward": [_P, _P, _L, _P, _P, _L, _P, _P, _P, _P, _F, _I, _L, _I, _I, _P, _P],
    "pcvae_candidate_draw": [_P, _L, _L, _I, _U64, _U64, _P, _P, _P, _P],
    "pcvae_candidate_scores": [_P, _L, _P, _L, _I, _P, _I, _P, _P],
    "pcvae_candidate_scores_bwd": [_P, _L, _P, _L, _I, _P, _I, _P, _P],
    "pcvae_dense_ce": [_P, _L, _L, _I, _P, _P, _P, _L, _P],
    "pcvae_candidate_ce": [_P, _L, _P, _I, _L, _I, _I, _P, _U64, _U64, _P, _P, _P, _P, _P, _F, _P, _P, _L, _P],
    "pcvae_set_words": [_P, _U64, _U64, _P],
    "pcvae_adam_step": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _F, _P],
    "pcvae_adam_step_l2": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _F, _F, _P],
    "pcvae_kernel_timer": [_I],
    "pcvae_kernel_timer_read": [_P, _P, _I],
    "pcvae_zero": [_P, _SZ, _P],
    "pcvae_elbo_pack": [_P, _P, _F, _P, _P],
    "pcvae_scatter_add_rows": [_P, _L, _I, _I, _P, _L, _P, _L, _P],
    "pcvae_normalize_rows_norm": [_P, _L, _L, _I, _P, _P],
    "pcvae_normalize_rows_bwd": [_P, _L, _P, _P, _L, _P, _L, _L, _I, _P],
    "pcvae_bce_sigmoid": [_P, _P, _L, _P, _P, _F, _P],
    "pcvae_relu_bwd": [_P, _L, _P, _L, _L, _I, _P],
    "pcvae_coverage_count": [_P, _L, _L, _P, _P, _P],
    "pcvae_ils": [_P, _L, _I, _P, _L, _I, _P, _P],
}
_RESTYPES = {"pcvae_last_error": _c.c_char_p, "pcvae_catalog_ws_bytes": _SZ, "pcvae_linear_group_ws_bytes": _SZ}

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    """Load libpcvae_hip.so or fail loudly (no CPU / eager fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -m pivotcvae_amd.build` "
                "(pivotcvae_amd has no CPU fallback; the HIP library is the product)")
        handle = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header/library mismatch
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, _I)
        if handle.pcvae_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libpcvae_hip.so ABI version {handle.pcvae_abi_version()} != {ABI_VERSION}: rebuild it "
                               "(python -m pivotcvae_amd.build)")
        _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().pcvae_last_error()
        raise RuntimeError(f"pcvae {what} failed ({rc}): {msg.decode() if msg else ''}")


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("pivotcvae_amd runs on a ROCm device only (tensor is on %s); there is no CPU "
                               "fallback - the reference's CPU path lives in oracle/ for tests" % t.device)


def ptr(t, dtype=None):
    """Device pointer of a tensor (None -> NULL), with dtype / layout checks done on the host."""
    if t is None:
        return None
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def row_major_ld(t):
    """Leading dimension (elements between rows) of a 2-D tensor whose rows are contiguous."""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError(f"need a 2-D tensor with unit column stride, got shape {tuple(t.shape)} stride {t.stride()}")
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])
