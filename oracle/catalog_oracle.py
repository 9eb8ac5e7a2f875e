"""ctypes binding of oracle/catalog_oracle.c.  TEST INFRASTRUCTURE ONLY (see that file's header)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcatalog_oracle.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "catalog_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None


def scores(x, E):
    x, E = _f(x), _f(E)
    R, D = x.shape
    N = E.shape[0]
    out = np.empty((R, N), np.float32)
    lib().catalog_scores_fma(_p(x, ctypes.c_float), _p(E, ctypes.c_float), ctypes.c_int64(R), ctypes.c_int64(N),
                             ctypes.c_int(D), _p(out, ctypes.c_float))
    return out


def argmax(x, E):
    """-> (idx int64 [R], best f32 [R]); first maximal index of the k-ordered fmaf-chain scores."""
    x, E = _f(x), _f(E)
    R, D = x.shape
    N = E.shape[0]
    idx = np.empty(R, np.int64)
    best = np.empty(R, np.float32)
    lib().catalog_argmax_fma(_p(x, ctypes.c_float), _p(E, ctypes.c_float), ctypes.c_int64(R), ctypes.c_int64(N),
                             ctypes.c_int(D), _p(idx, ctypes.c_int64), _p(best, ctypes.c_float))
    return idx, best


def ce(x, E, target, keep=None, want_dx=True):
    """-> (nll f64 [R], lse f64 [R], dx f32 [R, D] or None); see catalog_ce_fma."""
    x, E = _f(x), _f(E)
    R, D = x.shape
    N = E.shape[0]
    assert D <= 512
    t = np.ascontiguousarray(target, dtype=np.int64).reshape(-1)
    k = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint8)
    nll = np.empty(R, np.float64)
    lse = np.empty(R, np.float64)
    dx = np.empty((R, D), np.float32) if want_dx else None
    lib().catalog_ce_fma(_p(x, ctypes.c_float), _p(E, ctypes.c_float), _p(t, ctypes.c_int64), _p(k, ctypes.c_uint8),
                         ctypes.c_int64(R), ctypes.c_int64(N), ctypes.c_int(D), _p(nll, ctypes.c_double),
                         _p(lse, ctypes.c_double), _p(dx, ctypes.c_float))
    return nll, lse, dx
