"""numpy restatement of the in-kernel Philox4x32-10 streams (test infrastructure).

Lets the tests rebuild, on the host, exactly the Bernoulli keep-mask / eps the HIP kernels draw, so the
masked CE can be checked against the oracle with the SAME mask (not just statistically).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint32) for x in np.broadcast_arrays(c0, c1, c2, c3))
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK32).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK32).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def keep_mask(R, N, keep_prob, seed, row_offset=0):
    """uint8 [R, N]: the Bernoulli(keep_prob) draw of catalog_ce's MASK_PHILOX mode (target column not forced)."""
    rows = (np.arange(R, dtype=np.uint64) + np.uint64(row_offset))[:, None]
    n = np.arange(N, dtype=np.uint64)[None, :]
    nb = n & ~np.uint64(3)
    c0 = (rows & MASK32).astype(np.uint32)
    c1 = (rows >> np.uint64(32)).astype(np.uint32)
    c2 = ((nb >> np.uint64(2)) & MASK32).astype(np.uint32)
    c3 = ((nb >> np.uint64(34)) & MASK32).astype(np.uint32) ^ np.uint32(0x4D41534B)
    out = philox4x32_10(c0, c1, c2, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    j = (n & np.uint64(3)).astype(np.int64)
    u = np.choose(np.broadcast_to(j, out[0].shape), out)
    th = keep_prob * 4294967296.0
    thresh = 0 if th <= 0 else (0xFFFFFFFF if th >= 4294967295.0 else int(th))
    return (u < np.uint32(thresh)).astype(np.uint8)
