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


def sparse_keep_mask(R, N, keep_prob, seed, row_offset=0):
    """uint8 [R, N]: the kept set pcvae_catalog_ce_sparse enumerates (target column not forced).

    Lane l (0..63) of a row walks catalog segment [l * seg, min(N, (l + 1) * seg)), seg = ceil(N / 64), with gaps
    floor(ln U / ln(1 - p)); Philox call j of (row, lane) has counter (row_lo, row_hi, lane + 64 j, "SPAR") and yields the
    gaps 2 j (words x, y) and 2 j + 1 (words z, w); U = (2 u52 + 1) / 2^53 with u52 = x << 20 | y >> 12."""
    keep_prob = float(np.float32(keep_prob))     # the C ABI takes a float
    inv_log_q = 1.0 / np.log1p(-keep_prob)
    seg = (N + 63) // 64
    rows = np.arange(R, dtype=np.uint64) + np.uint64(row_offset)
    c0 = np.broadcast_to((rows & MASK32).astype(np.uint32)[:, None], (R, 64))
    c1 = np.broadcast_to((rows >> np.uint64(32)).astype(np.uint32)[:, None], (R, 64))
    lane = np.broadcast_to(np.arange(64, dtype=np.int64)[None, :], (R, 64))
    hi = np.minimum(N, (lane + 1) * seg)
    pos = lane * seg - 1
    done = pos + 1 >= hi
    out = np.zeros((R, N), np.uint8)
    rr = np.broadcast_to(np.arange(R)[:, None], (R, 64))

    def gaps(a, b):
        u52 = (a.astype(np.uint64) << np.uint64(20)) | (b.astype(np.uint64) >> np.uint64(12))
        U = (2 * u52 + 1).astype(np.float64) * 2.0 ** -53
        g = np.floor(np.log(U) * inv_log_q)
        return np.where(g < 4.0e18, g, 4.0e18).astype(np.int64)

    call = 0
    while not done.all():
        x, y, z, w = philox4x32_10(c0, c1, (lane + 64 * call).astype(np.uint32), np.uint32(0x53504152), seed & 0xFFFFFFFF,
                                   (seed >> 32) & 0xFFFFFFFF)
        call += 1
        for g in (gaps(x, y), gaps(z, w)):
            npos = np.where(g >= hi - pos, hi, pos + 1 + g)
            pos = np.where(done, pos, npos)
            done = done | (pos >= hi)
            live = ~done
            out[rr[live], pos[live]] = 1
    return out


def candidate_raw(R, Cn, n_items, seed, row_offset=0):
    """int64 [R, Cn]: the uniform draw of pcvae_candidate_draw before the first-hit / overwrite rule: columns 2k, 2k + 1 of row r
    are the words (x, y) and (z, w) of Philox call (row, k, "CAND") as 64-bit numbers mod n_items."""
    rows = (np.arange(R, dtype=np.uint64) + np.uint64(row_offset))[:, None]
    k = np.arange((Cn + 1) // 2, dtype=np.uint32)[None, :]
    c0 = np.broadcast_to((rows & MASK32).astype(np.uint32), (R, k.shape[1]))
    c1 = np.broadcast_to((rows >> np.uint64(32)).astype(np.uint32), (R, k.shape[1]))
    x, y, z, w = philox4x32_10(c0, c1, np.broadcast_to(k, c0.shape), np.uint32(0x43414E44), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    a = ((x.astype(np.uint64) << np.uint64(32)) | y.astype(np.uint64)) % np.uint64(n_items)
    b = ((z.astype(np.uint64) << np.uint64(32)) | w.astype(np.uint64)) % np.uint64(n_items)
    out = np.empty((R, 2 * k.shape[1]), np.int64)
    out[:, 0::2], out[:, 1::2] = a.astype(np.int64), b.astype(np.int64)
    return out[:, :Cn]


REJECT_KMAX = 512   # csrc/catalog_sample.hip: proposals before a row goes to the Gumbel-max kernel


def sample_reject(x, E, seed, row_offset=0, kmax=REJECT_KMAX, margin=2e-6):
    """pcvae_catalog_sample's rejection stage restated: proposal k of row r is Philox call (row, k, "RJCT") ->
    item n_k = (x << 32 | y) mod N, u_k = fp32((fp32(z) + 0.5) * 2^-32); the sample is n_k of the LOWEST k with
    u_k < sigmoid(<x_r, E_{n_k}>).  Scores here are fp64 (the kernel's are an fp32 fmaf chain), so a row is `safe` only when
    no proposal up to and including the accepted one has |u - sigmoid| <= margin.
    -> (idx [R] int64, -1 where all kmax proposals were rejected; k [R] accepted proposal number; safe [R] bool)"""
    x = np.asarray(x, np.float64)
    E64 = np.asarray(E, np.float64)
    R, N = x.shape[0], E64.shape[0]
    rows = np.arange(R, dtype=np.uint64) + np.uint64(row_offset)
    c0 = (rows & MASK32).astype(np.uint32)
    c1 = (rows >> np.uint64(32)).astype(np.uint32)
    idx = np.full(R, -1, np.int64)
    kacc = np.full(R, -1, np.int64)
    safe = np.ones(R, bool)
    live = np.ones(R, bool)
    for k in range(kmax):
        if not live.any():
            break
        a, b, z, _w = philox4x32_10(c0, c1, np.uint32(k), np.uint32(0x524A4354), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        n = (((a.astype(np.uint64) << np.uint64(32)) | b.astype(np.uint64)) % np.uint64(N)).astype(np.int64)
        u = ((z.astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -32)).astype(np.float64)
        sig = 1.0 / (1.0 + np.exp(-np.einsum("rd,rd->r", x, E64[n])))
        safe &= ~(live & (np.abs(u - sig) <= margin))
        acc = live & (u < sig)
        idx[acc], kacc[acc] = n[acc], k
        live &= ~acc
    return idx, kacc, safe
