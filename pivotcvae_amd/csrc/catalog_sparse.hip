// K5 for the reference's DEFAULT training mode, n_neg << N (train_generative.py:36-44,59: `downsample(pred, slates, 1000)` +
// CrossEntropyLoss): masked-out logits are the constant 0, so each of them adds exactly exp(0) = 1 to the softmax denominator and
// nothing to the gradient - only the ~n_neg + 1 KEPT items of a row need a dot product.  The dense masked kernels stream the
// whole catalog and draw a Philox number per (row, item): ~1000x the necessary work at keep_prob = 1e-3.  This kernel
//
//   1. enumerates the kept items of a row directly, by geometric gap sampling: the catalog is cut into 64 segments, lane l of
//      the row's wave walks segment l with gaps  g = floor(ln U / ln(1 - p)),  U from Philox4x32-10 keyed by (seed, GLOBAL row,
//      lane, draw) - a Bernoulli(p) process is memoryless, so restarting it at every segment boundary is still exactly
//      i.i.d. Bernoulli(p) per item, independent of launch geometry and of how a batch is sharded over ranks
//      (tests/philox_ref.py restates the stream on the host: the masked CE is checked EXACTLY against the oracle);
//   2. gathers those rows of the fp32 table (a lane group of D/8 lanes reads one 4 D-byte row with two 16-byte loads per lane,
//      several rows in flight per lane) and folds  s = <x, E_n>  into an online softmax with the closed-form
//      (N - n_kept) * exp(0) term added at the end;  dx = sum_kept p_n E_n / L - E_t  in the same pass.
//
// Exact fp32 (fmaf dot products of the fp32 table), HBM-bound: ~R * (n_neg + 1) * 4 D bytes of random row reads.
// One wave per row; the kept list is built in LDS in batches of <= CAP entries, so any keep_prob works (just slower when dense).
#include "common.h"
#include <cmath>

using namespace pcvae;

namespace {

constexpr int SP_CAP = 2048;          // kept-list entries per wave and batch (LDS: 4 waves x 8 KB)
constexpr uint32_t SP_TAG = 0x53504152u;   // "SPAR"

struct SparseParams {
    const float* rx;        // [R, D]
    const void* E;          // [N, D] fp32, or bf16 (the BF16 instantiations)
    const int64_t* target;  // [R]
    int64_t R, N;
    uint64_t seed, row_offset;
    double inv_log_q;       // 1 / log1p(-keep_prob)  (< 0), computed once on the host
    float* nll;             // [R]
    float* lse;             // [R] or null
    float* dx;              // [R, D] or null
    float dx_scale;         // dx is written times this
    const uint64_t* seed_dev;   // or null: the seed is read from this device word instead (a captured graph replays with a new seed)
};

// gap of the Bernoulli(p) process from 52 random bits: floor(ln U * inv_log_q), U = (2 u + 1) / 2^53 in (0, 1)
__device__ __forceinline__ int64_t sp_gap(uint32_t hi, uint32_t lo, double inv_log_q) {
    const uint64_t u52 = ((uint64_t)hi << 20) | (uint64_t)(lo >> 12);
    const double U = (double)(2 * u52 + 1) * 1.1102230246251565e-16;   // 2^-53
    const double g = floor(log(U) * inv_log_q);
    return g < 4.0e18 ? (int64_t)g : (int64_t)4000000000000000000LL;
}

// LDS operations of one wave execute in program order; this only keeps hipcc from moving them across the hand-over
__device__ __forceinline__ void sp_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int D, bool WANT_DX, bool BF16>
__global__ void __launch_bounds__(256) catalog_ce_sparse_kernel(SparseParams p) {
    using Row = GatherRow<D, BF16>;
    constexpr int LPI = D / 8;            // lanes per item: a lane holds columns [4 j, 4 j + 4) and [D/2 + 4 j, D/2 + 4 j + 4)
    constexpr int IPS = 64 / LPI;         // items per step of a wave
    __shared__ int lst_all[4][SP_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= p.R) return;                 // wave-uniform; no block-wide barrier below
    int* lst = lst_all[wave];
    const int j = lane % LPI, grp = lane / LPI;

    const float4 xa = *reinterpret_cast<const float4*>(p.rx + r * D + Row::col_a(j));
    const float4 xb = *reinterpret_cast<const float4*>(p.rx + r * D + Row::col_b(j));
    const int64_t tgt = p.target[r];
    const bool t_ok = tgt >= 0 && tgt < p.N;
    const uint64_t grow = p.row_offset + (uint64_t)r;
    const uint64_t seed = p.seed_dev ? *p.seed_dev : p.seed;

    // this lane's catalog segment and the state of its gap chain
    const int64_t seg = (p.N + 63) / 64;
    const int64_t seg_hi = min(p.N, (int64_t)(lane + 1) * seg);
    int64_t pos = (int64_t)lane * seg - 1;
    bool done = pos + 1 >= seg_hi;
    uint32_t call = 0;
    int64_t g_next = 0;
    bool have = false;

    // per lane-group online softmax stream over the items the group processed
    float m = -INFINITY, l = 0.f;
    float4 ua = make_float4(0.f, 0.f, 0.f, 0.f), ub = ua;
    float zt = 0.f;
    int64_t n_kept = 0;

    bool first = true;
    while (true) {
        // ---- fill the list: the target first, then the chains, one item per lane and round
        int cnt = 0;
        if (first) {
            if (t_ok) { if (lane == 0) lst[0] = (int)tgt; cnt = 1; }
            first = false;
        }
        while (cnt + 64 <= SP_CAP && __any(!done)) {
            bool emit = false;
            if (!done) {
                int64_t g;
                if (have) { g = g_next; have = false; }
                else {
                    const Philox4 ph = philox4x32_10((uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)lane + 64u * call, SP_TAG,
                                                     (uint32_t)seed, (uint32_t)(seed >> 32));
                    ++call;
                    g = sp_gap(ph.x, ph.y, p.inv_log_q);
                    g_next = sp_gap(ph.z, ph.w, p.inv_log_q);
                    have = true;
                }
                pos = (g >= seg_hi - pos) ? seg_hi : pos + 1 + g;   // (overflow-safe)
                if (pos >= seg_hi) done = true;
                else emit = pos != tgt;    // the target is in the list already
            }
            const uint64_t mask = __ballot(emit);
            if (emit) {
                const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                lst[cnt + rank] = (int)pos;
            }
            cnt += __popcll(mask);
        }
        n_kept += cnt;
        sp_wave_sync();   // the list is read back by other lanes of this wave

        // ---- gather + dot + online softmax, IPS items per step, GatherRow::UNR steps in flight
        for (int i0 = 0; i0 < cnt; i0 += IPS * Row::UNR) {
            typename Row::Raw raw[Row::UNR];
            bool ok[Row::UNR];
#pragma unroll
            for (int u = 0; u < Row::UNR; ++u) {
                const int i = i0 + u * IPS + grp;
                ok[u] = i < cnt;
                raw[u] = ok[u] ? Row::load_raw(p.E, (int64_t)lst[i], j) : Row::zero();
            }
#pragma unroll
            for (int u = 0; u < Row::UNR; ++u) {
                float4 ea_u, eb_u;
                Row::widen(raw[u], ea_u, eb_u);
                float s = ea_u.x * xa.x;
                s = fmaf(ea_u.y, xa.y, s); s = fmaf(ea_u.z, xa.z, s); s = fmaf(ea_u.w, xa.w, s);
                s = fmaf(eb_u.x, xb.x, s); s = fmaf(eb_u.y, xb.y, s); s = fmaf(eb_u.z, xb.z, s); s = fmaf(eb_u.w, xb.w, s);
#pragma unroll
                for (int o = 1; o < LPI; o <<= 1) s += __shfl_xor(s, o, 64);   // every lane of the group holds the logit
                if (ok[u]) {
                    if (t_ok && i0 == 0 && u == 0 && grp == 0 && n_kept == cnt) zt = s;   // entry 0 of the FIRST batch is the target
                    const float m_new = fmaxf(m, s);
                    const float sc = __expf(m - m_new);     // 0 for the group's first item (m = -inf)
                    const float pe = __expf(s - m_new);
                    l = l * sc + pe;
                    if (WANT_DX) {
                        ua.x = fmaf(pe, ea_u.x, ua.x * sc); ua.y = fmaf(pe, ea_u.y, ua.y * sc);
                        ua.z = fmaf(pe, ea_u.z, ua.z * sc); ua.w = fmaf(pe, ea_u.w, ua.w * sc);
                        ub.x = fmaf(pe, eb_u.x, ub.x * sc); ub.y = fmaf(pe, eb_u.y, ub.y * sc);
                        ub.z = fmaf(pe, eb_u.z, ub.z * sc); ub.w = fmaf(pe, eb_u.w, ub.w * sc);
                    }
                    m = m_new;
                }
            }
        }
        if (!__any(!done)) break;
        sp_wave_sync();   // everyone is done reading the list before the next batch overwrites it
    }

    // ---- merge the IPS lane-group streams (butterfly over the group index), then the closed-form masked-out term
#pragma unroll
    for (int o = LPI; o < 64; o <<= 1) {
        const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64);
        const float mm = fmaxf(m, m2);
        const float s1 = mm == -INFINITY ? 0.f : __expf(m - mm), s2 = mm == -INFINITY ? 0.f : __expf(m2 - mm);
        l = l * s1 + l2 * s2;
        if (WANT_DX) {
            ua.x = ua.x * s1 + __shfl_xor(ua.x, o, 64) * s2; ua.y = ua.y * s1 + __shfl_xor(ua.y, o, 64) * s2;
            ua.z = ua.z * s1 + __shfl_xor(ua.z, o, 64) * s2; ua.w = ua.w * s1 + __shfl_xor(ua.w, o, 64) * s2;
            ub.x = ub.x * s1 + __shfl_xor(ub.x, o, 64) * s2; ub.y = ub.y * s1 + __shfl_xor(ub.y, o, 64) * s2;
            ub.z = ub.z * s1 + __shfl_xor(ub.z, o, 64) * s2; ub.w = ub.w * s1 + __shfl_xor(ub.w, o, 64) * s2;
        }
        m = mm;
    }
    zt = __shfl(zt, 0, 64);
    const int64_t n_out = p.N - n_kept;                    // masked-out items: logit 0 each
    const float mf = n_out > 0 ? fmaxf(m, 0.f) : m;
    const float sk = __expf(m - mf);
    const float L = l * sk + (n_out > 0 ? (float)n_out * __expf(-mf) : 0.f);
    const float lse_r = mf + logf(L);
    if (lane == 0) {
        p.nll[r] = t_ok ? lse_r - zt : NAN;
        if (p.lse) p.lse[r] = lse_r;
    }
    if (WANT_DX && grp == 0) {
        const float w = sk / L * p.dx_scale;
        float4 ta = make_float4(NAN, NAN, NAN, NAN), tb = ta;
        if (t_ok) {
            float4 a, b;
            Row::load(p.E, tgt, j, a, b);
            const float q = p.dx_scale;
            ta = make_float4(ua.x * w - a.x * q, ua.y * w - a.y * q, ua.z * w - a.z * q, ua.w * w - a.w * q);
            tb = make_float4(ub.x * w - b.x * q, ub.y * w - b.y * q, ub.z * w - b.z * q, ub.w * w - b.w * q);
        }
        *reinterpret_cast<float4*>(p.dx + r * D + Row::col_a(j)) = ta;
        *reinterpret_cast<float4*>(p.dx + r * D + Row::col_b(j)) = tb;
    }
}

template <int D>
int launch_sparse(const SparseParams& p, bool bf16, hipStream_t st) {
    const dim3 grid((unsigned)cdiv(p.R, 4)), block(256);
#define PCVAE_SPARSE(DXV, BFV) hipLaunchKernelGGL((catalog_ce_sparse_kernel<D, DXV, BFV>), grid, block, 0, st, p)
    if (p.dx) { if (bf16) PCVAE_SPARSE(true, true); else PCVAE_SPARSE(true, false); }
    else { if (bf16) PCVAE_SPARSE(false, true); else PCVAE_SPARSE(false, false); }
#undef PCVAE_SPARSE
    return check_launch("catalog_ce_sparse");
}

}  // namespace

extern "C" int pcvae_catalog_ce_sparse(const float* rx, int64_t R, const float* E, int64_t N, int D, const int64_t* target,
                                       float keep_prob, uint64_t seed, uint64_t row_offset, float* nll, float* lse, float* dx,
                                       pcvae_stream_t stream) {
    return pcvae_catalog_ce_sparse_scaled(rx, R, E, PCVAE_PREC_F32, N, D, target, keep_prob, seed, row_offset, nll, lse, dx, 1.0f,
                                          nullptr, stream);
}

extern "C" int pcvae_catalog_ce_sparse_scaled(const float* rx, int64_t R, const void* E, int prec, int64_t N, int D,
                                              const int64_t* target,
                                              float keep_prob, uint64_t seed, uint64_t row_offset, float* nll, float* lse,
                                              float* dx, float dx_scale, const uint64_t* seed_dev, pcvae_stream_t stream) {
    PCVAE_REQUIRE(rx && E && target && nll, "catalog_ce_sparse: null pointer");
    PCVAE_REQUIRE(prec == PCVAE_PREC_F32 || prec == PCVAE_PREC_BF16, "catalog_ce_sparse: precision mode %d (fp32 or bf16 table rows)",
                  prec);
    const bool bf16 = prec == PCVAE_PREC_BF16;
    PCVAE_REQUIRE(R > 0 && N > 0 && N < 2147483647LL, "catalog_ce_sparse: bad problem R=%lld N=%lld", (long long)R, (long long)N);
    PCVAE_REQUIRE(keep_prob > 0.f && keep_prob < 1.f, "catalog_ce_sparse: keep_prob must be in (0, 1)");
    PCVAE_REQUIRE(((uintptr_t)rx % 16 == 0) && ((uintptr_t)E % 16 == 0) && (!dx || (uintptr_t)dx % 16 == 0),
                  "catalog_ce_sparse: rx/E/dx must be 16-byte aligned");
    PCVAE_REQUIRE(cdiv(R, 4) <= 2147483647LL, "catalog_ce_sparse: R too large");
    SparseParams p{rx, E, target, R, N, seed, row_offset, 1.0 / log1p(-(double)keep_prob), nll, lse, dx, dx_scale, seed_dev};
    switch (D) {
        case 16: return launch_sparse<16>(p, bf16, as_stream(stream));
        case 32: return launch_sparse<32>(p, bf16, as_stream(stream));
        case 64: return launch_sparse<64>(p, bf16, as_stream(stream));
        case 128: return launch_sparse<128>(p, bf16, as_stream(stream));
        case 256: return launch_sparse<256>(p, bf16, as_stream(stream));
    }
    pcvae::set_error("catalog_ce_sparse: unsupported D=%d (16, 32, 64, 128, 256)", D);
    return PCVAE_EINVAL;
}
