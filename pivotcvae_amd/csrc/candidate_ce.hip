// K9 fused: the reference's DEFAULT training mode (train_generative.py:52-57, 270-274: candidate sets unless --mask_train).
//   data_loader.py:46-58      per slot, Cn uniform ids with replacement; the slot's true item is the target - the FIRST column that
//                             holds it, else column 0 is overwritten with it and the target is 0
//   models/pivotcvae.py:265-271  candidateEmb [R, Cn, D] = docEmbed(candidates);  p = bmm(candidateEmb, rx)   -> [R, Cn]
//   train_generative.py:56    CrossEntropyLoss(p, sample_targets)   (a candidate drawn twice counts twice in the denominator)
//   autograd                  d rx_r = sum_c (softmax_c - [c == t]) E[cand[r, c]]
// The reference materialises the ids [R, Cn] (int64), the gathered rows [R, Cn, D] and p / dp [R, Cn].  Here ONE launch does all of
// it and none of those arrays exists: a wave owns a slate row, holds rx_r in registers, produces the row's Cn ids in LDS (drawn
// in-kernel from the Philox stream of pcvae_candidate_draw - keyed by (seed, GLOBAL row, column): independent of launch geometry and
// of how a batch is sharded over ranks - or read from a given [R, Cn] array, which is how recorded draws of the reference are
// replayed), gathers each candidate row of the fp32 table with 16-byte loads (a lane group of D/8 lanes per row, several rows in
// flight per lane), and folds  s = <rx_r, E_n>  into an online softmax whose accumulator U = sum_c p_c E_c rides along.
// Exact fp32 fmaf dot products; bound by the random-row gather rate of the fabric (R * Cn * 4 D bytes requested out of a table
// that is re-read ~R * Cn / N times), like catalog_ce_sparse_kernel, whose gather loop this shares.
#include "common.h"
#include <cmath>

using namespace pcvae;

namespace {

constexpr int CC_CAP = 2048;   // ids per wave and LDS batch (4 waves x 8 KB); Cn <= CC_CAP is one batch

struct CandParams {
    const float* rx;             // [R, D]
    const void* E;               // [N, D] fp32, or bf16 (the BF16 instantiations)
    const int64_t* feature;      // [R] true items              (drawn mode: cand == null)
    const int64_t* cand;         // [R, Cn] given candidate ids (given mode) or null
    const int64_t* cand_target;  // [R] target columns          (given mode)
    int64_t R, N;
    int64_t n_draw;              // drawn ids are uniform in [0, n_draw), n_draw <= N: the DATASET's id range (data_loader.py:23, :46
                                 // max_iid + 1), which is smaller than the table when the table has rows no slate uses
    int Cn;
    uint64_t seed, row_offset, magic;   // magic = floor((2^64 - 1) / n_draw): exact x % n_draw without a 64-bit division
    const uint64_t* seed_dev;    // or null: the seed is read from this device word instead (a captured graph replays with a new seed)
    float* nll;                  // [R]
    float* lse;                  // [R] or null
    float* dx;                   // [R, D] or null
    float dx_scale;
    int64_t* tgt_out;            // [R] or null: the target column the row used
};

__device__ __forceinline__ void cc_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int D, bool WANT_DX, bool BF16>
__global__ void __launch_bounds__(256) candidate_ce_kernel(CandParams p) {
    using Row = GatherRow<D, BF16>;
    constexpr int LPI = D / 8;            // lanes per item: a lane holds columns [4 j, 4 j + 4) and [D/2 + 4 j, D/2 + 4 j + 4)
    constexpr int IPS = 64 / LPI;         // items per step of a wave
    __shared__ int lst_all[4][CC_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= p.R) return;                 // wave-uniform; no block-wide barrier below
    int* lst = lst_all[wave];
    const int j = lane % LPI, grp = lane / LPI;
    const int Cn = p.Cn;

    const float4 xa = *reinterpret_cast<const float4*>(p.rx + r * D + Row::col_a(j));
    const float4 xb = *reinterpret_cast<const float4*>(p.rx + r * D + Row::col_b(j));
    const uint64_t grow = p.row_offset + (uint64_t)r;
    const uint64_t seed = p.seed_dev ? *p.seed_dev : p.seed;
    const bool drawn = p.cand == nullptr;
    const int64_t f = drawn ? p.feature[r] : -1;
    const int64_t* crow = drawn ? nullptr : p.cand + r * (int64_t)Cn;

    // columns [c0, c0 + cnt) of the row's candidate ids -> lst[0 .. cnt) (c0 even).  Drawn mode returns the lowest column of the
    // range that holds the true item (Cn if none); `store` = false only scans.  An id outside [0, N) (given mode) sets `bad`.
    bool bad = false;
    auto fill = [&](int c0, int cnt, bool store) -> int {
        int first = Cn;
        if (drawn) {
            for (int c = c0 + 2 * lane; c < c0 + cnt; c += 128) {   // a lane draws columns c, c + 1 from one Philox call
                const Philox4 ph = philox4x32_10((uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)(c >> 1), 0x43414E44u /*"CAND"*/,
                                                 (uint32_t)seed, (uint32_t)(seed >> 32));
                const int64_t v0 = (int64_t)mod_magic(((uint64_t)ph.x << 32) | ph.y, (uint64_t)p.n_draw, p.magic);
                const int64_t v1 = (int64_t)mod_magic(((uint64_t)ph.z << 32) | ph.w, (uint64_t)p.n_draw, p.magic);
                const bool two = c + 1 < c0 + cnt;
                if (store) {
                    lst[c - c0] = (int)v0;
                    if (two) lst[c - c0 + 1] = (int)v1;
                }
                if (two && v1 == f) first = min(first, c + 1);
                if (v0 == f) first = min(first, c);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
        } else {
            for (int c = c0 + lane; c < c0 + cnt; c += 64) {
                const int64_t v = crow[c];
                const bool okv = v >= 0 && v < p.N;
                bad |= !okv;
                lst[c - c0] = okv ? (int)v : 0;
            }
        }
        return first;
    };

    // ---- the target: column and item id
    int tcol;
    int64_t tid;
    bool overwrite = false, filled0 = false;
    if (drawn) {
        const int first = fill(0, min(Cn, CC_CAP), true);
        filled0 = true;
        int rest = Cn;
        if (first == Cn && Cn > CC_CAP) rest = fill(CC_CAP, Cn - CC_CAP, false);   // (scan only; those batches are redrawn below)
        const int hit = min(first, rest);
        overwrite = hit == Cn;
        tcol = overwrite ? 0 : hit;
        tid = f;
    } else {
        const int64_t t = p.cand_target[r];
        const bool ok = t >= 0 && t < Cn;
        tcol = ok ? (int)t : -1;
        tid = ok ? crow[t] : -1;
    }
    const bool t_ok = tid >= 0 && tid < p.N;

    // per lane-group online softmax stream over the candidates the group processed
    float m = -INFINITY, l = 0.f;
    float4 ua = make_float4(0.f, 0.f, 0.f, 0.f), ub = ua;
    float zt = -INFINITY;

    for (int c0 = 0; c0 < Cn; c0 += CC_CAP) {
        const int cnt = min(CC_CAP, Cn - c0);
        if (!(filled0 && c0 == 0)) fill(c0, cnt, true);
        if (overwrite && c0 == 0 && lane == 0) lst[0] = t_ok ? (int)f : 0;   // (lane 0 drew column 0 itself: same-thread ordering)
        cc_wave_sync();   // the list is read back by other lanes of this wave

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
                    if (c0 + i0 + u * IPS + grp == tcol) zt = s;
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
        cc_wave_sync();   // everyone is done reading the list before the next batch overwrites it
    }

    // ---- merge the IPS lane-group streams (butterfly over the group index)
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
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zt = fmaxf(zt, __shfl_xor(zt, o, 64));   // one lane group saw the target column
    const bool poison = __any(bad) || !t_ok;
    const float lse_r = m + logf(l);
    if (lane == 0) {
        p.nll[r] = poison ? NAN : lse_r - zt;
        if (p.lse) p.lse[r] = poison ? NAN : lse_r;
        if (p.tgt_out) p.tgt_out[r] = tcol;
    }
    if (WANT_DX && grp == 0) {
        float4 ta = make_float4(NAN, NAN, NAN, NAN), tb = ta;
        if (!poison) {
            const float w = p.dx_scale / l, q = p.dx_scale;
            float4 a, b;
            Row::load(p.E, tid, j, a, b);
            ta = make_float4(ua.x * w - a.x * q, ua.y * w - a.y * q, ua.z * w - a.z * q, ua.w * w - a.w * q);
            tb = make_float4(ub.x * w - b.x * q, ub.y * w - b.y * q, ub.z * w - b.z * q, ub.w * w - b.w * q);
        }
        *reinterpret_cast<float4*>(p.dx + r * D + Row::col_a(j)) = ta;
        *reinterpret_cast<float4*>(p.dx + r * D + Row::col_b(j)) = tb;
    }
}

template <int D>
int launch_cand(const CandParams& p, bool bf16, hipStream_t st) {
    const dim3 grid((unsigned)cdiv(p.R, 4)), block(256);
#define PCVAE_CAND(DXV, BFV) PCVAE_LAUNCH_TIMED(PCVAE_TIMER_CANDIDATE_CE, (candidate_ce_kernel<D, DXV, BFV>), grid, block, 0, st, p)
    if (p.dx) { if (bf16) PCVAE_CAND(true, true); else PCVAE_CAND(true, false); }
    else { if (bf16) PCVAE_CAND(false, true); else PCVAE_CAND(false, false); }
#undef PCVAE_CAND
    return check_launch("candidate_ce");
}

}  // namespace

extern "C" int pcvae_candidate_ce(const float* rx, int64_t R, const void* E, int prec, int64_t N, int D, int Cn, const int64_t* feature,
                                  uint64_t seed, uint64_t row_offset, const int64_t* cand, const int64_t* cand_target, float* nll,
                                  float* lse, float* dx, float dx_scale, int64_t* tgt_out, const uint64_t* seed_dev,
                                  int64_t n_items, pcvae_stream_t stream) {
    PCVAE_REQUIRE(rx && E && nll, "candidate_ce: null pointer");
    PCVAE_REQUIRE(prec == PCVAE_PREC_F32 || prec == PCVAE_PREC_BF16, "candidate_ce: precision mode %d (fp32 or bf16 table rows)", prec);
    const bool bf16 = prec == PCVAE_PREC_BF16;
    PCVAE_REQUIRE((cand != nullptr) == (cand_target != nullptr), "candidate_ce: cand and cand_target come together");
    PCVAE_REQUIRE(cand || feature, "candidate_ce: give the slots' true items (feature) or candidate sets (cand + cand_target)");
    PCVAE_REQUIRE(R >= 0 && N > 0 && N < 2147483647LL && Cn > 0, "candidate_ce: bad problem R=%lld N=%lld Cn=%d", (long long)R,
                  (long long)N, Cn);
    PCVAE_REQUIRE(((uintptr_t)rx % 16 == 0) && ((uintptr_t)E % 16 == 0) && (!dx || (uintptr_t)dx % 16 == 0),
                  "candidate_ce: rx/E/dx must be 16-byte aligned");
    PCVAE_REQUIRE(cdiv(R, 4) <= 2147483647LL, "candidate_ce: R too large");
    PCVAE_REQUIRE(n_items <= N, "candidate_ce: n_items=%lld exceeds the table's %lld rows", (long long)n_items, (long long)N);
    if (R == 0) return PCVAE_OK;
    const int64_t n_draw = n_items > 0 ? n_items : N;   // <= 0: the whole table (ABI 2's behaviour)
    CandParams p{rx, E, feature, cand, cand_target, R, N, n_draw, Cn, seed, row_offset, ~0ull / (uint64_t)n_draw, seed_dev, nll, lse, dx,
                 dx_scale, tgt_out};
    switch (D) {
        case 16: return launch_cand<16>(p, bf16, as_stream(stream));
        case 32: return launch_cand<32>(p, bf16, as_stream(stream));
        case 64: return launch_cand<64>(p, bf16, as_stream(stream));
        case 128: return launch_cand<128>(p, bf16, as_stream(stream));
        case 256: return launch_cand<256>(p, bf16, as_stream(stream));
    }
    pcvae::set_error("candidate_ce: unsupported D=%d (16, 32, 64, 128, 256)", D);
    return PCVAE_EINVAL;
}
