// K5 / K6 in exact fp32: fused full-catalog softmax cross-entropy (loss + gradient direction in one
// streaming pass) and fused catalog argmax, on v_mfma_f32_32x32x2_f32 (gfx950).
//
// Shape of the problem: rows rx[R, D] (R = batch x slate, up to ~10^5) against the whole frozen item
// table E[N, D] (N up to 10^7, D = 16..256).  The [R, N] logits matrix of the reference
// (models/pivotcvae.py:274) would be 328 GB at the north-star config and is never formed:
//
//   * a workgroup (4 waves) owns 128 rows x one contiguous range of 32-item catalog tiles;
//   * each wave keeps ITS 32 rows of rx in registers as the MFMA B operand for the whole range
//     (lane (j, h) holds rx[row j][2s + h]), and streams E tiles through double-buffered LDS;
//   * the MFMA is issued "swapped" (C[n][r] = sum_k E[n][k] rx[r][k]) so a lane holds 16 logits of
//     ONE row r: max / exp / sum are lane-local, only one cross-half max per tile;
//   * the softmax numerators, still in the accumulator layout, are directly the B operand of the
//     second MFMA chain U^T[d][r] += sum_n E[n][d] P[n][r] (flash-attention with K = V = E), so the
//     gradient direction needs no extra pass over the catalog and no transposition;
//   * catalog ranges ("splits") of one row block run as separate workgroups (fills the chip when R
//     is small, e.g. 10 240 rows per GPU under 8-way data parallel) and are combined by a tiny
//     deterministic merge kernel (log-sum-exp merge, no atomics -> bitwise reproducible).
//
// Numerics: every logit is the k-ordered fmaf chain fma(rx[2s+1] E[2s+1], fma(rx[2s] E[2s], acc))
// (that is what the f32 MFMA computes), identical to oracle/catalog_oracle.c, so argmax ids are
// bit-exact against the oracle including the lowest-index tie rule.
#include "catalog_plan.h"

using namespace pcvae;

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

enum { MASK_NONE = 0, MASK_PHILOX = 1, MASK_BYTES = 2 };

struct CatParams {
    const float* rx;      // [R, D]
    const float* E;       // [N, D]
    const int64_t* target;  // [R] (CE only)
    const uint8_t* keep;  // [R, N] or null
    uint32_t keep_thresh; // Philox: keep iff u32 < thresh
    uint64_t seed, row_offset;
    int64_t R, N;
    int nrb, nsplit, tiles_per_split, ntiles;
    float* pm;            // [nsplit][R]   running max         (argmax: best value)
    float* pl;            // [nsplit][R]   sum of exp          (argmax: unused)
    float* pU;            // [nsplit][R][D] numerator vector   (CE with dx only)
    int64_t* pn;          // [nsplit][R]   argmax index
    const uint8_t* flags; // CE only, or null: one byte per 256-row block; only blocks whose flag is 1 are computed / written
    float dx_scale;       // CE: dx is written times this (the 1 / (R W) of the mean reduction: no separate scaling launch)
    const uint64_t* row_offset_dev;   // sampling only, or null: added to row_offset (graph replay at a new stream position)
    const uint8_t* unres; // sampling only, or null: one byte per ROW; only workgroups with a flagged row run, only flagged rows are written
};

template <int D>
struct Geo {
    static constexpr int DP = (D + 31) / 32 * 32;   // table columns padded to the 32-wide U blocks
    static constexpr int KS = D / 2;                // MFMA k-steps of the logits chain
    static constexpr int NDB = DP / 32;             // 32-wide d blocks of the U accumulator
    static constexpr int LDE = DP + 1;              // LDS row stride (floats): conflict-free both ways
    static constexpr int NV = (32 * D / 4 + 255) / 256;  // float4 staging loads per thread per tile
    static constexpr int TILE_F4 = 32 * D / 4;
};

// ---- E tile staging: global (contiguous 32 x D floats) -> registers -> padded LDS rows -----------
template <int D>
__device__ __forceinline__ void stage_load(const float* __restrict__ E, int64_t N, int64_t n0, float4 (&v)[Geo<D>::NV]) {
    using G = Geo<D>;
#pragma unroll
    for (int i = 0; i < G::NV; ++i) {
        const int f = i * 256 + threadIdx.x;
        const int row = (4 * f) / D;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < G::TILE_F4 && n0 + row < N) v[i] = *reinterpret_cast<const float4*>(E + n0 * D + 4 * (int64_t)f);
    }
}

template <int D>
__device__ __forceinline__ void stage_store(float* S, const float4 (&v)[Geo<D>::NV]) {
    using G = Geo<D>;
#pragma unroll
    for (int i = 0; i < G::NV; ++i) {
        const int f = i * 256 + threadIdx.x;
        if (f < G::TILE_F4) {
            const int row = (4 * f) / D, col = (4 * f) % D;
            float* d = S + row * G::LDE + col;
            d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
        }
    }
}

__device__ __forceinline__ int nloc(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// =============================================================================================
// fused softmax-CE over a catalog range
// =============================================================================================
#ifdef CAT_STAMPS   // probe builds only: shader-clock stamps of workgroup 0 and of the LAST workgroup of the launch
__device__ unsigned long long g_cat_stamps[64];
#define CSTAMP(i) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) g_cat_stamps[(blockIdx.x ? 32 : 0) + (i)] = wall_clock64(); } while (0)
extern "C" int pcvae_cat_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cat_stamps), sizeof(g_cat_stamps)); }
#else
#define CSTAMP(i) do { } while (0)
#endif

template <int D, int MASK, bool WANT_DX>
__global__ void __launch_bounds__(256, (D <= 128 ? 2 : 1)) catalog_ce_f32_kernel(CatParams p) {
    using G = Geo<D>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Es0 = smem;
    float* Es1 = smem + 32 * G::LDE;

    CSTAMP(0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / p.nrb, rb = logical % p.nrb;
    if (p.flags && p.flags[rb >> 1] != 1) return;   // fallback mode of the bf16x3 path: this row block was done there
    const int t_beg = split * p.tiles_per_split;
    const int t_end = min(t_beg + p.tiles_per_split, p.ntiles);

    const int64_t r = (int64_t)rb * 128 + wave * 32 + li;
    const bool row_ok = r < p.R;
    const int64_t rl = row_ok ? r : p.R - 1;

    // zero the pad columns once (D < DP only); staging never writes them
    if (G::DP != D) {
        for (int i = threadIdx.x; i < 2 * 32 * (G::DP - D); i += 256) {
            const int b = i / (32 * (G::DP - D)), j = i % (32 * (G::DP - D));
            smem[b * 32 * G::LDE + (j / (G::DP - D)) * G::LDE + D + j % (G::DP - D)] = 0.f;
        }
    }

    // B operand of the logits chain: lane (row li, k-half h) holds rx[row][2s + h]
    float xb[G::KS];
#pragma unroll
    for (int s = 0; s < G::KS; ++s) xb[s] = p.rx[rl * D + 2 * s + h];

    const int64_t tgt = (MASK != MASK_NONE) ? p.target[rl] : -1;
    const uint64_t grow = p.row_offset + (uint64_t)rl;

    f32x16 U[G::NDB];
#pragma unroll
    for (int b = 0; b < G::NDB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) U[b][i] = 0.f;

    float m = -INFINITY, lsum = 0.f;

    float4 stg[G::NV];
    stage_load<D>(p.E, p.N, (int64_t)t_beg * 32, stg);
    stage_store<D>(Es0, stg);
    __syncthreads();
    CSTAMP(1);

    for (int t = t_beg; t < t_end; ++t) {
        float* Es = ((t - t_beg) & 1) ? Es1 : Es0;
        float* En = ((t - t_beg) & 1) ? Es0 : Es1;
        const int64_t n0 = (int64_t)t * 32;
        if (t + 1 < t_end) stage_load<D>(p.E, p.N, n0 + 32, stg);

        // ---- logits: C[n][r] = sum_k E[n][k] rx[r][k]
        f32x16 sacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < G::KS; ++s) {
            const float a = Es[li * G::LDE + 2 * s + h];
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xb[s], sacc, 0, 0, 0);
        }

        // ---- keep mask + downsample semantics (masked-out logit := 0), ragged last tile := -inf
        float z[16];
        bool kp[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { z[i] = sacc[i]; kp[i] = true; }
        if (MASK == MASK_PHILOX) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t nb = (uint64_t)(n0 + 8 * q + 4 * h);  // 4 consecutive items share one Philox call
                const Philox4 ph = philox4x32_10((uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)(nb >> 2),
                                                 (uint32_t)(nb >> 34) ^ 0x4D41534Bu /*"MASK"*/, (uint32_t)p.seed,
                                                 (uint32_t)(p.seed >> 32));
                const uint32_t u[4] = {ph.x, ph.y, ph.z, ph.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t n = n0 + 8 * q + 4 * h + j;
                    kp[4 * q + j] = (u[j] < p.keep_thresh) || (n == tgt);
                }
            }
        } else if (MASK == MASK_BYTES) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t n = n0 + nloc(i, h);
                kp[i] = (n == tgt) || (n < p.N && p.keep[rl * p.N + n] != 0);
            }
        }
        if (MASK != MASK_NONE) {
#pragma unroll
            for (int i = 0; i < 16; ++i) z[i] = kp[i] ? z[i] : 0.f;
        }
        if (n0 + 32 > p.N) {  // wave-uniform: only the ragged last tile
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (n0 + nloc(i, h) >= p.N) { z[i] = -INFINITY; kp[i] = false; }
        }

        // ---- online softmax; the two lane halves of a row share one running max
        float zmax = z[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) zmax = fmaxf(zmax, z[i]);
        zmax = fmaxf(zmax, __shfl_xor(zmax, 32, 64));
        const float m_new = fmaxf(m, zmax);
        if (__any(m_new > m)) {
            const float alpha = __expf(m - m_new);  // 1 where unchanged, 0 on the first tile
            lsum *= alpha;
            if (WANT_DX) {
#pragma unroll
                for (int b = 0; b < G::NDB; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) U[b][i] *= alpha;
            }
            m = m_new;
        }
        float pk[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float e = __expf(z[i] - m);
            lsum += e;
            pk[i] = (MASK == MASK_NONE || kp[i]) ? e : 0.f;  // gradient flows through kept logits only
        }
        // ---- U^T[d][r] += sum_n E[n][d] P[n][r]: accumulator register i IS the B operand of k-step i
        if (WANT_DX) {
#pragma unroll
            for (int b = 0; b < G::NDB; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float a = Es[nloc(i, h) * G::LDE + b * 32 + li];
                    U[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pk[i], U[b], 0, 0, 0);
                }
        }

        if (t + 1 < t_end) stage_store<D>(En, stg);
        __syncthreads();
    }

    CSTAMP(2);
    // ---- per-split partials
    const float ltot = lsum + __shfl_xor(lsum, 32, 64);
    if (row_ok) {
        const int64_t o = (int64_t)split * p.R + r;
        if (h == 0) { p.pm[o] = m; p.pl[o] = ltot; }
        if (WANT_DX) {
#pragma unroll
            for (int b = 0; b < G::NDB; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int d0 = b * 32 + 8 * q + 4 * h;
                    if (d0 < D)
                        *reinterpret_cast<float4*>(p.pU + o * D + d0) =
                            make_float4(U[b][4 * q], U[b][4 * q + 1], U[b][4 * q + 2], U[b][4 * q + 3]);
                }
        }
    }
    CSTAMP(3);
}

// one wave per row: merge the split partials, target logit, nll, lse, gradient direction
template <int D>
__global__ void __launch_bounds__(256) catalog_ce_merge_f32_kernel(CatParams p, float* __restrict__ nll,
                                                                   float* __restrict__ lse, float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    // wave-uniform row, and known to be: the split maxima, the target index and the k-ordered logit chain below go through the
    // scalar cache instead of 64-lane broadcast loads (see catalog_ce_merge_x3_kernel)
    const int64_t r = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (r >= p.R) return;
    if (p.flags && p.flags[r >> 8] != 1) return;
    // The ranges' partials are requested TOGETHER - (m, l) of range j by lane j (nsplit <= 64: catalog_plan), the U rows eight ranges
    // at a time - and combined in the order j = 0, 1, ..: the same sums, bit for bit, as the loops over j that paid a memory round
    // trip per range (three loops of 12 at config 2: 13.5 us for a 5 120-row merge).
    const int ns = p.nsplit;
    const float pm_l = lane < ns ? p.pm[(int64_t)lane * p.R + r] : -INFINITY;
    const float pl_l = lane < ns ? p.pl[(int64_t)lane * p.R + r] : 0.f;
    float M = pm_l;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o, 64));
    const float sc_l = lane < ns ? __expf(pm_l - M) : 0.f;
    const float term_l = pl_l * sc_l;
    float L = 0.f;
    for (int j = 0; j < ns; ++j) L += __shfl(term_l, j, 64);
    const int64_t t = p.target[r];
    const bool t_ok = t >= 0 && t < p.N;
    // target logit, same k order as the MFMA chain (lane 0 result is used; D <= 256 is cheap)
    float zt = 0.f;
    if (t_ok)
        for (int k = 0; k < D; ++k) zt = fmaf(p.E[t * D + k], p.rx[r * D + k], zt);
    const float lse_r = M + logf(L);
    if (lane == 0) {
        nll[r] = t_ok ? lse_r - zt : NAN;
        if (lse) lse[r] = lse_r;
    }
    if (dx) {
        const float invL = 1.f / L;
        for (int d0 = 0; d0 < D; d0 += 64) {   // wave-uniform trip count: the shuffles below need every lane
            const int d = d0 + lane;
            const bool d_ok = d < D;
            float u = 0.f;
            for (int j0 = 0; j0 < ns; j0 += 8) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int j = j0 + q < ns ? j0 + q : ns - 1;
                    v[q] = d_ok ? p.pU[((int64_t)j * p.R + r) * D + d] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float sc = __shfl(sc_l, j0 + q < ns ? j0 + q : 0, 64);
                    if (j0 + q < ns) u += v[q] * sc;
                }
            }
            if (d_ok) dx[r * D + d] = t_ok ? (u * invL - p.E[t * D + d]) * p.dx_scale : NAN;
        }
    }
}

// =============================================================================================
// fused argmax over a catalog range
// =============================================================================================
// SAMPLE: instead of the plain argmax, draw n ~ Categorical(sigmoid(score_n)) (the reference's
// Categorical(sigmoid(x E^T)).sample(), models/pivotcvae.py:349-351) by the Gumbel-max trick:
// argmax_n log(sigmoid(score_n)) - log(-log(u_n)), u_n from Philox keyed by (seed, row, n).
template <int D, bool SAMPLE>
__global__ void __launch_bounds__(256, (D <= 128 ? 2 : 1)) catalog_argmax_f32_kernel(CatParams p) {
    using G = Geo<D>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Es0 = smem;
    float* Es1 = smem + 32 * G::LDE;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / p.nrb, rb = logical % p.nrb;
    const int t_beg = split * p.tiles_per_split;
    const int t_end = min(t_beg + p.tiles_per_split, p.ntiles);

    const int64_t r = (int64_t)rb * 128 + wave * 32 + li;
    const bool row_ok = r < p.R;
    const int64_t rl = row_ok ? r : p.R - 1;
    if (SAMPLE && p.unres) {   // the fallback of the rejection sampler: leave at once unless one of this workgroup's rows needs it
        const int64_t rr = (int64_t)rb * 128 + (threadIdx.x & 127);
        if (!__syncthreads_or(rr < p.R && p.unres[rr] != 0)) return;
    }

    float xb[G::KS];
#pragma unroll
    for (int s = 0; s < G::KS; ++s) xb[s] = p.rx[rl * D + 2 * s + h];

    float best = -INFINITY;
    int64_t bidx = (int64_t)t_beg * 32;

    float4 stg[G::NV];
    stage_load<D>(p.E, p.N, (int64_t)t_beg * 32, stg);
    stage_store<D>(Es0, stg);
    __syncthreads();

    for (int t = t_beg; t < t_end; ++t) {
        float* Es = ((t - t_beg) & 1) ? Es1 : Es0;
        float* En = ((t - t_beg) & 1) ? Es0 : Es1;
        const int64_t n0 = (int64_t)t * 32;
        if (t + 1 < t_end) stage_load<D>(p.E, p.N, n0 + 32, stg);

        f32x16 sacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < G::KS; ++s) {
            const float a = Es[li * G::LDE + 2 * s + h];
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xb[s], sacc, 0, 0, 0);
        }
        if (SAMPLE) {
            const uint64_t grow = p.row_offset + (p.row_offset_dev ? *p.row_offset_dev : 0ull) + (uint64_t)rl;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t nb = (uint64_t)(n0 + 8 * q + 4 * h);
                const Philox4 ph = philox4x32_10((uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)(nb >> 2),
                                                 (uint32_t)(nb >> 34) ^ 0x47554D42u /*"GUMB"*/, (uint32_t)p.seed,
                                                 (uint32_t)(p.seed >> 32));
                const uint32_t u[4] = {ph.x, ph.y, ph.z, ph.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float uu = ((float)(u[j] >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0, 1)
                    const float sc = sacc[4 * q + j];
                    const float logsig = fminf(sc, 0.f) - log1pf(__expf(-fabsf(sc)));
                    sacc[4 * q + j] = logsig - __logf(-__logf(uu));
                }
            }
        }
        // registers walk n upward for a fixed lane, tiles walk n upward: strict '>' keeps the first maximum
        const bool ragged = n0 + 32 > p.N;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t n = n0 + nloc(i, h);
            const bool ok = !ragged || n < p.N;
            if (ok && sacc[i] > best) { best = sacc[i]; bidx = n; }
        }
        if (t + 1 < t_end) stage_store<D>(En, stg);
        __syncthreads();
    }

    // combine the two lane halves of a row: larger value, then lower index
    const float ov = __shfl_xor(best, 32, 64);
    const int lo = __shfl_xor((int)(bidx & 0xffffffff), 32, 64);
    const int hi = __shfl_xor((int)(bidx >> 32), 32, 64);
    const int64_t on = ((int64_t)hi << 32) | (uint32_t)lo;
    if (ov > best || (ov == best && on < bidx)) { best = ov; bidx = on; }
    if (row_ok && h == 0) {
        const int64_t o = (int64_t)split * p.R + r;
        p.pm[o] = best;
        p.pn[o] = bidx;
    }
}

__global__ void catalog_argmax_merge_kernel(CatParams p, int64_t* __restrict__ idx, float* __restrict__ bestv) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= p.R) return;
    if (p.unres && !p.unres[r]) return;   // the rejection sampler drew this row: idx[r] stands (its partials were never written)
    float b = p.pm[r];
    int64_t n = p.pn[r];
    for (int j = 1; j < p.nsplit; ++j) {  // splits are in increasing-n order: strict '>' keeps the first maximum
        const float v = p.pm[(int64_t)j * p.R + r];
        if (v > b) { b = v; n = p.pn[(int64_t)j * p.R + r]; }
    }
    idx[r] = n;
    if (bestv) bestv[r] = b;
}

template <int D>
int launch_ce(const CatParams& p, int mask_mode, bool want_dx, float* nll, float* lse, float* dx, hipStream_t st) {
    using G = Geo<D>;
    const size_t lds = 2 * 32 * G::LDE * sizeof(float);
    const dim3 grid((unsigned)(p.nrb * p.nsplit)), block(256);
#define PCVAE_CE_LAUNCH(MASKV, DXV) \
    hipLaunchKernelGGL((catalog_ce_f32_kernel<D, MASKV, DXV>), grid, block, lds, st, p)
    if (want_dx) {
        if (mask_mode == MASK_NONE) PCVAE_CE_LAUNCH(MASK_NONE, true);
        else if (mask_mode == MASK_PHILOX) PCVAE_CE_LAUNCH(MASK_PHILOX, true);
        else PCVAE_CE_LAUNCH(MASK_BYTES, true);
    } else {
        if (mask_mode == MASK_NONE) PCVAE_CE_LAUNCH(MASK_NONE, false);
        else if (mask_mode == MASK_PHILOX) PCVAE_CE_LAUNCH(MASK_PHILOX, false);
        else PCVAE_CE_LAUNCH(MASK_BYTES, false);
    }
#undef PCVAE_CE_LAUNCH
    int rc = check_launch("catalog_ce_f32");
    if (rc != PCVAE_OK) return rc;
    hipLaunchKernelGGL((catalog_ce_merge_f32_kernel<D>), dim3((unsigned)cdiv(p.R, 4)), dim3(256), 0, st, p, nll, lse,
                       want_dx ? dx : nullptr);
    return check_launch("catalog_ce_merge_f32");
}

template <int D>
int launch_argmax(const CatParams& p, bool sample, int64_t* idx, float* best, hipStream_t st) {
    using G = Geo<D>;
    const size_t lds = 2 * 32 * G::LDE * sizeof(float);
    if (sample)
        hipLaunchKernelGGL((catalog_argmax_f32_kernel<D, true>), dim3((unsigned)(p.nrb * p.nsplit)), dim3(256), lds, st, p);
    else
        hipLaunchKernelGGL((catalog_argmax_f32_kernel<D, false>), dim3((unsigned)(p.nrb * p.nsplit)), dim3(256), lds, st, p);
    int rc = check_launch("catalog_argmax_f32");
    if (rc != PCVAE_OK) return rc;
    hipLaunchKernelGGL(catalog_argmax_merge_kernel, dim3((unsigned)cdiv(p.R, 256)), dim3(256), 0, st, p, idx, best);
    return check_launch("catalog_argmax_merge");
}

}  // namespace

namespace pcvae {

int catalog_ce_f32_flagged(const float* rx, int64_t R, const float* E, int64_t N, int D, const int64_t* target, float* nll,
                           float* lse, float* dx, float dx_scale, void* ws, const uint8_t* flags, hipStream_t st) {
    const CatalogPlan pl = catalog_plan(R, N, D, PCVAE_PREC_F32);
    CatParams p{};
    p.rx = rx; p.E = E; p.target = target; p.R = R; p.N = N; p.flags = flags; p.dx_scale = dx_scale;
    p.nrb = pl.nrb; p.nsplit = pl.nsplit; p.tiles_per_split = pl.tiles_per_split; p.ntiles = pl.ntiles;
    p.pm = reinterpret_cast<float*>(ws);
    p.pl = p.pm + (int64_t)pl.nsplit * R;
    p.pU = p.pl + (int64_t)pl.nsplit * R;
    if (D == 128) return launch_ce<128>(p, MASK_NONE, dx != nullptr, nll, lse, dx, st);
    if (D == 256) return launch_ce<256>(p, MASK_NONE, dx != nullptr, nll, lse, dx, st);
    set_error("catalog_ce_f32_flagged: D=%d", D);
    return PCVAE_EINVAL;
}

int catalog_ce_f32(const float* rx, int64_t R, const float* E, int64_t N, int D, const int64_t* target,
                   float keep_prob, uint64_t seed, uint64_t row_offset, const uint8_t* keep_mask, float* nll,
                   float* lse, float* dx, float dx_scale, void* ws, hipStream_t st) {
    const CatalogPlan pl = catalog_plan(R, N, D, PCVAE_PREC_F32);
    CatParams p{};
    p.dx_scale = dx_scale;
    p.rx = rx; p.E = E; p.target = target; p.keep = keep_mask;
    p.seed = seed; p.row_offset = row_offset; p.R = R; p.N = N;
    p.nrb = pl.nrb; p.nsplit = pl.nsplit; p.tiles_per_split = pl.tiles_per_split; p.ntiles = pl.ntiles;
    p.pm = reinterpret_cast<float*>(ws);
    p.pl = p.pm + (int64_t)pl.nsplit * R;
    p.pU = p.pl + (int64_t)pl.nsplit * R;
    int mask_mode = MASK_NONE;
    if (keep_mask) mask_mode = MASK_BYTES;
    else if (keep_prob < 1.0f) {
        mask_mode = MASK_PHILOX;
        const double th = (double)keep_prob * 4294967296.0;
        p.keep_thresh = th <= 0.0 ? 0u : (th >= 4294967295.0 ? 0xffffffffu : (uint32_t)th);
    }
    switch (D) {
        case 16: return launch_ce<16>(p, mask_mode, dx != nullptr, nll, lse, dx, st);
        case 32: return launch_ce<32>(p, mask_mode, dx != nullptr, nll, lse, dx, st);
        case 64: return launch_ce<64>(p, mask_mode, dx != nullptr, nll, lse, dx, st);
        case 128: return launch_ce<128>(p, mask_mode, dx != nullptr, nll, lse, dx, st);
        case 256: return launch_ce<256>(p, mask_mode, dx != nullptr, nll, lse, dx, st);
    }
    set_error("catalog_ce: unsupported D=%d (16, 32, 64, 128, 256)", D);
    return PCVAE_EINVAL;
}

int catalog_argmax_f32(const float* x, int64_t R, const float* E, int64_t N, int D, bool sample, uint64_t seed,
                       uint64_t row_offset, int64_t* idx, float* best, void* ws, hipStream_t st, const uint8_t* unres,
                       const uint64_t* row_offset_dev) {
    const CatalogPlan pl = catalog_plan(R, N, D, PCVAE_PREC_F32);
    CatParams p{};
    p.rx = x; p.E = E; p.R = R; p.N = N; p.seed = seed; p.row_offset = row_offset; p.unres = sample ? unres : nullptr;
    p.row_offset_dev = sample ? row_offset_dev : nullptr;
    p.nrb = pl.nrb; p.nsplit = pl.nsplit; p.tiles_per_split = pl.tiles_per_split; p.ntiles = pl.ntiles;
    p.pm = reinterpret_cast<float*>(ws);
    p.pn = reinterpret_cast<int64_t*>(p.pm + (((int64_t)pl.nsplit * R + 1) & ~(int64_t)1));
    switch (D) {
        case 16: return launch_argmax<16>(p, sample, idx, best, st);
        case 32: return launch_argmax<32>(p, sample, idx, best, st);
        case 64: return launch_argmax<64>(p, sample, idx, best, st);
        case 128: return launch_argmax<128>(p, sample, idx, best, st);
        case 256: return launch_argmax<256>(p, sample, idx, best, st);
    }
    set_error("catalog_argmax: unsupported D=%d (16, 32, 64, 128, 256)", D);
    return PCVAE_EINVAL;
}

}  // namespace pcvae
