// K5 on the bf16 matrix cores: fused full-catalog softmax cross-entropy (loss + gradient direction in ONE
// streaming pass) with v_mfma_f32_32x32x16_bf16, fp32 accumulate (gfx950).
//
// Same algorithm as catalog_f32.hip (flash-attention with K = V = E, split over catalog ranges, merged by a
// deterministic log-sum-exp kernel) re-tiled for the 16x faster bf16 pipe, where the softmax VALU work and
// the LDS / L2 feed - not the MFMA - are the things to budget:
//
//   * workgroup = 8 waves = 256 rows of rx; wave w owns rows 32w..32w+31 for the whole catalog range and
//     keeps them in registers as bf16 B fragments, pre-multiplied by log2(e) so exp is a bare v_exp_f32;
//     two waves share a SIMD, so one wave's softmax VALU runs under its partner's MFMAs;
//   * the bf16 copy of E streams through LDS by global_load_lds_dwordx4 (no staging VGPRs, asynchronous); the
//     LDS image keeps 2*D-byte rows and XOR-swizzles the 16-byte chunks with ((row&3)<<2 | (row>>2)&3) on the
//     SOURCE address, which makes both the row reads (ds_read_b128, logits A operand) and the transposed reads
//     (ds_read_b64_tr_b16, E^T A operand of the gradient chain) bank-conflict free on one image
//     (tools/lds_bank_check.py; measured SQ_LDS_BANK_CONFLICT = 0);
//   * logits are produced "swapped" (C[n][r]): a lane holds 16 logits of one row, exp / sum are lane-local, and
//     the exp2 values converted pairwise to bf16 are, in place, the B operand of U^T[d][r] += E^T[d][n] P[n][r].
//
// Kernels:
//   catalog_row_bound_kernel<D>        per 256-row block: is ||rx|| * max||E|| * log2(e) <= 90 for every row?
//   catalog_ce_bf16_fast_kernel<D>     D = 64 / 128 / 256, blocks that pass: NO running max (every exp2(logit) is a normal
//                                      fp32 number, sums of 10^7 of them stay < 2^114); v_mfma_f32_16x16x32_bf16; 4-deep ring
//                                      of 16 KB LDS buffers requested three chunks ahead, counted s_waitcnt vmcnt + raw
//                                      s_barrier at the seams, all LDS offsets immediates, transposed reads through inline
//                                      asm (the builtin makes hipcc drain every in-flight LDS-DMA), row sums of the
//                                      numerators by an MFMA against ones
//   catalog_ce_bf16_kernel<D>          any norms, masks, loss-only: lazy running max (raised only when a tile exceeds it
//                                      by 2^8), 32x32x16 MFMA, 128-item double-buffered chunks
//   catalog_screen_bf16_kernel<D,PASS> exact greedy argmax at bf16 speed (screening + fp32 rescoring of the candidates)
//
// Numerics: bf16 inputs (round-to-nearest-even), fp32 accumulation, softmax statistics in fp32.  Against
// the fp32 reference the per-logit error is ~2^-9 relative per product and zero-mean, so the ELBO terms of
// a batch agree to ~1e-6 while individual gradients agree to ~1e-3 (tests/test_hip_bf16.py).
#include "catalog_plan.h"
#include <climits>
#include <cstdlib>

using namespace pcvae;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

namespace {

enum { MASK_NONE = 0, MASK_PHILOX = 1, MASK_BYTES = 2 };
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kRaiseThr = 8.0f;  // raise the running max when a tile exceeds it by more than 2^8
constexpr int BN = 128;            // catalog items per LDS chunk
constexpr int ROWS_WG = 256;

struct CatParamsB {
    const float* rx;        // [R, D] fp32
    const uint16_t* E;      // [N, D] bf16 bits
    const int64_t* target;  // [R]
    const uint8_t* keep;    // [R, N] or null
    uint32_t keep_thresh;
    uint64_t seed, row_offset;
    int64_t R, N;
    int nrb, nsplit, tiles_per_split, ntiles;  // tiles = 32-item subtiles; tiles_per_split % 4 == 0
    float* pm;              // [nsplit][R] running max, log2 domain
    float* pl;              // [nsplit][R]
    float* pU;              // [nsplit][R][D]
    const uint8_t* safe_flags;  // [nrb] or null: 1 = this row block needs the lazy-max kernel (large |rx|)
    int run_if_flag;        // this launch handles the row blocks whose flag equals this value
    float dx_scale;         // merge kernels: dx is written times this (the 1 / (R W) of the mean reduction)
    int rem_mode;           // fast kernel only: G > 0 = handle what the pipelined kernel of this shape leaves over of every range
                            // (its last tiles, see pipe_slots_of) and ADD the result into that kernel's partials (both are
                            // max-free: pm = 0, so partials simply add up).  The grid is then nrb x G: a workgroup walks the
                            // ranges g, g + G, ... of its row block and adds the sum into the partial of range g - the
                            // leftovers are a few tiles per range, far less than a workgroup's fixed cost
};

template <int D>
struct GeoB {
    static constexpr int RB = 2 * D;          // bytes per table row
    static constexpr int CPR = D / 8;         // 16-byte chunks per row
    static constexpr int KS = D / 16;         // k-steps of the logits chain
    static constexpr int NDB = D / 32;        // 32-wide d blocks of the U accumulator
    static constexpr int CHUNK_BYTES = BN * RB;
    static constexpr int PIECES = CHUNK_BYTES / 1024;  // 1 KiB global_load_lds pieces per chunk
    static constexpr int ROWS_PER_PIECE = 1024 / RB;
};

// 16-byte-chunk swizzle of the LDS image (an involution on the chunk index of one row)
template <int D>
__device__ __forceinline__ int swz_chunk(int row, int c) {
    // low field = 2-bit reversal of (row >> 2) & 3: keeps the 32x32x16 read patterns conflict-free and makes the
    // 16x16x32 ones (fast kernel) conflict-free too - tools/lds_bank_check.py models both
    if (D >= 128) return (c & ~15) | ((c & 15) ^ (((row & 3) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1)));
    return c ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));  // D == 64: two rows per 256-B bank row
}

template <int D>
__device__ __forceinline__ int lds_off(int row, int col) {  // byte offset of element (row, col) in a chunk image
    return row * GeoB<D>::RB + (swz_chunk<D>(row, col >> 3) << 4) + ((col & 7) << 1);
}

// issue the global->LDS copy of one 128-item chunk (asynchronous; completed by the next __syncthreads)
template <int D>
__device__ __forceinline__ void stage_chunk(const uint16_t* __restrict__ E, int64_t N, int64_t n0, char* buf) {
    using G = GeoB<D>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < G::PIECES / 8; ++i) {
        const int pc = wave * (G::PIECES / 8) + i;
        const int row = pc * G::ROWS_PER_PIECE + (lane * 16) / G::RB;   // LDS destination is lane-linear
        const int cdst = ((lane * 16) % G::RB) >> 4;
        const int csrc = swz_chunk<D>(row, cdst);                        // swizzle on the SOURCE address
        int64_t n = n0 + row;
        n = n < N ? n : N - 1;                                           // ragged tail: clamp, masked later
        const char* src = reinterpret_cast<const char*>(E) + n * G::RB + (csrc << 4);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(buf + pc * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ int nloc(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

template <int D, int MASK, bool WANT_DX>
__global__ void __launch_bounds__(512, 1) catalog_ce_bf16_kernel(CatParamsB p) {
    using G = GeoB<D>;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* buf0 = smem;
    char* buf1 = smem + G::CHUNK_BYTES;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / p.nrb, rb = logical % p.nrb;
    if (p.safe_flags && p.safe_flags[rb] != p.run_if_flag) return;  // the other kernel owns this row block
    const int t_beg = split * p.tiles_per_split;
    const int t_end = min(t_beg + p.tiles_per_split, p.ntiles);
    const int n_chunks = (t_end - t_beg + 3) / 4;

    const int64_t r = (int64_t)rb * ROWS_WG + wave * 32 + li;
    const bool row_ok = r < p.R;
    const int64_t rl = row_ok ? r : p.R - 1;

    stage_chunk<D>(p.E, p.N, (int64_t)t_beg * 32, buf0);

    // B operand of the logits chain: lane (row li, half h) holds bf16(rx[row][16s + 8h + j] * log2e), j = 0..7
    bf16x8 xb[G::KS];
#pragma unroll
    for (int s = 0; s < G::KS; ++s) {
        const float4 v0 = *reinterpret_cast<const float4*>(p.rx + rl * D + 16 * s + 8 * h);
        const float4 v1 = *reinterpret_cast<const float4*>(p.rx + rl * D + 16 * s + 8 * h + 4);
        xb[s][0] = (__bf16)(v0.x * kLog2e); xb[s][1] = (__bf16)(v0.y * kLog2e);
        xb[s][2] = (__bf16)(v0.z * kLog2e); xb[s][3] = (__bf16)(v0.w * kLog2e);
        xb[s][4] = (__bf16)(v1.x * kLog2e); xb[s][5] = (__bf16)(v1.y * kLog2e);
        xb[s][6] = (__bf16)(v1.z * kLog2e); xb[s][7] = (__bf16)(v1.w * kLog2e);
    }

    const int64_t tgt = (MASK != MASK_NONE) ? p.target[rl] : -1;
    const uint64_t grow = p.row_offset + (uint64_t)rl;

    f32x16 U[G::NDB];
#pragma unroll
    for (int b = 0; b < G::NDB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) U[b][i] = 0.f;
    float m_run = 0.f, lsum = 0.f;
    bool first = true;

    // per-lane pieces of the LDS addresses
    const int grp = lane >> 4, gi = lane & 15, q = gi >> 2, pp = gi & 3;

    __syncthreads();  // chunk 0 landed (the barrier drains the LDS-DMA)

    for (int c = 0; c < n_chunks; ++c) {
        const char* cur = (c & 1) ? buf1 : buf0;
        char* nxt = (c & 1) ? buf0 : buf1;
        const int t0 = t_beg + 4 * c;
        if (c + 1 < n_chunks) stage_chunk<D>(p.E, p.N, (int64_t)(t0 + 4) * 32, nxt);
        const int nsub = min(4, t_end - t0);

        for (int st = 0; st < nsub; ++st) {
            const int nb = st * 32;                      // first LDS row of this 32-item subtile
            const int64_t n0 = (int64_t)(t0 + st) * 32;  // first catalog item of this subtile

            // ---- logits (log2 domain) minus the running max: acc starts at -m
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = -m_run;
#pragma unroll
            for (int s = 0; s < G::KS; ++s) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(cur + lds_off<D>(nb + li, 16 * s + 8 * h));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xb[s], acc, 0, 0, 0);
            }

            bool kp[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) kp[i] = true;
            if (MASK == MASK_PHILOX) {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const uint64_t nbq = (uint64_t)(n0 + 8 * qq + 4 * h);
                    const Philox4 ph = philox4x32_10((uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)(nbq >> 2),
                                                     (uint32_t)(nbq >> 34) ^ 0x4D41534Bu, (uint32_t)p.seed,
                                                     (uint32_t)(p.seed >> 32));
                    const uint32_t u[4] = {ph.x, ph.y, ph.z, ph.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        kp[4 * qq + j] = (u[j] < p.keep_thresh) || ((int64_t)(n0 + 8 * qq + 4 * h + j) == tgt);
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
                for (int i = 0; i < 16; ++i) acc[i] = kp[i] ? acc[i] : -m_run;  // masked-out logit is 0
            }
            if (n0 + 32 > p.N) {  // ragged last subtile (wave-uniform)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (n0 + nloc(i, h) >= p.N) { acc[i] = -INFINITY; kp[i] = false; }
            }

            // ---- lazy running max, shared by the two lane halves of a row
            float zmax = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
            for (int i = 3; i < 15; i += 2) zmax = fmaxf(fmaxf(zmax, acc[i]), acc[i + 1]);
            zmax = fmaxf(zmax, acc[15]);
            zmax = fmaxf(zmax, __shfl_xor(zmax, 32, 64));
            if (first) {
                m_run = zmax;  // m_run was 0: acc holds the raw logits
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] -= zmax;
                first = false;
            } else if (__any(zmax > kRaiseThr)) {
                const float shift = zmax > kRaiseThr ? zmax : 0.f;
                const float alpha = exp2f(-shift);
                lsum *= alpha;
                if (WANT_DX) {
#pragma unroll
                    for (int b = 0; b < G::NDB; ++b)
#pragma unroll
                        for (int i = 0; i < 16; ++i) U[b][i] *= alpha;
                }
                m_run += shift;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] -= shift;
            }

            // ---- numerators; bf16 pairs of them are the B operand of the gradient chain
            float pk[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float e = __builtin_amdgcn_exp2f(acc[i]);
                lsum += e;
                pk[i] = (MASK == MASK_NONE || kp[i]) ? e : 0.f;
            }
            if (WANT_DX) {
                bf16x8 pb[2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; ++j) pb[ks][j] = (__bf16)pk[8 * ks + j];
                // U^T[d][r] += sum_n E[n][d] P[n][r]; A operand = E^T fragments by transposed LDS reads:
                // element j of lane (d, h) is E[16ks + 8(j>>2) + 4h + (j&3)][d]
#pragma unroll
                for (int b = 0; b < G::NDB; ++b)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int col = 32 * b + 16 * (grp & 1) + 4 * pp;
                        const int rowa = nb + 16 * ks + 4 * (grp >> 1) + q;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4*)(cur + lds_off<D>(rowa, col)));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4*)(cur + lds_off<D>(rowa + 8, col)));
                        const s16x8 a16 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        U[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a16), pb[ks], U[b], 0, 0, 0);
                    }
            }
        }
        __syncthreads();  // next chunk landed; everyone is done with `cur`
    }

    const float ltot = lsum + __shfl_xor(lsum, 32, 64);
    if (row_ok) {
        const int64_t o = (int64_t)split * p.R + r;
        if (h == 0) { p.pm[o] = m_run; p.pl[o] = ltot; }
        if (WANT_DX) {
#pragma unroll
            for (int b = 0; b < G::NDB; ++b)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int d0 = b * 32 + 8 * qq + 4 * h;
                    *reinterpret_cast<float4*>(p.pU + o * D + d0) =
                        make_float4(U[b][4 * qq], U[b][4 * qq + 1], U[b][4 * qq + 2], U[b][4 * qq + 3]);
                }
        }
    }
}

// =============================================================================================
// Row bound prologue: a row whose |logit| bound  ||rx_r|| * max_n ||E_n|| * log2(e)  is <= 90 can use raw
// exp2(logit) with NO running max at all: every term is in [2^-90, 2^90], a sum of 10^7 of them is
// < 2^114, all normal fp32 numbers.  Row blocks with a larger bound are flagged for the lazy-max kernel.
// =============================================================================================
constexpr float kFastBound = 90.0f;

template <int D>
__global__ void __launch_bounds__(256) catalog_row_bound_kernel(const float* __restrict__ rx, int64_t R, float e_max_norm,
                                                                uint8_t* __restrict__ flags) {
    __shared__ int any_unsafe;
    if (threadIdx.x == 0) any_unsafe = 0;
    __syncthreads();
    // eight lanes per row: a wave-load covers 8 rows x 128 contiguous bytes (one thread per row made every load instruction
    // touch 64 different cache lines), 32 rows of the block per pass
    static_assert(ROWS_WG % 32 == 0 && D % 32 == 0, "8 lanes x 16 bytes per row and step");
    const int part = threadIdx.x & 7, sub = threadIdx.x >> 3;
    bool unsafe = !(e_max_norm > 0.f);
    if (!unsafe)
        for (int pass = 0; pass < ROWS_WG / 32; ++pass) {
            const int64_t r = (int64_t)blockIdx.x * ROWS_WG + pass * 32 + sub;
            float ss = 0.f;
            if (r < R) {
#pragma unroll
                for (int k = 0; k < D / 32; ++k) {
                    const float4 v = *reinterpret_cast<const float4*>(rx + r * D + 4 * (part + 8 * k));
                    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                }
            }
            ss += __shfl_xor(ss, 1, 64);
            ss += __shfl_xor(ss, 2, 64);
            ss += __shfl_xor(ss, 4, 64);
            if (r < R) unsafe |= !(sqrtf(ss) * e_max_norm * kLog2e <= kFastBound);  // NaN/inf rows count as unsafe
        }
    if (unsafe) atomicOr(&any_unsafe, 1);
    __syncthreads();
    if (threadIdx.x == 0) flags[blockIdx.x] = (uint8_t)any_unsafe;
}

// =============================================================================================
// Fast path, D = 128: no running max, LDS offsets of all reads are lane base ^ constant + immediate
// (tools/lds_bank_check.py proves the decomposition), the (buffer, subtile) loops are unrolled so that
// the hot loop carries no address arithmetic, no compare and no branch besides the chunk loop itself.
// =============================================================================================

// ds_read_b64_tr_b16 through inline asm: the builtin makes hipcc wait vmcnt(0) for every in-flight
// global_load_lds before the read (it cannot prove the read does not alias the LDS-DMA write), which serialises
// the whole staging stream behind the compute.  The asm reads are invisible to hipcc's counters, so their
// completion is awaited explicitly (tr_wait) before the first consumer.
template <int OFF>
__device__ __forceinline__ s16x4 tr_read(const unsigned addr) {
    s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
__device__ __forceinline__ void tr_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------
// The fast kernels compute with v_mfma_f32_16x16x32_bf16: at equal FLOPs per cycle the chip holds a ~12 % higher
// clock on this shape than on 32x32x16 under MFMA-dense load (bare loops on random operands: 1.95 vs 1.74 PFLOP/s,
// tools/mfma_shape_probe.hip), and the kernel runs at the power-limited ceiling.
// Lane l: c = l & 15, g = l >> 4.  A wave owns 32 rows = two 16-row column tiles ct; a subtile = 32 items = two
// 16-item row tiles rt.
//   logits   acc[rt][ct] (f32x4) = sum over D/32 k-steps; lane holds logit(n = 16 rt + 4 g + reg, r = 16 ct + c).
//            Lane group g takes a fixed set of 16-byte chunks of a row over the k-steps (any bijection of chunks to
//            (k-step, lane group) is a valid k order as long as both operands use it) - chosen so that the row reads
//            are bank-conflict free.
//   softmax  16 exp2 per lane; the 8 values of column tile ct, [acc[0][ct][0..3], acc[1][ct][0..3]], are IN PLACE
//            the B operand of the gradient MFMA (k slot (g, j) <-> item 16 (j >> 2) + 4 g + (j & 3)); their row sums
//            come from one more MFMA against an all-ones A operand (no v_add_f32 at all).
//   gradient U[dt][ct] (f32x4) += E^T tile dt (16 d x 32 items, two transposed reads per lane) . P tile ct: one MFMA
//            per (d tile, column tile) covers the whole subtile (K = 32 items).
// Ring of 4 x 16 KB LDS buffers: chunk c lives in buffer c & 3 and is requested three chunks before it is consumed.
// The seam between chunks is a counted s_waitcnt vmcnt (the two younger chunks stay in flight) + a raw s_barrier; all
// LDS offsets of the reads are immediates.
// ---------------------------------------------------------------------------------------------------------------

// =============================================================================================
// One template for D = 64 / 128 / 256: 16 KB ring chunks of 8192 / D items (128 / 64 / 32); a wave stages 16 / NW
// 1 KiB pieces per chunk, so the seam is vmcnt(2 * 16 / NW) + s_barrier.
//   D = 128  rows are 256 B = one bank row: 16 chunks XOR-swizzled by ((row & 3) << 2) | bitrev2((row >> 2) & 3).
//   D = 256  rows are 512 B: the 16-chunk XOR swizzle of D = 128 applies inside each 256-B half (bank = address mod
//            256 B, so the banking of both read patterns is that of D = 128); a k-step / d tile in the upper half is
//            an immediate +256.  32 rows per wave: 128 accumulator registers (U) + 64 of rx fragments.
//   D = 64   rows are 128 B (two per bank row): swizzle ((row >> 1) & 3) << 1 | (row >> 3) & 1 on the 3-bit chunk index,
//            lane group g takes chunks 2g, 2g+1 over the two k-steps (tools/lds_bank_check.py: conflict-free).
// =============================================================================================
template <int D>
struct FastGeo {
    static constexpr int RB = 2 * D;        // bytes per table row
    static constexpr int KS = D / 32;       // k-steps of the logits chain
    static constexpr int NDT = D / 16;      // 16-wide d tiles of the gradient accumulator
    static constexpr int BNF = 8192 / D;    // items per 16 KB ring chunk
    static constexpr int SUB = BNF / 32;    // 32-item subtiles per chunk
    static constexpr int RT = 16 * RB;      // byte stride between the two 16-item row tiles of a subtile
    static constexpr int ST = 32 * RB;      // byte stride between subtiles
    static constexpr int RPP = 1024 / RB;   // table rows per 1 KiB LDS-DMA piece
    static constexpr int CPR = D / 8;       // 16-byte chunks per row
    static constexpr int KMASK = D == 64 ? 1 : 3;
    // (D = 256 does not fit this kernel: 32 rows of a wave are 128 accumulator + 64 fragment registers, more than the 256 a
    // wave owns when two share a SIMD - 181 spilled registers, 765 TFLOP/s; it runs the pipelined kernel below)
    static constexpr int NW = 8;                      // waves per workgroup
    static constexpr int ROWS = NW * 32;              // rows of rx per workgroup
    static constexpr int PPW = 16 / NW;               // 1 KiB LDS-DMA pieces per wave and ring chunk
};

// swizzle of the fast kernels' LDS image: chunk c of row `row` lives at chunk position fswz(row, c) (an involution)
template <int D>
__device__ __forceinline__ int fswz(int row, int c) {
    if (D >= 128) return (c & ~15) | ((c & 15) ^ (((row & 3) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1)));
    return c ^ ((((row >> 1) & 3) << 1) | ((row >> 3) & 1));
}
// 16-byte chunk of a row that lane group g multiplies in k-step s
template <int D>
__device__ __forceinline__ constexpr int fchunk(int s, int g) {
    return D == 64 ? 2 * g + s : (D == 128 ? 4 * g + s : 16 * (s >> 2) + 4 * g + (s & 3));
}

struct FastLane {  // lane bases of the two LDS read patterns: address = base ^ constant + immediate
    int a0, t0, g;
    bf16x8 ones;   // bf16 1.0 x 8: A operand of the row-sum MFMA (laundered through an empty asm: never rematerialised)
};

template <int D>
__device__ __forceinline__ FastLane fast_lane(int lane) {
    using G = FastGeo<D>;
    const int c = lane & 15, g = lane >> 4, q = c >> 2, pp = c & 3;
    FastLane L;
    L.g = g;
    L.a0 = c * G::RB + ((fchunk<D>(0, g) ^ fswz<D>(c, 0)) << 4);
    const int row = 4 * g + q;
    L.t0 = row * G::RB + (((pp >> 1) ^ fswz<D>(row, 0)) << 4) + (pp & 1) * 8;
    const s16x8 ones16 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    L.ones = __builtin_bit_cast(bf16x8, ones16);
    asm volatile("" : "+v"(L.ones));
    return L;
}

// DS instructions carry a 16-bit immediate offset: ring offsets beyond 48 KB move into the base register
constexpr int ds_hi(int off) { return (off / 49152) * 49152; }

template <int D, int OFF, int DT>
__device__ __forceinline__ void tr_issue(const unsigned lbase, const int t0, s16x4& lo, s16x4& hi) {
    constexpr int HI = ds_hi(OFF), LO = OFF - HI;
    const unsigned ad = lbase + (unsigned)HI + (unsigned)(t0 ^ ((DT & 7) << 5));
    lo = tr_read<LO + (DT >> 3) * 256>(ad);
    hi = tr_read<LO + (DT >> 3) * 256 + FastGeo<D>::RT>(ad);
}

// gradient chain, d tile DT: request d tile DT + 2, wait for the pieces of DT, two MFMAs
template <int D, int OFF, int DT>
__device__ __forceinline__ void grad_chain(const unsigned lbase, const int t0, s16x4 (&tl)[FastGeo<D>::NDT],
                                           s16x4 (&th)[FastGeo<D>::NDT], const bf16x8 (&pb)[2],
                                           f32x4 (&U)[FastGeo<D>::NDT][2]) {
    constexpr int NDT = FastGeo<D>::NDT;
    if constexpr (DT < NDT) {
        if constexpr (DT + 2 < NDT) {
            tr_issue<D, OFF, DT + 2>(lbase, t0, tl[DT + 2], th[DT + 2]);
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");  // pieces of d tile DT landed, two d tiles in flight
        } else if constexpr (DT + 2 == NDT) {
            asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        const s16x8 a16 = __builtin_shufflevector(tl[DT], th[DT], 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 a = __builtin_bit_cast(bf16x8, a16);
        U[DT][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb[0], U[DT][0], 0, 0, 0);
        U[DT][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb[1], U[DT][1], 0, 0, 0);
        grad_chain<D, OFF, DT + 1>(lbase, t0, tl, th, pb, U);
    }
}

// logits of one 32-item subtile: acc[rt][ct][i] = <E[n0 + 16 rt + 4 g + i], x[16 ct + c]> (bf16 operands, fp32 accumulate)
template <int D, bool CHECK_N, int OFF>
__device__ __forceinline__ void fast_logits(const char* smem, const int off, const int64_t n0, const int64_t N,
                                            const bf16x8 (&xb)[2][FastGeo<D>::KS], f32x4 (&acc)[2][2], const FastLane& L) {
    using G = FastGeo<D>;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[rt][ct][i] = 0.f;
#pragma unroll
    for (int s = 0; s < G::KS; ++s)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(
                smem + ((L.a0 ^ ((s & G::KMASK) << 4)) + off + (OFF + rt * G::RT + (s >> 2) * 256)));
            acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xb[0][s], acc[rt][0], 0, 0, 0);
            acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xb[1][s], acc[rt][1], 0, 0, 0);
        }
    if (CHECK_N) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (n0 + 16 * rt + 4 * L.g + i >= N) { acc[rt][0][i] = -INFINITY; acc[rt][1][i] = -INFINITY; }
    }
}

template <int D, bool CHECK_N, int OFF>
__device__ __forceinline__ void fast_subtile(const char* smem, const int off, const int64_t n0, const int64_t N,
                                             const bf16x8 (&xb)[2][FastGeo<D>::KS], f32x4 (&U)[FastGeo<D>::NDT][2],
                                             f32x4 (&lsum)[2], const FastLane& L) {
    using G = FastGeo<D>;
    f32x4 acc[2][2];
    fast_logits<D, CHECK_N, OFF>(smem, off, n0, N, xb, acc, L);
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)off;
    // ---- numerators.  Their row sums come from the matrix core as well: an all-ones A operand makes every row of the
    // 16 x 16 result the sum over the 32 items of the bf16 numerators (the very values the gradient chain multiplies),
    // which replaces 16 v_add_f32 per subtile - VALU issue, not the MFMA pipe, is the scarcer resource here
    bf16x8 pb[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                const float e0 = __builtin_amdgcn_exp2f(acc[rt][ct][i]);
                const float e1 = __builtin_amdgcn_exp2f(acc[rt][ct][i + 1]);
                pb[ct][4 * rt + i] = (__bf16)e0;
                pb[ct][4 * rt + i + 1] = (__bf16)e1;
            }
    lsum[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(L.ones, pb[0], lsum[0], 0, 0, 0);
    lsum[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(L.ones, pb[1], lsum[1], 0, 0, 0);
    // ---- gradient chain: E^T pieces by asm transposed reads, requested two d tiles ahead
    s16x4 tl[G::NDT], th[G::NDT];
    tr_issue<D, OFF, 0>(lbase, L.t0, tl[0], th[0]);
    tr_issue<D, OFF, 1>(lbase, L.t0, tl[1], th[1]);
    grad_chain<D, OFF, 0>(lbase, L.t0, tl, th, pb, U);
}

// all subtiles of the 16 KB ring chunk that starts at byte OFFB (+ runtime `off`)
template <int D, int OFFB>
__device__ __forceinline__ void fast_chunk(const char* smem, const int off, const int64_t nA, const int64_t N,
                                           const bf16x8 (&xb)[2][FastGeo<D>::KS], f32x4 (&U)[FastGeo<D>::NDT][2],
                                           f32x4 (&lsum)[2], const FastLane& L) {
    using G = FastGeo<D>;
    fast_subtile<D, false, OFFB>(smem, off, nA, N, xb, U, lsum, L);
    if constexpr (G::SUB >= 2) fast_subtile<D, false, OFFB + G::ST>(smem, off, nA + 32, N, xb, U, lsum, L);
    if constexpr (G::SUB >= 4) {
        fast_subtile<D, false, OFFB + 2 * G::ST>(smem, off, nA + 64, N, xb, U, lsum, L);
        fast_subtile<D, false, OFFB + 3 * G::ST>(smem, off, nA + 96, N, xb, U, lsum, L);
    }
}

// per-lane source offsets of the two 1 KiB LDS-DMA pieces a wave stages per ring chunk (swizzle on the SOURCE address)
template <int D, int NW>
__device__ __forceinline__ void fast_lane_off(int lane, int wave, int (&lane_off)[16 / NW]) {
    using G = FastGeo<D>;
#pragma unroll
    for (int i = 0; i < 16 / NW; ++i) {
        const int pc = wave * (16 / NW) + i;
        const int rip = lane / G::CPR, pos = lane % G::CPR;   // row inside the piece, chunk position inside the row
        const int row16 = (pc * G::RPP + rip) & 15;
        lane_off[i] = rip * G::RB + (fswz<D>(row16, pos) << 4);
    }
}

template <int D, int NW>
__device__ __forceinline__ void fast_stage(const uint16_t* __restrict__ E, int64_t n0, char* buf, const int wave_u,
                                           const int (&lane_off)[16 / NW]) {
    using G = FastGeo<D>;
#pragma unroll
    for (int i = 0; i < 16 / NW; ++i) {
        const int pc = wave_u * (16 / NW) + i;
        const char* base = reinterpret_cast<const char*>(E) + (n0 + pc * G::RPP) * G::RB;  // wave-uniform (SGPR pair)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(base + (uint64_t)(uint32_t)lane_off[i]),
            (__attribute__((address_space(3))) void*)(buf + pc * 1024), 16, 0, 0);
    }
}

// The same four pieces of a wave (NW = 4) as ONE asm statement for the steady-state seams of the pipelined kernels: scalar base
// + 32-bit lane offset addressing, the piece stride as the instruction's immediate (it is added to both the global and the LDS
// address), M0 written once.  hipcc's own lowering of the builtin spends a 64-bit VALU add, an M0 update and two or three SALU
// instructions on every piece - in a seam all of that sits in ONE MFMA gap.  Invisible to hipcc's vmcnt bookkeeping: the
// callers wait with counted vmcnt by hand (they already do).  (hipcc warns that m0 on a clobber list is a reserved register: it
// never keeps a value in M0 across statements - its own LDS-DMA lowering rewrites M0 in front of every use - and the clobber
// stays so that this is stated, not assumed.)
template <int D>
__device__ __forceinline__ void pipe_stage(const uint16_t* __restrict__ E, int64_t n0, unsigned lds_dst, const int wave_u,
                                           const int (&lane_off)[4]) {
    using G = FastGeo<D>;
    static_assert(G::RPP * G::RB == 1024, "a piece is 1 KiB of whole rows");
    const char* base = reinterpret_cast<const char*>(E) + (n0 + (int64_t)wave_u * 4 * G::RPP) * G::RB;
    const unsigned m0v = lds_dst + (unsigned)wave_u * 4096u;
    asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %4\n\t"
                 "global_load_lds_dwordx4 %1, %4 offset:1024\n\t"
                 "global_load_lds_dwordx4 %2, %4 offset:2048\n\t"
                 "global_load_lds_dwordx4 %3, %4 offset:3072"
                 ::"v"(lane_off[0]), "v"(lane_off[1]), "v"(lane_off[2]), "v"(lane_off[3]), "s"(base), "s"(m0v)
                 : "memory", "m0");
}

// pieces [I0, I0 + K) of a wave's four, alone (PIPE_DMA_SPREAD: the seam's refill goes out over the gradient steps behind the
// seam instead of as a burst of four at it: the burst stalled the wave's MFMA issue behind the vector-memory queue)
template <int D, int I0, int K>
__device__ __forceinline__ void pipe_stage_pieces(const uint16_t* __restrict__ E, int64_t n0, unsigned lds_dst, const int wave_u,
                                                  const int (&lane_off)[4]) {
    using G = FastGeo<D>;
    static_assert(K == 1 || K == 2, "one or two pieces per step");
    const char* base = reinterpret_cast<const char*>(E) + (n0 + (int64_t)wave_u * 4 * G::RPP) * G::RB;
    const unsigned m0v = lds_dst + (unsigned)wave_u * 4096u;
    if constexpr (K == 1)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3"
                     ::"v"(lane_off[I0]), "s"(base), "s"(m0v), "n"(I0 * 1024) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 offset:%4\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:%5"
                     ::"v"(lane_off[I0]), "v"(lane_off[I0 + 1]), "s"(base), "s"(m0v), "n"(I0 * 1024), "n"(I0 * 1024 + 1024)
                     : "memory", "m0");
}

// ragged tail: up to 128 items staged synchronously with clamped addresses (same image: row * RB, fswz)
template <int D, int NW>
__device__ __forceinline__ void fast_stage_tail(const uint16_t* __restrict__ E, int64_t N, int64_t n0, char* buf) {
    using G = FastGeo<D>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PIECES = 128 * G::RB / 1024;
#pragma unroll
    for (int i = 0; i < PIECES / NW; ++i) {
        const int pc = wave * (PIECES / NW) + i;
        const int row = pc * G::RPP + lane / G::CPR;
        const int csrc = fswz<D>(row & 15, lane % G::CPR);
        int64_t n = n0 + row;
        n = n < N ? n : N - 1;
        const char* src = reinterpret_cast<const char*>(E) + n * G::RB + (csrc << 4);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(buf + pc * 1024), 16, 0, 0);
    }
}

template <int VM>
__device__ __forceinline__ void fast_seam() {  // chunk landed for every wave: counted wait (the younger chunks stay in flight)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VM) : "memory");
}

template <int D, int CT>
__host__ __device__ inline int pipe_slots_of(int Cn);   // defined with the pipelined kernel below

template <int D>
__global__ void __launch_bounds__(FastGeo<D>::NW * 64, 1) catalog_ce_bf16_fast_kernel(CatParamsB p) {
    using G = FastGeo<D>;
    constexpr int CB = 16384;
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    // p.nrb counts 256-row blocks (the lazy-max kernel's, and the flags'); this kernel's blocks are G::ROWS rows
    const int nrb = p.nrb * (ROWS_WG / G::ROWS);
    const int split0 = logical / nrb, rb = logical % nrb;
    if ((int64_t)rb * G::ROWS >= p.R) return;
    if (p.safe_flags[rb / (ROWS_WG / G::ROWS)] != 0) return;  // large |rx| in this row block: the lazy-max kernel handles it
    const int split_step = p.rem_mode ? p.rem_mode : p.nsplit;    // normal launches: one range per workgroup
    int t_beg, t_end, n_full;   // the current range: tiles [t_beg, t_end), n_full ring chunks that exist in full
    int64_t nbase;
    auto set_range = [&](int split) {
        t_beg = split * p.tiles_per_split;
        t_end = min(t_beg + p.tiles_per_split, p.ntiles);
        if (p.rem_mode) {   // skip the tiles the pipelined kernel takes
            const int Cn = max((int)min((int64_t)((t_end - t_beg) / G::SUB), (p.N - (int64_t)t_beg * 32) / G::BNF), 0);
            t_beg += pipe_slots_of<D, 4>(Cn);
        }
        nbase = (int64_t)t_beg * 32;
        // (no per-element bound checks in the bodies of full chunks)
        n_full = max((int)min((int64_t)((t_end - t_beg) / G::SUB), (p.N - nbase) / G::BNF), 0);
    };
    set_range(split0);

    const int64_t rw = (int64_t)rb * G::ROWS + wave * 32;  // first row of this wave
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int lane_off[G::PPW];
    fast_lane_off<D, G::NW>(lane, wave, lane_off);
    auto stage_prologue = [&]() {
#pragma unroll
        for (int c0 = 0; c0 < 3; ++c0)  // chunks 0..2 in flight
            if (c0 < n_full) fast_stage<D, G::NW>(p.E, nbase + (int64_t)c0 * G::BNF, smem + c0 * CB, wave_u, lane_off);
    };
    stage_prologue();

    // B operand of the logits chain: column tile ct, k-step s: rx[row 16 ct + c][8 fchunk(s, g) .. + 7] * log2 e
    bf16x8 xb[2][G::KS];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int64_t r = rw + 16 * ct + c;
        const int64_t rl = r < p.R ? r : p.R - 1;
#pragma unroll
        for (int s = 0; s < G::KS; ++s) {
            const float4 v0 = *reinterpret_cast<const float4*>(p.rx + rl * D + 8 * fchunk<D>(s, g));
            const float4 v1 = *reinterpret_cast<const float4*>(p.rx + rl * D + 8 * fchunk<D>(s, g) + 4);
            xb[ct][s][0] = (__bf16)(v0.x * kLog2e); xb[ct][s][1] = (__bf16)(v0.y * kLog2e);
            xb[ct][s][2] = (__bf16)(v0.z * kLog2e); xb[ct][s][3] = (__bf16)(v0.w * kLog2e);
            xb[ct][s][4] = (__bf16)(v1.x * kLog2e); xb[ct][s][5] = (__bf16)(v1.y * kLog2e);
            xb[ct][s][6] = (__bf16)(v1.z * kLog2e); xb[ct][s][7] = (__bf16)(v1.w * kLog2e);
        }
    }

    f32x4 U[G::NDT][2];
#pragma unroll
    for (int dt = 0; dt < G::NDT; ++dt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int i = 0; i < 4; ++i) U[dt][ct][i] = 0.f;
    f32x4 lsum[2];  // every register of lane (c, g): sum of the numerators of row 16 ct + c
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int i = 0; i < 4; ++i) lsum[ct][i] = 0.f;
    const FastLane L = fast_lane<D>(lane);

    for (int split = split0;;) {
    int cc = 0;
    const int n_pipe = n_full >= 3 ? n_full - 2 : 0;   // chunks consumed with two younger chunks in flight
    for (; cc + 4 <= n_pipe; cc += 4) {
#define PCVAE_RING_STEP(UU)                                                                                          \
        {                                                                                                            \
            const int64_t nA = nbase + (int64_t)(cc + UU) * G::BNF;                                                  \
            fast_seam<2 * G::PPW>();                                                                                 \
            /* buffer (UU+3)&3 held chunk cc+UU-1, which every wave has finished: refill it */                       \
            if (cc + UU + 3 < n_full)                                                                                \
                fast_stage<D, G::NW>(p.E, nA + 3 * G::BNF, smem + ((UU + 3) & 3) * CB, wave_u, lane_off);                   \
            fast_chunk<D, UU * CB>(smem, 0, nA, p.N, xb, U, lsum, L);                                                \
        }
        PCVAE_RING_STEP(0)
        PCVAE_RING_STEP(1)
        PCVAE_RING_STEP(2)
        PCVAE_RING_STEP(3)
#undef PCVAE_RING_STEP
    }
    // ---- remaining full chunks: drain the ring (vmcnt(0)), runtime offsets
    for (; cc < n_full; ++cc) {
        const int64_t nA = nbase + (int64_t)cc * G::BNF;
        fast_seam<0>();
        if (cc + 3 < n_full) fast_stage<D, G::NW>(p.E, nA + 3 * G::BNF, smem + ((cc + 3) & 3) * CB, wave_u, lane_off);
        fast_chunk<D, 0>(smem, (cc & 3) * CB, nA, p.N, xb, U, lsum, L);
    }
    // ---- tail: short / ragged chunks (at most a few subtiles), staged synchronously with clamped addresses
    for (int t = t_beg + G::SUB * n_full; t < t_end; t += 4) {
        __syncthreads();
        fast_stage_tail<D, G::NW>(p.E, p.N, (int64_t)t * 32, smem);
        __syncthreads();
        const int nsub = min(4, t_end - t);
        for (int st = 0; st < nsub; ++st)
            fast_subtile<D, true, 0>(smem, st * G::ST, (int64_t)(t + st) * 32, p.N, xb, U, lsum, L);
    }
    split += split_step;
    if (split >= p.nsplit) break;
    __syncthreads();          // rest-of-range launches: the next range of this row block reuses the ring (all loads have landed)
    set_range(split);
    stage_prologue();
    }

#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const float l = lsum[ct][0];
        const int64_t r = rw + 16 * ct + c;
        if (r < p.R) {
            const int64_t o = (int64_t)split0 * p.R + r;
            if (p.rem_mode) {   // add into the partial the pipelined kernel wrote for this range (it ran before this launch)
                if (g == 0) p.pl[o] += l;
#pragma unroll
                for (int dt = 0; dt < G::NDT; ++dt) {
                    float4* q = reinterpret_cast<float4*>(p.pU + o * D + 16 * dt + 4 * g);
                    const float4 old = *q;
                    *q = make_float4(old.x + U[dt][ct][0], old.y + U[dt][ct][1], old.z + U[dt][ct][2], old.w + U[dt][ct][3]);
                }
            } else {
                if (g == 0) { p.pm[o] = 0.f; p.pl[o] = l; }
#pragma unroll
                for (int dt = 0; dt < G::NDT; ++dt)  // U[dt][ct][reg] = U^T[d = 16 dt + 4 g + reg][r]
                    *reinterpret_cast<float4*>(p.pU + o * D + 16 * dt + 4 * g) =
                        make_float4(U[dt][ct][0], U[dt][ct][1], U[dt][ct][2], U[dt][ct][3]);
            }
        }
    }
}


// =============================================================================================
// Software-pipelined form of the fast kernel: ONE wave per SIMD (4 waves x 32 rows, 512 registers each), every MFMA and
// LDS read an inline-asm statement so that the instruction order below IS the schedule.  hipcc, left to itself at this
// occupancy, parks the logits accumulators in AGPRs (a v_accvgpr_read per exponential) and reloads one A fragment
// register per k-step behind s_waitcnt lgkmcnt(0) - a fully exposed LDS latency sixteen times per subtile.
//
// Slot t of the pipeline (one 32-item subtile per slot):
//     L(t)               logits chain of subtile t: D/16 A fragments (ds_read_b128, two ahead) x 2 MFMAs each
//     G(t-1) || X(t)     gradient chain of the PREVIOUS subtile (2 + D/8 MFMAs, transposed reads two d tiles ahead) with
//                        the exponentials / bf16 packing of THIS subtile in its issue gaps (an MFMA holds the issue
//                        port for 8 of its 16 cycles, a v_exp_f32 for 8: one per MFMA fits)
//     seam(t+1)          in the middle of that chain: counted vmcnt + s_barrier, refill of the ring buffer that held
//                        chunk t-2, and the first two A fragments of L(t+1) - so no slot starts with a cold LDS read
// Ring: 6 x 16 KB, chunks requested three ahead (chunk t is read by L(t) and by G(t) one slot later).
// Hazards hipcc cannot see through inline asm are kept by construction: an accumulator is read by the VALU >= 7 MFMAs
// after the MFMA that wrote it; the bf16 numerators are consumed by MFMAs one slot after they were packed.
// =============================================================================================
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read_b128(const unsigned addr) {
    bf16x8 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// COLD = true: the statement carries its own wait states (2 before: a copy hipcc placed in front may have just written an
// operand; 16 behind: a copy placed after may read the result).  Used everywhere except the steady-state loop, whose
// generated code contains no such copies.
template <bool COLD>
__device__ __forceinline__ void mfma_v0(f32x4& acc, const bf16x8& a, const bf16x8& b) {  // acc = A . B
    if constexpr (COLD) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0\n\ts_nop 15" : "=&v"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
template <bool COLD>
__device__ __forceinline__ void mfma_v(f32x4& acc, const bf16x8& a, const bf16x8& b) {   // acc += A . B
    if constexpr (COLD) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 15" : "+v"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <bool COLD>
__device__ __forceinline__ void mfma_a(f32x4& acc, const bf16x8& a, const bf16x8& b) {   // acc (AGPR) += A . B
    if constexpr (COLD) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 15" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
template <int N>
__device__ __forceinline__ void lgkm_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

// hipcc cannot see that the asm statements are MFMAs, so it inserts no wait states around the register copies it
// makes where control flow merges (v_accvgpr_mov of U between the cold paths' register assignments): every cold path is
// fenced - 32 idle cycles let the last MFMA retire before a copy reads its result.  The steady-state loop body has no
// copies (tests/test_host_logic.py checks the generated ISA) and is fenced once per trip, at its latch.
__device__ __forceinline__ void pipe_fence() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

#ifndef PIPE_TD
#define PIPE_TD 2            // transposed reads: d tiles requested ahead (2 reads each)
#endif
#ifndef PIPE_AD
#define PIPE_AD 2            // logits chain: A fragments requested ahead (the first PIPE_AD are issued at the previous seam)
#endif
// Timing probes (tools/build_variant.sh ... -DPIPE_PROBE=<mask>; results are garbage, only the time means something): drop from the
// steady-state slots 1: the numerator VALU ops, 2: the A-fragment reads, 4: the transposed reads, 8: the seam (wait + barrier +
// refill), 16: the LDS waits, 32: the row-sum MFMAs, 64: only the refill of the seam
#ifndef PIPE_DMA_SPREAD
#define PIPE_DMA_SPREAD 0    // n > 0: the seam's four LDS-DMA pieces as two pairs, the second n gradient steps behind the seam, instead of a
                             // burst of four at it.  Measured (round 4, same box, n = 1 / 2 / 3): D = 128 config 4 +0.2 / +0.5 / +0.5 %,
                             // D = 256 -0.5 / -0.3 / 0 %, config 3 +0.5 / +0.3 / 0 % - nothing, so the burst stays.  (One piece per
                             // step: +1.4 ... +2.5 % AND three times the L2 misses.)  The bf16x6 kernel gains 1.1 % from pairs:
                             // catalog_x3.h, X3_DMA_SPREAD; profiles/r04_dma_spread.txt
#endif
#ifndef PIPE_PROBE
#define PIPE_PROBE 0
#endif

template <int D, int OFF, int I>
__device__ __forceinline__ void pipe_a_issue(const unsigned lbase, const int a0, bf16x8& a) {
    using G = FastGeo<D>;
    constexpr int s = I >> 1, rt = I & 1;
    constexpr int HI = ds_hi(OFF), LO = OFF - HI;
    a = lds_read_b128<LO + rt * G::RT + (s >> 2) * 256>(lbase + (unsigned)HI + (unsigned)(a0 ^ ((s & G::KMASK) << 4)));
}

// issue the first K A fragments of a logits chain
template <int D, int OFF, int K>
__device__ __forceinline__ void pipe_a_prologue(const unsigned lbase, const int a0, bf16x8 (&af)[2 * FastGeo<D>::KS]) {
    if constexpr (K > 0) {
        pipe_a_prologue<D, OFF, K - 1>(lbase, a0, af);
        pipe_a_issue<D, OFF, K - 1>(lbase, a0, af[K - 1]);
    }
}
template <int D, int OFF, int K>
__device__ __forceinline__ void pipe_tr_prologue(const unsigned lbase, const int t0, s16x4 (&tl)[FastGeo<D>::NDT],
                                                 s16x4 (&th)[FastGeo<D>::NDT]) {
    if constexpr (K > 0) {
        pipe_tr_prologue<D, OFF, K - 1>(lbase, t0, tl, th);
        tr_issue<D, OFF, K - 1>(lbase, t0, tl[K - 1], th[K - 1]);
    }
}

// =============================================================================================
// The pipelined kernel generalised: CT column tiles (16 rows each) per wave and SUB subtiles per ring chunk.
//   CT = 4 (64 rows per wave, D = 128 / 64): one A fragment / transposed pair feeds FOUR MFMAs - half the LDS reads, waits
//   and issue slots per MFMA of the 32-row form.  Then the exponentials no longer fit under the gradient chain alone
//   (48 VALU ops against 36 MFMAs at D = 128), so the logits accumulators are double-buffered and the numerators X(t) of
//   slot t are spread over BOTH the gradient chain G(t-1) of their own slot and the logits chain L(t+1) of the next one:
//   VALU op v of a slot sits behind MFMA position v * MPOS / VOPS of that stretch (a compile-time schedule).
//     slot t:   L(t)   || second part of X(t-1)        (acc[t&1] written, acc[(t-1)&1] read)
//               G(t-1) || first part of X(t)            with the seam / next A fragments in the middle
// =============================================================================================
template <int D, int CT>
struct PipeGeo {
    using G = FastGeo<D>;
    static constexpr int NI = 2 * G::KS;            // steps of the logits chain (k-step, row tile)
    static constexpr int ML = NI * CT;              // MFMAs of L
    static constexpr int MG = CT + G::NDT * CT;     // MFMAs of G (row sums first)
    static constexpr int P = 4 * CT;                // numerator pairs per slot: (row tile, column tile, half)
    static constexpr int VOPS = 3 * P;              // 2 exponentials + 1 packed conversion per pair
    static constexpr int MPOS = MG + ML - 2;        // MFMA positions that may carry VALU ops (the last two of L stay free:
                                                    // the packed numerators are MFMA operands right after L)
    static constexpr int ROWS = 4 * 16 * CT;        // rows per workgroup (4 waves)
    // ring: a chunk is read by L of its slots and, one slot later, by G.  With one subtile per chunk (D = 256) the chunk
    // before the current one is still in use: 6 buffers, 3 chunks requested ahead.  With 2 or 4 subtiles per chunk 4
    // buffers / 2 ahead do - 64 KB, every DS offset inside the 16-bit immediate (one base register per read pattern
    // instead of two: D = 128 sits at the 256-VGPR limit)
    static constexpr int NB = G::SUB == 4 ? 4 : 6;  // ring buffers
    static constexpr int PF = G::SUB == 4 ? 2 : 3;  // chunks requested ahead
    // requests beyond the last chunk repeat it, so that a constant number of chunks is in flight and the steady-state trips
    // (counted vmcnt) can run to the end of the range.  Not for D = 128 CT = 4: that instantiation sits at the 256-VGPR
    // limit and hipcc's allocation of its steady-state loop only stays copy-free with the simpler protocol (trips stop three
    // chunks early, nothing is requested past the end) - tools/isa_loop_check.py is the judge.
    static constexpr bool DUMMY = !(D == 128 && CT == 4);
    static constexpr int TR = NB * G::SUB;          // slots per steady-state trip
    // ---- where the numerator ops sit.  Vector issue is the scarce resource of a slot (an MFMA holds it for 8 of its 16 cycles,
    // a v_exp_f32 for 8, a conversion, an LDS read or a wait for 4-5; MI355X_MICROARCH.md, issue-cost row): the gap behind every
    // CT-th MFMA already carries the LDS reads and the wait of the next group (and the seam), so the WEIGHTED schedule leaves
    // those gaps alone and gives every other gap at most one op, spread evenly.  Where a slot has fewer such gaps than ops
    // (D = 64) the ops are spread uniformly by count over all gaps.  (-DPIPE_UNIFORM_OPS: the uniform spread everywhere, for A/B.)
    static constexpr int FREE = MPOS - MPOS / CT;
#ifdef PIPE_UNIFORM_OPS
    static constexpr bool WEIGHTED = false;
#else
    static constexpr bool WEIGHTED = CT > 1 && VOPS <= FREE;
#endif
    static constexpr int first_op(int m) {              // ops in gap m: [first_op(m), first_op(m + 1))
        if (!WEIGHTED) return (m * VOPS + MPOS - 1) / MPOS;
        int n = 0;
        for (int v = 0; v < VOPS; ++v) {
            const int f = v * FREE / VOPS;              // op v sits in the f-th free gap; CT - 1 of them per group of CT
            n += f + f / (CT - 1) < m;
        }
        return n;
    }
    // the last exponential of a row-tile-0 pair must sit behind a gradient-chain MFMA (its accumulators are not double-buffered)
    static constexpr int LAST_RT0_EXP = 3 * (2 * CT - 1);
};

// D = 128 (CT = 4) splits every catalog range: the pipelined kernel takes the fill slot and the whole steady-state trips,
// the two-waves-per-SIMD kernel, launched right behind it, takes the rest (last tiles, ragged tail) and adds its result into
// the same partial.  That keeps the fenced paths of the pipelined kernel to a fill slot and a drain - at 256 VGPRs it has no room
// for remainder loops - and lets short ranges (the 8-GPU shards) use it.  Cn = full ring chunks of the range.
template <int D, int CT>
__host__ __device__ constexpr bool pipe_splits_range() { return D == 128 && CT == 4; }
template <int D, int CT>
__host__ __device__ inline int pipe_slots_of(int Cn) {
    using PG = PipeGeo<D, CT>;
    const int k = Cn >= 4 ? (Cn - 4) / PG::NB : 0;     // trips whose seams all have two younger chunks in flight
    return k >= 1 ? 1 + k * PG::TR : 0;
}

template <int CT>
struct PipeRegs {                 // per slot parity
    f32x4 acc1[CT];               // logits accumulators of row tile 1 (their numerators are taken during the NEXT logits chain)
    unsigned w[CT][4];            // w[ct][2 rt + h]: bf16 pair (half h of row tile rt, column tile ct)
};
// The accumulators of row tile 0 (acc0[CT]) are NOT double-buffered: the op schedule takes all their exponentials during
// the gradient chain of their own slot (static_assert in PipeGeo), before the next logits chain overwrites them.

// VALU op V of a slot's numerator stream: v = 0, 1: exponentials of pair 0; then for pair j >= 1: exp, exp, conversion of
// pair j - 1 (a transcendental result needs an independent instruction before its VALU consumer); last: conversion P-1
template <int CT, int V>
__device__ __forceinline__ void pipe2_op(const f32x4 (&acc0)[CT], const PipeRegs<CT>& src, PipeRegs<CT>& dst,
                                         float (&e)[4 * CT][2]) {
    constexpr int P = 4 * CT, VOPS = 3 * P;
    if constexpr (V < 2 || (V < VOPS - 1 && (V + 1) % 3 != 2)) {            // an exponential
        constexpr int k = V < 2 ? 0 : (V + 1) / 3, which = V < 2 ? V : (V + 1) % 3;
        constexpr int rt = k / (2 * CT), ct = (k / 2) % CT, h = k & 1;
        if constexpr (rt == 0) asm volatile("v_exp_f32 %0, %1" : "=v"(e[k][which]) : "v"(acc0[ct][2 * h + which]));
        else asm volatile("v_exp_f32 %0, %1" : "=v"(e[k][which]) : "v"(src.acc1[ct][2 * h + which]));
    } else {                                                                // a packed conversion
        constexpr int k = V == VOPS - 1 ? P - 1 : (V + 1) / 3 - 1;
        constexpr int rt = k / (2 * CT), ct = (k / 2) % CT, h = k & 1;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(dst.w[ct][2 * rt + h]) : "v"(e[k][0]), "v"(e[k][1]));
    }
}
// all ops scheduled behind MFMA position M
template <int D, int CT, int M, int V = PipeGeo<D, CT>::first_op(M)>
__device__ __forceinline__ void pipe2_ops(const f32x4 (&acc0)[CT], const PipeRegs<CT>& src, PipeRegs<CT>& dst,
                                          float (&e)[4 * CT][2]) {
    using PG = PipeGeo<D, CT>;
    if constexpr (M < PG::MPOS && V < PG::first_op(M + 1) && V < PG::VOPS) {
        if constexpr (!(PIPE_PROBE & 1)) pipe2_op<CT, V>(acc0, src, dst, e);
        pipe2_ops<D, CT, M, V + 1>(acc0, src, dst, e);
    }
}
template <int CT>
__device__ __forceinline__ void pipe2_pack(const PipeRegs<CT>& r, bf16x8 (&pb)[CT]) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const u32x4 v = {r.w[ct][0], r.w[ct][1], r.w[ct][2], r.w[ct][3]};
        pb[ct] = __builtin_bit_cast(bf16x8, v);
    }
}

struct Pipe2Seam {          // all wave-uniform
    const uint16_t* E;
    bool do_seam;           // this slot ends a ring chunk: wait for the next chunk + barrier (+ request one more)
    int64_t n_stage;        // first item of the chunk to request, < 0: nothing to request
    char* stage_buf;
    unsigned stage_lds;     // the same ring buffer as an LDS byte address (steady-state seams: pipe_stage)
    unsigned next_lbase;    // LDS address (minus the immediate) of the NEXT slot's subtile
};

// phase B: logits chain of slot t (into `cur`) with the second part of the previous slot's numerators (prev -> prev.w)
template <int D, int CT, int OFF, int I, bool HAS_XPREV, bool COLD>
__device__ __forceinline__ void pipe2_logits(const unsigned lbase, const int a0, bf16x8 (&af)[2 * FastGeo<D>::KS],
                                             const bf16x8 (&xb)[CT][FastGeo<D>::KS], f32x4 (&acc0)[CT], PipeRegs<CT>& cur,
                                             PipeRegs<CT>& prev, float (&e)[4 * CT][2]) {
    using PG = PipeGeo<D, CT>;
    if constexpr (I < PG::NI) {
        if constexpr (I + PIPE_AD < PG::NI && !(!COLD && (PIPE_PROBE & 2))) pipe_a_issue<D, OFF, I + PIPE_AD>(lbase, a0, af[I + PIPE_AD]);
        if constexpr (COLD || !(PIPE_PROBE & (2 | 16))) lgkm_wait<(I + PIPE_AD < PG::NI ? PIPE_AD : PG::NI - 1 - I)>();
        constexpr int s = I >> 1, rt = I & 1;
#define PCVAE_L_MFMA(CTI)                                                                                  \
        if constexpr (CTI < CT) {                                                                          \
            f32x4& a_ = rt == 0 ? acc0[CTI] : cur.acc1[CTI];                                               \
            if constexpr (s == 0) mfma_v0<COLD>(a_, af[I], xb[CTI][s]);                                    \
            else mfma_v<COLD>(a_, af[I], xb[CTI][s]);                                                      \
            if constexpr (HAS_XPREV) pipe2_ops<D, CT, PG::MG + I * CT + CTI>(acc0, prev, prev, e);         \
        }
        PCVAE_L_MFMA(0) PCVAE_L_MFMA(1) PCVAE_L_MFMA(2) PCVAE_L_MFMA(3)
#undef PCVAE_L_MFMA
        pipe2_logits<D, CT, OFF, I + 1, HAS_XPREV, COLD>(lbase, a0, af, xb, acc0, cur, prev, e);
    }
}

// phase A: gradient chain of slot t-1 (numerators pb_prev) with the first part of slot t's numerators (cur.acc -> cur.w)
template <int D, int CT, int OFFG, int OFFL_NEXT, int DT, bool SEAM, bool HAS_G, int VM, bool COLD>
__device__ __forceinline__ void pipe2_grad(const unsigned lbase_g, const int t0, s16x4 (&tl)[FastGeo<D>::NDT],
                                           s16x4 (&th)[FastGeo<D>::NDT], const bf16x8 (&pb_prev)[CT], const f32x4 (&acc0)[CT],
                                           PipeRegs<CT>& cur, float (&e)[4 * CT][2], f32x4 (&U)[FastGeo<D>::NDT][CT],
                                           const Pipe2Seam& sm,
                                           const int wave_u, const int (&lane_off)[4], const int a0,
                                           bf16x8 (&af)[2 * FastGeo<D>::KS]) {
    using G = FastGeo<D>;
    constexpr int NDT = G::NDT, SEAM_AT = NDT / 2;
    constexpr int DMA_PS = 2;   // pieces per group of the spread refill: pairs (single pieces cost L2 hits: catalog_x3.h, X3_DMA_SPREAD)
    constexpr int DMA_STEP = PIPE_DMA_SPREAD < NDT - SEAM_AT ? PIPE_DMA_SPREAD : NDT - SEAM_AT - 1;   // steps between the two groups
    static_assert(!PIPE_DMA_SPREAD || (DMA_STEP >= 1 && SEAM_AT + DMA_STEP < NDT), "the spread refill is out before the gradient chain ends");
    if constexpr (DT < NDT) {
        if constexpr (DT == SEAM_AT) {
            if constexpr (SEAM) {
                if constexpr (COLD) {
                    pipe_fence();
                    if (sm.do_seam) {
                        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                        if (sm.n_stage >= 0) fast_stage<D, 4>(sm.E, sm.n_stage, sm.stage_buf, wave_u, lane_off);
                    }
                    pipe_fence();
                } else if constexpr (!(PIPE_PROBE & 8)) {
                    if constexpr (PIPE_PROBE & 64) asm volatile("s_barrier" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(VM) : "memory");
                    if constexpr (!(PIPE_PROBE & 64)) {
                        if constexpr (PIPE_DMA_SPREAD) {
                            if (sm.n_stage >= 0) pipe_stage_pieces<D, 0, DMA_PS>(sm.E, sm.n_stage, sm.stage_lds, wave_u, lane_off);
                        } else {
                            if (sm.n_stage >= 0) pipe_stage<D>(sm.E, sm.n_stage, sm.stage_lds, wave_u, lane_off);
                        }
                    }
                }
            }
            if constexpr (COLD || !(PIPE_PROBE & 2))
                pipe_a_prologue<D, OFFL_NEXT, PIPE_AD>(sm.next_lbase, a0, af);   // first A fragments of the next slot
        }
        // the second pair of the seam's refill, DMA_STEP steps later (out before the next seam: the counted vmcnt there sees the
        // same queue as with the burst)
        if constexpr (SEAM && !COLD && PIPE_DMA_SPREAD && DT == SEAM_AT + DMA_STEP && !(PIPE_PROBE & (8 | 64)))
            if (sm.n_stage >= 0) pipe_stage_pieces<D, 2, DMA_PS>(sm.E, sm.n_stage, sm.stage_lds, wave_u, lane_off);
        if constexpr (HAS_G) {
            constexpr int extra = (DT >= SEAM_AT && DT < SEAM_AT + PIPE_TD) ? PIPE_AD : 0;
            if constexpr (DT + PIPE_TD < NDT && !(!COLD && (PIPE_PROBE & 4)))
                tr_issue<D, OFFG, DT + PIPE_TD>(lbase_g, t0, tl[DT + PIPE_TD], th[DT + PIPE_TD]);
            if constexpr (COLD || !(PIPE_PROBE & (2 | 4 | 16)))
                lgkm_wait<2 * ((DT + PIPE_TD < NDT ? DT + PIPE_TD : NDT - 1) - DT) + extra>();
        }
        const s16x8 a16 = __builtin_shufflevector(tl[DT], th[DT], 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 a = __builtin_bit_cast(bf16x8, a16);
#define PCVAE_G_MFMA(CTI)                                                                                  \
        if constexpr (CTI < CT) {                                                                          \
            if constexpr (HAS_G) mfma_a<COLD>(U[DT][CTI], a, pb_prev[CTI]);                                \
            pipe2_ops<D, CT, CT + DT * CT + CTI>(acc0, cur, cur, e);                                       \
        }
        PCVAE_G_MFMA(0) PCVAE_G_MFMA(1) PCVAE_G_MFMA(2) PCVAE_G_MFMA(3)
#undef PCVAE_G_MFMA
        pipe2_grad<D, CT, OFFG, OFFL_NEXT, DT + 1, SEAM, HAS_G, VM, COLD>(lbase_g, t0, tl, th, pb_prev, acc0, cur, e, U, sm,
                                                                           wave_u, lane_off, a0, af);
    } else if constexpr (COLD) {
        pipe_fence();
    }
}

template <int D, int CT, int OFFL, int OFFG, int OFFL_NEXT, bool SEAM, bool HAS_G, int VM, bool COLD>
__device__ __forceinline__ void pipe2_slot(const unsigned lbase_l, const unsigned lbase_g, const FastLane& L,
                                           const bf16x8 (&xb)[CT][FastGeo<D>::KS], bf16x8 (&af)[2 * FastGeo<D>::KS],
                                           f32x4 (&acc0)[CT], PipeRegs<CT>& cur, PipeRegs<CT>& prev,
                                           f32x4 (&U)[FastGeo<D>::NDT][CT], f32x4 (&lsum)[CT], const Pipe2Seam& sm,
                                           const int wave_u, const int (&lane_off)[4], float (&e)[4 * CT][2]) {
    using G = FastGeo<D>;
    if constexpr (COLD) pipe_fence();
    pipe2_logits<D, CT, OFFL, 0, HAS_G, COLD>(lbase_l, L.a0, af, xb, acc0, cur, prev, e);
    bf16x8 pb_prev[CT];
    pipe2_pack<CT>(prev, pb_prev);
    s16x4 tl[G::NDT], th[G::NDT];
    if constexpr (HAS_G) {
        if constexpr (COLD || !(PIPE_PROBE & 4)) pipe_tr_prologue<D, OFFG, PIPE_TD>(lbase_g, L.t0, tl, th);
#define PCVAE_ONES(CTI)                                                                                    \
        if constexpr (CTI < CT) {                                                                          \
            if constexpr (COLD) mfma_a<true>(lsum[CTI], L.ones, pb_prev[CTI]);                             \
            else if constexpr (!(PIPE_PROBE & 32)) mfma_a<false>(lsum[CTI], L.ones, pb_prev[CTI]);         \
            pipe2_ops<D, CT, CTI>(acc0, cur, cur, e);                                                      \
        }
        PCVAE_ONES(0) PCVAE_ONES(1) PCVAE_ONES(2) PCVAE_ONES(3)
#undef PCVAE_ONES
    } else {
        pipe_fence();  // no MFMAs between the logits chain and the first exponential
        pipe2_ops<D, CT, 0>(acc0, cur, cur, e);
        if constexpr (CT > 1) pipe2_ops<D, CT, 1>(acc0, cur, cur, e);
        if constexpr (CT > 2) { pipe2_ops<D, CT, 2>(acc0, cur, cur, e); pipe2_ops<D, CT, 3>(acc0, cur, cur, e); }
    }
    pipe2_grad<D, CT, OFFG, OFFL_NEXT, 0, SEAM, HAS_G, VM, COLD>(lbase_g, L.t0, tl, th, pb_prev, acc0, cur, e, U, sm, wave_u,
                                                                 lane_off, L.a0, af);
}

// pipeline drain after the last slot: the rest of its numerators, then its gradient chain
template <int D, int CT, int M = PipeGeo<D, CT>::MG>
__device__ __forceinline__ void pipe2_drain_ops(const f32x4 (&acc0)[CT], PipeRegs<CT>& last, float (&e)[4 * CT][2]) {
    if constexpr (M < PipeGeo<D, CT>::MPOS) {
        pipe2_ops<D, CT, M>(acc0, last, last, e);
        pipe2_drain_ops<D, CT, M + 1>(acc0, last, e);
    }
}
template <int D, int CT, int DT = 0>
__device__ __forceinline__ void pipe2_cold_grad(const unsigned lbase_g, const int t0, s16x4 (&tl)[FastGeo<D>::NDT],
                                                s16x4 (&th)[FastGeo<D>::NDT], const bf16x8 (&pb)[CT],
                                                f32x4 (&U)[FastGeo<D>::NDT][CT]) {
    constexpr int NDT = FastGeo<D>::NDT;
    if constexpr (DT < NDT) {
        if constexpr (DT + 2 < NDT) tr_issue<D, 0, DT + 2>(lbase_g, t0, tl[DT + 2], th[DT + 2]);
        lgkm_wait<2 * ((DT + 2 < NDT ? DT + 2 : NDT - 1) - DT)>();
        const s16x8 a16 = __builtin_shufflevector(tl[DT], th[DT], 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 a = __builtin_bit_cast(bf16x8, a16);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) mfma_a<true>(U[DT][ct], a, pb[ct]);
        pipe2_cold_grad<D, CT, DT + 1>(lbase_g, t0, tl, th, pb, U);
    }
}
template <int D, int CT>
__device__ __forceinline__ void pipe2_cold_gradient(const unsigned lbase_g, const FastLane& L, const bf16x8 (&pb)[CT],
                                                    f32x4 (&U)[FastGeo<D>::NDT][CT], f32x4 (&lsum)[CT]) {
    using G = FastGeo<D>;
    s16x4 tl[G::NDT], th[G::NDT];
    pipe_fence();
    tr_issue<D, 0, 0>(lbase_g, L.t0, tl[0], th[0]);
    tr_issue<D, 0, 1>(lbase_g, L.t0, tl[1], th[1]);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) mfma_a<true>(lsum[ct], L.ones, pb[ct]);
    pipe2_cold_grad<D, CT>(lbase_g, L.t0, tl, th, pb, U);
    pipe_fence();
}

// one subtile on its own (ragged tail): logits, bound check, numerators, gradient - nothing overlapped, every MFMA fenced
template <int D, int CT, int I = 0>
__device__ __forceinline__ void pipe2_cold_logits(const unsigned lbase, const int a0, bf16x8 (&af)[2 * FastGeo<D>::KS],
                                                  const bf16x8 (&xb)[CT][FastGeo<D>::KS], f32x4 (&acc)[2][CT]) {
    constexpr int NI = 2 * FastGeo<D>::KS;
    if constexpr (I < NI) {
        pipe_a_issue<D, 0, I>(lbase, a0, af[I]);
        lgkm_wait<0>();
        constexpr int s = I >> 1, rt = I & 1;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            if constexpr (s == 0) mfma_v0<true>(acc[rt][ct], af[I], xb[ct][s]);
            else mfma_v<true>(acc[rt][ct], af[I], xb[ct][s]);
        }
        pipe2_cold_logits<D, CT, I + 1>(lbase, a0, af, xb, acc);
    }
}
template <int D, int CT>
__device__ __forceinline__ void pipe2_solo(const unsigned lbase, const int64_t n0, const int64_t N, const FastLane& L,
                                           const bf16x8 (&xb)[CT][FastGeo<D>::KS], f32x4 (&U)[FastGeo<D>::NDT][CT],
                                           f32x4 (&lsum)[CT]) {
    f32x4 acc[2][CT];
    bf16x8 af[2 * FastGeo<D>::KS];
    pipe_fence();
    pipe2_cold_logits<D, CT>(lbase, L.a0, af, xb, acc);
    pipe_fence();
    bf16x8 pb[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                const bool ok0 = n0 + 16 * rt + 4 * L.g + i < N, ok1 = n0 + 16 * rt + 4 * L.g + i + 1 < N;
                const float e0 = ok0 ? __builtin_amdgcn_exp2f(acc[rt][ct][i]) : 0.f;
                const float e1 = ok1 ? __builtin_amdgcn_exp2f(acc[rt][ct][i + 1]) : 0.f;
                pb[ct][4 * rt + i] = (__bf16)e0;
                pb[ct][4 * rt + i + 1] = (__bf16)e1;
            }
    pipe2_cold_gradient<D, CT>(lbase, L, pb, U, lsum);
}

template <int D, int CT>
__global__ void __launch_bounds__(256, 1) catalog_ce_bf16_pipe_kernel(CatParamsB p) {
    using G = FastGeo<D>;
    using PG = PipeGeo<D, CT>;
    constexpr int CB = 16384, NW = 4, ROWS = PG::ROWS, SUB = G::SUB, TR = PG::TR;
    static_assert(TR % 2 == 0, "the accumulator parity must repeat every trip");
    static_assert(PG::LAST_RT0_EXP < PG::first_op(PG::MG), "row tile 0 numerators must finish in their own slot");
    static_assert(PG::first_op(PG::MPOS) == PG::VOPS, "every numerator op has a gap");
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int nrb = (int)((p.R + ROWS - 1) / ROWS);
    const int split = logical / nrb, rb = logical % nrb;
    {   // large |rx|: the lazy-max kernel handles those 256-row blocks (flags are per ROWS_WG rows; a workgroup whose rows straddle
        // two of them - ROWS does not divide ROWS_WG at CT = 3 - leaves only when BOTH are flagged: the merge ignores what it
        // computes for rows of a flagged block)
        const int64_t r_first = (int64_t)rb * ROWS, r_last = min(r_first + ROWS, p.R) - 1;
        if (p.safe_flags[(int)(r_first / ROWS_WG)] != 0 && p.safe_flags[(int)(r_last / ROWS_WG)] != 0) return;
    }
    const int t_beg = split * p.tiles_per_split;
    const int t_end = min(t_beg + p.tiles_per_split, p.ntiles);
    const int64_t nbase = (int64_t)t_beg * 32;
    // ring chunks of this range that exist in full; T = pipelined slots (32-item subtiles)
    int Cn = (int)min((int64_t)((t_end - t_beg) / SUB), (p.N - nbase) / G::BNF);
    Cn = max(Cn, 0);
    constexpr bool SPLIT_REM = pipe_splits_range<D, CT>();   // the rest of the range belongs to the other kernel
    const int T = SPLIT_REM ? pipe_slots_of<D, CT>(Cn) : Cn * SUB;

    const int64_t rw = (int64_t)rb * ROWS + wave * 16 * CT;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int lane_off[4];
    fast_lane_off<D, NW>(lane, wave, lane_off);
#pragma unroll
    // (requests beyond the last chunk repeat it: a constant number of chunks in flight keeps every counted vmcnt valid)
    for (int c0 = 0; c0 <= PG::PF; ++c0) {
        if constexpr (PG::DUMMY) {
            if (Cn > 0) fast_stage<D, NW>(p.E, nbase + (int64_t)min(c0, Cn - 1) * G::BNF, smem + c0 * CB, wave_u, lane_off);
        } else {
            if (c0 < Cn) fast_stage<D, NW>(p.E, nbase + (int64_t)c0 * G::BNF, smem + c0 * CB, wave_u, lane_off);
        }
    }

    bf16x8 xb[CT][G::KS];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int64_t r = rw + 16 * ct + c;
        const int64_t rl = r < p.R ? r : p.R - 1;
#pragma unroll
        for (int s = 0; s < G::KS; ++s) {
            const float4 v0 = *reinterpret_cast<const float4*>(p.rx + rl * D + 8 * fchunk<D>(s, g));
            const float4 v1 = *reinterpret_cast<const float4*>(p.rx + rl * D + 8 * fchunk<D>(s, g) + 4);
            xb[ct][s][0] = (__bf16)(v0.x * kLog2e); xb[ct][s][1] = (__bf16)(v0.y * kLog2e);
            xb[ct][s][2] = (__bf16)(v0.z * kLog2e); xb[ct][s][3] = (__bf16)(v0.w * kLog2e);
            xb[ct][s][4] = (__bf16)(v1.x * kLog2e); xb[ct][s][5] = (__bf16)(v1.y * kLog2e);
            xb[ct][s][6] = (__bf16)(v1.z * kLog2e); xb[ct][s][7] = (__bf16)(v1.w * kLog2e);
        }
    }
    f32x4 U[G::NDT][CT];
    f32x4 lsum[CT];
#pragma unroll
    for (int dt = 0; dt < G::NDT; ++dt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int i = 0; i < 4; ++i) U[dt][ct][i] = 0.f;
            asm volatile("" : "+a"(U[dt][ct]));
        }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int i = 0; i < 4; ++i) lsum[ct][i] = 0.f;
        asm volatile("" : "+a"(lsum[ct]));
    }
    const FastLane L = fast_lane<D>(lane);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    PipeRegs<CT> R2[2];
    f32x4 acc0[CT];
    bf16x8 af[2 * G::KS];
    float e[4 * CT][2];   // exponentials of the numerator stream in progress (it spans two slots)

    // LDS address of slot t's subtile and the seam a slot carries (runtime forms)
    auto lds_of = [&](int t) { return lds0 + (unsigned)(((t / SUB) % PG::NB) * CB + (t % SUB) * G::ST); };
    auto seam_of = [&](int t) {
        Pipe2Seam sm;
        sm.E = p.E;
        sm.do_seam = (t % SUB) == SUB - 1;
        const int cs = t / SUB + 1 + PG::PF;   // beyond the last chunk: request the last one again (never read)
        if constexpr (PG::DUMMY) sm.n_stage = sm.do_seam ? nbase + (int64_t)min(cs, Cn - 1) * G::BNF : -1;
        else sm.n_stage = (sm.do_seam && cs < Cn) ? nbase + (int64_t)cs * G::BNF : -1;
        sm.stage_buf = smem + (cs % PG::NB) * CB;
        sm.stage_lds = lds0 + (unsigned)((cs % PG::NB) * CB);
        sm.next_lbase = lds0;
        return sm;
    };

    int t = 0;
    if (T > 0) {
        if (PG::DUMMY || Cn > PG::PF) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PG::PF * 4) : "memory");   // chunk 0 landed
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        pipe_a_prologue<D, 0, PIPE_AD>(lds0, L.a0, af);
        {   // slot 0: nothing to drain yet
            Pipe2Seam sm = seam_of(0);
            sm.next_lbase = lds_of(T > 1 ? 1 : 0);
            pipe2_slot<D, CT, 0, 0, 0, true, false, 0, true>(lds0, lds0, L, xb, af, acc0, R2[0], R2[1], U, lsum, sm, wave_u, lane_off, e);
        }
        t = 1;
        // steady state: TR slots per trip, every LDS offset an immediate
        for (; PG::DUMMY ? (t + TR <= T) : ((t + TR - 1) / SUB + 3 <= Cn - 1); t += TR) {
            pipe_fence();   // loop entry: hipcc's preheader copies of the loop-carried values (plain v_mov's) need wait states before the
                            // first asm MFMA reads them (catalog_x3.h, round 3: a stale fragment in the first MFMA of the first trip)
#define PCVAE_P2(UU)                                                                                                      \
            if constexpr (UU < TR) {                                                                                      \
                constexpr int TL = 1 + UU, TG = UU, TN = 2 + UU;                                                          \
                constexpr int OL = ((TL / SUB) % PG::NB) * CB + (TL % SUB) * G::ST;                                      \
                constexpr int OG = ((TG / SUB) % PG::NB) * CB + (TG % SUB) * G::ST;                                      \
                constexpr int ON = ((TN / SUB) % PG::NB) * CB + (TN % SUB) * G::ST;                                      \
                constexpr bool SEAM = (TL % SUB) == SUB - 1;                                                              \
                Pipe2Seam s2 = seam_of(t + UU);                                                                           \
                s2.stage_lds = lds0 + (((1 + UU) / SUB + 1 + PG::PF) % PG::NB) * CB;   /* t = 1 (mod TR): a constant */      \
                pipe2_slot<D, CT, OL, OG, ON, SEAM, true, (PG::PF - 1) * 4, false>(lds0, lds0, L, xb, af, acc0, R2[TL & 1], R2[TG & 1], U, \
                                                                        lsum, s2, wave_u, lane_off, e);                   \
            }
            PCVAE_P2(0) PCVAE_P2(1) PCVAE_P2(2) PCVAE_P2(3) PCVAE_P2(4) PCVAE_P2(5) PCVAE_P2(6) PCVAE_P2(7)
            PCVAE_P2(8) PCVAE_P2(9) PCVAE_P2(10) PCVAE_P2(11) PCVAE_P2(12) PCVAE_P2(13) PCVAE_P2(14) PCVAE_P2(15)
            PCVAE_P2(16) PCVAE_P2(17) PCVAE_P2(18) PCVAE_P2(19) PCVAE_P2(20) PCVAE_P2(21) PCVAE_P2(22) PCVAE_P2(23)
#undef PCVAE_P2
            pipe_fence();  // latch
        }
        // the slots after the last full trip, at steady-state speed: a short trip (one chunk; two for SUB = 1) whose ring
        // position is a runtime base register; its seams drain the ring (vmcnt(0): how many chunks are still in flight is
        // no longer a compile-time number)
        // (D = 128: hipcc spills registers in this loop - a spill next to an unfenced asm MFMA stores stale data - so those
        // slots take the fenced path below; tools/isa_loop_check.py guards every unfenced loop)
        constexpr int TR2 = SUB > 2 ? SUB : 2;
        constexpr bool REM_FAST = D == 64;
        if constexpr (!SPLIT_REM) {
        for (; REM_FAST && t + TR2 <= T; t += TR2) {
            pipe_fence();   // loop entry: hipcc's preheader copies of the loop-carried values (plain v_mov's) need wait states before the
                            // first asm MFMA reads them (catalog_x3.h, round 3: a stale fragment in the first MFMA of the first trip)
#define PCVAE_P2R(UU)                                                                                                     \
            if constexpr (UU < TR2) {                                                                                     \
                constexpr bool SEAM = ((1 + UU) % SUB) == SUB - 1;                                                        \
                Pipe2Seam s2 = seam_of(t + UU);                                                                           \
                s2.next_lbase = lds_of(t + UU + 1 < T ? t + UU + 1 : t + UU);                                             \
                pipe2_slot<D, CT, 0, 0, 0, SEAM, true, 0, false>(lds_of(t + UU), lds_of(t + UU - 1), L, xb, af, acc0,     \
                                                                 R2[(1 + UU) & 1], R2[UU & 1], U, lsum, s2, wave_u, lane_off, e); \
            }
            PCVAE_P2R(0) PCVAE_P2R(1) PCVAE_P2R(2) PCVAE_P2R(3)
#undef PCVAE_P2R
            pipe_fence();  // latch
        }
        // at most TR2 - 1 slots are left: fenced (cold) slots
        for (; t < T; ++t) {
            Pipe2Seam s2 = seam_of(t);
            s2.next_lbase = lds_of(t + 1 < T ? t + 1 : t);
            if (t & 1) pipe2_slot<D, CT, 0, 0, 0, true, true, 0, true>(lds_of(t), lds_of(t - 1), L, xb, af, acc0, R2[1], R2[0], U, lsum, s2, wave_u, lane_off, e);
            else pipe2_slot<D, CT, 0, 0, 0, true, true, 0, true>(lds_of(t), lds_of(t - 1), L, xb, af, acc0, R2[0], R2[1], U, lsum, s2, wave_u, lane_off, e);
        }
        }
        {   // drain: the rest of the last slot's numerators, then its gradient chain
            bf16x8 pb[CT];
            pipe_fence();
            if ((T - 1) & 1) { pipe2_drain_ops<D, CT>(acc0, R2[1], e); pipe2_pack<CT>(R2[1], pb); }
            else { pipe2_drain_ops<D, CT>(acc0, R2[0], e); pipe2_pack<CT>(R2[0], pb); }
            asm volatile("s_nop 1" ::: "memory");
            pipe2_cold_gradient<D, CT>(lds_of(T - 1), L, pb, U, lsum);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive its wave
    // ---- tail: short / ragged chunks, staged synchronously with clamped addresses, one subtile at a time
    for (int tt = t_beg + T; !SPLIT_REM && tt < t_end; tt += 4) {
        __syncthreads();
        fast_stage_tail<D, NW>(p.E, p.N, (int64_t)tt * 32, smem);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int nsub = min(4, t_end - tt);
        for (int st = 0; st < nsub; ++st) pipe2_solo<D, CT>(lds0 + st * G::ST, (int64_t)(tt + st) * 32, p.N, L, xb, U, lsum);
    }
    pipe_fence();
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const float l = lsum[ct][0];
        const int64_t r = rw + 16 * ct + c;
        if (r < p.R) {
            const int64_t o = (int64_t)split * p.R + r;
            if (g == 0) { p.pm[o] = 0.f; p.pl[o] = l; }
#pragma unroll
            for (int dt = 0; dt < G::NDT; ++dt)
                *reinterpret_cast<float4*>(p.pU + o * D + 16 * dt + 4 * g) =
                    make_float4(U[dt][ct][0], U[dt][ct][1], U[dt][ct][2], U[dt][ct][3]);
        }
    }
}

__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }

// one wave per row: merge the split partials (log2 domain), target logit in the kernel's own arithmetic.
// Lane j owns split j's (m, l) (nsplit <= 64), so max / rescale / sum are wave reductions; the U rows of the
// splits are then streamed with 8 independent loads in flight per lane.
template <int D>
__global__ void __launch_bounds__(256) catalog_ce_merge_bf16_kernel(CatParamsB p, float* __restrict__ nll,
                                                                    float* __restrict__ lse, float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.R) return;
    const float mj = lane < p.nsplit ? p.pm[(int64_t)lane * p.R + r] : -INFINITY;
    const float lj = lane < p.nsplit ? p.pl[(int64_t)lane * p.R + r] : 0.f;
    float M = mj;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o, 64));
    const float sj = lane < p.nsplit ? exp2f(mj - M) : 0.f;
    float L = lj * sj;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) L += __shfl_xor(L, o, 64);
    const int64_t t = p.target[r];
    const bool t_ok = t >= 0 && t < p.N;
    // target logit: each lane takes D/64 of the products, same bf16 operands as the MFMA chain
    float zt = 0.f;
    if (t_ok)
        for (int k = lane; k < D; k += 64)
            zt = fmaf(bf16_to_f32(p.E[t * D + k]), (float)(__bf16)(p.rx[r * D + k] * kLog2e), zt);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zt += __shfl_xor(zt, o, 64);
    const float lse_r = (M + log2f(L)) * kLn2;
    if (lane == 0) {
        nll[r] = t_ok ? lse_r - zt * kLn2 : NAN;
        if (lse) lse[r] = lse_r;
    }
    if (dx) {
        const float invL = 1.f / L;
        for (int d = lane; d < D; d += 64) {
            float u = 0.f;
            int j = 0;
            for (; j + 8 <= p.nsplit; j += 8) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = p.pU[((int64_t)(j + q) * p.R + r) * D + d];
#pragma unroll
                for (int q = 0; q < 8; ++q) u = fmaf(v[q], __shfl(sj, j + q, 64), u);
            }
            for (; j < p.nsplit; ++j) u = fmaf(p.pU[((int64_t)j * p.R + r) * D + d], __shfl(sj, j, 64), u);
            dx[r * D + d] = t_ok ? (u * invL - bf16_to_f32(p.E[t * D + d])) * p.dx_scale : NAN;
        }
    }
}

template <int D>
int launch_ce_b(CatParamsB p, int mask_mode, bool want_dx, float e_max_norm, uint8_t* flags, float* nll, float* lse,
                float* dx, hipStream_t st) {
    using G = GeoB<D>;
    const size_t lds = 2 * G::CHUNK_BYTES;
    const dim3 grid((unsigned)(p.nrb * p.nsplit)), block(512);
    p.safe_flags = nullptr;
    p.run_if_flag = 1;
    {
        // row blocks with a small logit bound run the max-free kernel, the others the lazy-max kernel;
        // both launches cover the whole grid and each workgroup exits at once if the other kernel owns it
        // masked / loss-only calls (validation, n_neg < N) all take the lazy-max kernel
        const bool fast_ok = (mask_mode == MASK_NONE) && want_dx;
        hipLaunchKernelGGL((catalog_row_bound_kernel<D>), dim3((unsigned)p.nrb), dim3(256), 0, st, p.rx, p.R,
                           fast_ok ? e_max_norm : 0.f, flags);
        p.safe_flags = flags;
        if (fast_ok) {
            // D = 256 always, D = 128 / 64 on long catalog ranges: the software-pipelined kernel (64 rows per wave for
            // D <= 128).  Its fill / drain / last slots run fenced at about half speed, so short ranges (the 8-GPU shards,
            // small catalogs) stay on the two-waves-per-SIMD kernel.  PCVAE_PIPE_MIN_TILES (read per launch) moves the
            // threshold: the tests force the pipelined kernels onto small shapes with it.
            constexpr bool ALWAYS_PIPE = D == 256;
            constexpr int CT = D == 256 ? PIPE_CT256 : D == 64 ? PIPE_CT64 : 4;
            if (catalog_bf16_pipelined(D, p.tiles_per_split)) {
                constexpr int lds_pipe = PipeGeo<D, CT>::NB * 16384;
                if (int rc_optin = lds_optin(reinterpret_cast<const void*>(&catalog_ce_bf16_pipe_kernel<D, CT>), lds_pipe)) return rc_optin;
                const dim3 g2((unsigned)(cdiv(p.R, PipeGeo<D, CT>::ROWS) * p.nsplit));
                hipLaunchKernelGGL((catalog_ce_bf16_pipe_kernel<D, CT>), g2, dim3(256), lds_pipe, st, p);
                if constexpr (pipe_splits_range<D, CT>()) {   // the rest of every range, added into the same partial
                    constexpr int lds_fast = 65536;
                    if (int rc_optin = lds_optin(reinterpret_cast<const void*>(&catalog_ce_bf16_fast_kernel<D>), lds_fast)) return rc_optin;
                    CatParamsB pr = p;
                    const int64_t nblk = (int64_t)grid.x / p.nsplit;   // row blocks of this kernel
                    pr.rem_mode = (int)std::max<int64_t>(1, std::min<int64_t>(p.nsplit, 256 / std::max<int64_t>(nblk, 1)));
                    hipLaunchKernelGGL((catalog_ce_bf16_fast_kernel<D>), dim3((unsigned)(nblk * pr.rem_mode)), block, lds_fast, st, pr);
                }
            } else if constexpr (!ALWAYS_PIPE) {
                constexpr int lds_fast = 65536;  // ring of four 16 KB chunks (also holds the <= 64 KB synchronous tail image)
                if (int rc_optin = lds_optin(reinterpret_cast<const void*>(&catalog_ce_bf16_fast_kernel<D>), lds_fast)) return rc_optin;
                hipLaunchKernelGGL((catalog_ce_bf16_fast_kernel<D>), grid, block, lds_fast, st, p);
            }
        }
        int rc0 = check_launch("catalog_ce_bf16_fast");
        if (rc0 != PCVAE_OK) return rc0;
    }
#define PCVAE_CEB(MASKV, DXV)                                                                                    \
    do {                                                                                                         \
        if (int rc_optin = lds_optin(reinterpret_cast<const void*>(&catalog_ce_bf16_kernel<D, MASKV, DXV>), (int)lds)) return rc_optin;                                                                                                        \
        hipLaunchKernelGGL((catalog_ce_bf16_kernel<D, MASKV, DXV>), grid, block, lds, st, p);                    \
    } while (0)
    if (want_dx) {
        if (mask_mode == MASK_NONE) PCVAE_CEB(MASK_NONE, true);
        else if (mask_mode == MASK_PHILOX) PCVAE_CEB(MASK_PHILOX, true);
        else PCVAE_CEB(MASK_BYTES, true);
    } else {
        if (mask_mode == MASK_NONE) PCVAE_CEB(MASK_NONE, false);
        else if (mask_mode == MASK_PHILOX) PCVAE_CEB(MASK_PHILOX, false);
        else PCVAE_CEB(MASK_BYTES, false);
    }
#undef PCVAE_CEB
    int rc = check_launch("catalog_ce_bf16");
    if (rc != PCVAE_OK) return rc;
    hipLaunchKernelGGL((catalog_ce_merge_bf16_kernel<D>), dim3((unsigned)cdiv(p.R, 4)), dim3(256), 0, st, p, nll, lse,
                       want_dx ? dx : nullptr);
    return check_launch("catalog_ce_merge_bf16");
}

// =============================================================================================
// K6 at bf16 speed with EXACT fp32 results ("screened" argmax), D = 64 / 128 / 256.
//
// The greedy item id must be bit-exact against the fp32 arithmetic (k-ordered fmaf chain, lowest index on ties), but
// the exact f32-MFMA kernel runs at 1/16 of the bf16 rate.  Two bf16 passes over the catalog give the same answer:
//   bound   approximate scores s~_n = <bf16(x), bf16(E_n)> (fp32 accumulate) satisfy |s~_n - s_n| <= eps_r :=
//           (2^-8 * 1.02 + 2e-5) * ||x_r|| * max_n ||E_n||  (two RNE roundings of 2^-9 each per product, Cauchy-Schwarz
//           over k, fp32 accumulation slack).  With m~ = max_n s~_n every exact maximiser n* has
//           s~_{n*} >= s_{n*} - eps >= s_{n~} - eps >= m~ - 2 eps, and the same holds for any LOWER bound of m~.
//   pass A  m0_r = max of s~ over a PREFIX of the catalog (N/16 items; the whole catalog when it is small): a lower
//           bound of m~ that is already within a handful of items of it.
//   pass B  one full bf16 pass; every item with s~_n >= max(m0_r, lane-local running max) - 2 eps is a candidate
//           (~10-20 per row of 10^6): its EXACT score is computed on the spot as the k-ordered fmaf chain over the
//           fp32 table - the same chain v_mfma_f32_32x32x2_f32 and oracle/catalog_oracle.c evaluate - and folded into
//           best[r] with a 64-bit atomicMax on (ordered score bits << 32 | ~n): largest score, then lowest index.
// Result: ids and winning scores identical to catalog_argmax_f32_kernel at several times its speed
// (an adversarially ordered catalog only costs time - more candidates - never correctness).
// =============================================================================================
struct ScreenParams {
    const float* x;        // [R, D] fp32
    const uint16_t* Eb;    // [N, D] bf16 bits
    const float* Ef;       // [N, D] fp32 (exact rescoring)
    int64_t R, N;
    int nrb, nsplit, tiles_per_split, ntiles;
    float* pm;                       // [nsplit][R] approximate maxima (pass A)
    float* thr;                      // [R] m~ - 2 eps
    float* eps2;                     // [R] 2 eps
    unsigned long long* best_key;    // [R]
    float e_max_norm;
    unsigned int* overflow;          // set by the pipelined pass B when a workgroup's candidate list is full
    int only_if_overflow;            // this launch is the exact redo of pass B: it returns at once unless *overflow != 0
};

__device__ __forceinline__ unsigned int ordered_bits(float f) {
    const unsigned int u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float unordered_bits(unsigned int o) {
    return __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xffffffffu));
}

// exact fp32 score of (row, n) - k-ordered fmaf chain - folded into best_key[row]
template <int D>
__device__ __forceinline__ void screen_rescore(const ScreenParams& p, const int64_t row, const int64_t n) {
    const float4* e = reinterpret_cast<const float4*>(p.Ef + n * D);
    const float4* xr = reinterpret_cast<const float4*>(p.x + row * D);
    float sc = 0.f;
#pragma unroll 1
    for (int k = 0; k < D / 4; k += 8) {  // 16 independent 16-byte loads in flight, then 32 fmaf
        float4 ev[8], xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { ev[j] = e[k + j]; xv[j] = xr[k + j]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc = fmaf(ev[j].x, xv[j].x, sc); sc = fmaf(ev[j].y, xv[j].y, sc);
            sc = fmaf(ev[j].z, xv[j].z, sc); sc = fmaf(ev[j].w, xv[j].w, sc);
        }
    }
    const unsigned long long key =
        ((unsigned long long)ordered_bits(sc) << 32) | (unsigned long long)(0xffffffffu - (unsigned int)n);
    atomicMax(p.best_key + row, key);
}

// Candidates are parked in an LDS list (one ds_add_rtn + one ds_write) and rescored by the whole workgroup after the
// catalog range is done: an inline rescoring stalls its wave for microseconds and, through the per-chunk barrier, the
// seven other waves with it.  A full list degrades to inline rescoring (slow, still exact).
constexpr int SCREEN_RING = 4 * 16384;
constexpr int SCREEN_CAND_CAP = 3072;
constexpr int SCREEN_LDS_BYTES = SCREEN_RING + SCREEN_CAND_CAP * 8 + 16;
constexpr int SCREEN_NW = 8;  // no U accumulator here: 8 waves (two per SIMD) fit for every D

template <int D, int PASS, bool CHECK_N, int OFF>
__device__ __forceinline__ void screen_subtile(const ScreenParams& p, char* smem, const int off, const int64_t n0,
                                               const bf16x8 (&xb)[2][FastGeo<D>::KS], float (&m)[2], float (&thr)[2],
                                               const float (&eps2)[2], const int64_t (&row)[2], const FastLane& L) {
    f32x4 acc[2][2];
    fast_logits<D, CHECK_N, OFF>(smem, off, n0, p.N, xb, acc, L);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        float v = fmaxf(fmaxf(acc[0][ct][0], acc[0][ct][1]), fmaxf(acc[0][ct][2], acc[0][ct][3]));
        v = fmaxf(v, fmaxf(fmaxf(acc[1][ct][0], acc[1][ct][1]), fmaxf(acc[1][ct][2], acc[1][ct][3])));
        if (PASS == 0) {
            m[ct] = fmaxf(m[ct], v);
        } else if (v >= thr[ct]) {
            unsigned int* cnt = reinterpret_cast<unsigned int*>(smem + SCREEN_RING + SCREEN_CAND_CAP * 8);
            uint2* list = reinterpret_cast<uint2*>(smem + SCREEN_RING);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (acc[rt][ct][i] >= thr[ct]) {
                        const int64_t n = n0 + 16 * rt + 4 * L.g + i;
                        const unsigned int slot = atomicAdd(cnt, 1u);
                        if (slot < (unsigned)SCREEN_CAND_CAP) list[slot] = make_uint2((unsigned int)row[ct], (unsigned int)n);
                        else screen_rescore<D>(p, row[ct], n);
                    }
            // everything this lane meets later only matters if it comes within 2 eps of what it has already seen
            thr[ct] = fmaxf(thr[ct], v - eps2[ct]);
        }
    }
}

template <int D, int PASS, int OFFB>
__device__ __forceinline__ void screen_chunk(const ScreenParams& p, char* smem, const int off, const int64_t nA,
                                             const bf16x8 (&xb)[2][FastGeo<D>::KS], float (&m)[2], float (&thr)[2],
                                             const float (&eps2)[2], const int64_t (&row)[2], const FastLane& L) {
    using G = FastGeo<D>;
    screen_subtile<D, PASS, false, OFFB>(p, smem, off, nA, xb, m, thr, eps2, row, L);
    if constexpr (G::SUB >= 2) screen_subtile<D, PASS, false, OFFB + G::ST>(p, smem, off, nA + 32, xb, m, thr, eps2, row, L);
    if constexpr (G::SUB >= 4) {
        screen_subtile<D, PASS, false, OFFB + 2 * G::ST>(p, smem, off, nA + 64, xb, m, thr, eps2, row, L);
        screen_subtile<D, PASS, false, OFFB + 3 * G::ST>(p, smem, off, nA + 96, xb, m, thr, eps2, row, L);
    }
}

template <int D, int PASS>
__global__ void __launch_bounds__(512, 1) catalog_screen_bf16_kernel(ScreenParams p) {
    using G = FastGeo<D>;
    constexpr int CB = 16384, NW = SCREEN_NW;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / p.nrb, rb = logical % p.nrb;
    const int t_beg = split * p.tiles_per_split;
    const int t_end = min(t_beg + p.tiles_per_split, p.ntiles);
    const int64_t nbase = (int64_t)t_beg * 32;
    int n_full = (int)min((int64_t)((t_end - t_beg) / G::SUB), (p.N - nbase) / G::BNF);
    n_full = max(n_full, 0);
    const int64_t rw = (int64_t)rb * ROWS_WG + wave * 32;
    if (PASS == 1) {
        if (p.only_if_overflow && *p.overflow == 0u) return;   // the pipelined pass B parked everything: nothing to redo
        if (threadIdx.x == 0) *reinterpret_cast<unsigned int*>(smem + SCREEN_RING + SCREEN_CAND_CAP * 8) = 0u;
        __syncthreads();
    }

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int lane_off[16 / NW];
    fast_lane_off<D, NW>(lane, wave, lane_off);
#pragma unroll
    for (int c0 = 0; c0 < 3; ++c0)
        if (c0 < n_full) fast_stage<D, NW>(p.Eb, nbase + (int64_t)c0 * G::BNF, smem + c0 * CB, wave_u, lane_off);

    bf16x8 xb[2][G::KS];
    int64_t row[2];
    float thr[2] = {INFINITY, INFINITY}, eps2[2] = {0.f, 0.f}, m[2] = {-INFINITY, -INFINITY};
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int64_t r = rw + 16 * ct + c;
        row[ct] = r < p.R ? r : p.R - 1;
        if (PASS == 1 && r < p.R) { thr[ct] = p.thr[r]; eps2[ct] = p.eps2[r]; }  // padding rows: never a candidate
#pragma unroll
        for (int s = 0; s < G::KS; ++s) {
            const float4 v0 = *reinterpret_cast<const float4*>(p.x + row[ct] * D + 8 * fchunk<D>(s, g));
            const float4 v1 = *reinterpret_cast<const float4*>(p.x + row[ct] * D + 8 * fchunk<D>(s, g) + 4);
            xb[ct][s][0] = (__bf16)v0.x; xb[ct][s][1] = (__bf16)v0.y; xb[ct][s][2] = (__bf16)v0.z; xb[ct][s][3] = (__bf16)v0.w;
            xb[ct][s][4] = (__bf16)v1.x; xb[ct][s][5] = (__bf16)v1.y; xb[ct][s][6] = (__bf16)v1.z; xb[ct][s][7] = (__bf16)v1.w;
        }
    }
    const FastLane L = fast_lane<D>(lane);

    int cc = 0;
    const int n_pipe = n_full >= 3 ? n_full - 2 : 0;
    for (; cc + 4 <= n_pipe; cc += 4) {
#define PCVAE_RING_STEP(UU)                                                                                          \
        {                                                                                                            \
            const int64_t nA = nbase + (int64_t)(cc + UU) * G::BNF;                                                  \
            fast_seam<2 * (16 / NW)>();                                                                              \
            if (cc + UU + 3 < n_full)                                                                                \
                fast_stage<D, NW>(p.Eb, nA + 3 * G::BNF, smem + ((UU + 3) & 3) * CB, wave_u, lane_off);              \
            screen_chunk<D, PASS, UU * CB>(p, smem, 0, nA, xb, m, thr, eps2, row, L);                                \
        }
        PCVAE_RING_STEP(0)
        PCVAE_RING_STEP(1)
        PCVAE_RING_STEP(2)
        PCVAE_RING_STEP(3)
#undef PCVAE_RING_STEP
    }
    for (; cc < n_full; ++cc) {
        const int64_t nA = nbase + (int64_t)cc * G::BNF;
        fast_seam<0>();
        if (cc + 3 < n_full) fast_stage<D, NW>(p.Eb, nA + 3 * G::BNF, smem + ((cc + 3) & 3) * CB, wave_u, lane_off);
        screen_chunk<D, PASS, 0>(p, smem, (cc & 3) * CB, nA, xb, m, thr, eps2, row, L);
    }
    for (int t = t_beg + G::SUB * n_full; t < t_end; t += 4) {
        __syncthreads();
        fast_stage_tail<D, NW>(p.Eb, p.N, (int64_t)t * 32, smem);
        __syncthreads();
        const int nsub = min(4, t_end - t);
        for (int st = 0; st < nsub; ++st)
            screen_subtile<D, PASS, true, 0>(p, smem, st * G::ST, (int64_t)(t + st) * 32, xb, m, thr, eps2, row, L);
    }
    if (PASS == 1) {  // rescore the parked candidates, one per thread
        __syncthreads();
        const unsigned int cnt =
            min(*reinterpret_cast<const unsigned int*>(smem + SCREEN_RING + SCREEN_CAND_CAP * 8), (unsigned)SCREEN_CAND_CAP);
        const uint2* list = reinterpret_cast<const uint2*>(smem + SCREEN_RING);
        for (unsigned int j = threadIdx.x; j < cnt; j += 512) screen_rescore<D>(p, (int64_t)list[j].x, (int64_t)list[j].y);
    }
    if (PASS == 0) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            float v = m[ct];
            v = fmaxf(v, __shfl_xor(v, 16, 64));
            v = fmaxf(v, __shfl_xor(v, 32, 64));
            const int64_t r = rw + 16 * ct + c;
            if (g == 0 && r < p.R) p.pm[(int64_t)split * p.R + r] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The screening passes on the software-pipelined logits chain (one wave per SIMD, CT column tiles of 16 rows per wave, every
// MFMA / LDS read an inline-asm statement in schedule order - see catalog_ce_bf16_pipe_kernel).  Slot t = logits chain L(t)
// into accumulator set t & 1, while the VALU looks at the logits of subtile t-1 (maxima / threshold tests, plain C++ that
// hipcc schedules between the MFMAs; an empty asm on the previous accumulators a few MFMAs into the slot keeps those reads
// behind the MFMAs that wrote them).  Ring and seams as in the CE kernel; the seam sits at the end of a chunk's last slot.
// ---------------------------------------------------------------------------------------------------------------
// 8-byte entries: (row, first item of the lane's eight / 4 << 8 | bit per item) - the first item is a multiple of 4 below 2^26 (larger
// catalogs take the two-waves kernel, see launch_screened); a quarter of the list per wave.
//   D = 64 : 2032 entries - ring (64 KB) + list + counters = 81 808 bytes, so that TWO workgroups fit a CU's 160 KB (round 6: -17 % on
//            config 3's generate batch, tools/screen_wg_ab.sh; until then 3072 entries of 16 bytes and one workgroup per CU everywhere);
//   D >= 128: 6144 entries (48 KB), one workgroup per CU (two were measured at D = 128, config 4: +0.9 %; at D = 256 the ring is 96 KB).
// SCREEN_PIPE_ENTRIES_64: A/B builds.
#ifndef SCREEN_PIPE_ENTRIES_64
#define SCREEN_PIPE_ENTRIES_64 2032
#endif
template <int D>
__host__ __device__ constexpr int screen_pipe_cap() { return D == 64 ? SCREEN_PIPE_ENTRIES_64 : 6144; }
constexpr int SCREEN_PIPE_ENTRY_BYTES = 8;
static_assert(2 * (4 * 16384 + screen_pipe_cap<64>() * SCREEN_PIPE_ENTRY_BYTES + 16) <= 160 * 1024 || SCREEN_PIPE_ENTRIES_64 != 2032,
              "two D = 64 screening workgroups (64 KB ring + list + counters each) must fit a CU's 160 KB of LDS");
constexpr int64_t SCREEN_PIPE_MAX_ITEMS = 1ll << 26;

template <int CT>
struct ScreenAcc { f32x4 acc[2][CT]; };

template <int CT>
struct ScreenState {        // per lane: rows 16 ct + c of the wave
    float m[CT], thr[CT], eps2[CT];
    float tmp;           // the looks' temporary, alive over the whole range (see vmax8_into)
    int64_t row[CT];
    uint64_t mk[CT];     // steady-state slots of pass B: lanes of column tile ct that saw a logit over their threshold (SGPR pair)
    unsigned int cnt;    // pass B: entries this WAVE has parked in its own quarter of the candidate list (wave-uniform)
};

// Largest of a lane's eight logits as ONE asm statement (three v_max3_f32 + one more): fmaxf() on values that come out of asm
// MFMAs costs a canonicalising v_max x, x per input (11 VALU ops instead of 4), and hipcc puts an s_nop 0 in front of every asm
// statement that reads the result of another one (its hazard recogniser does not count asm as wait states) - one statement,
// one nop.  Vector issue, not the MFMA pipe, is what bounds a screening slot.
__device__ __forceinline__ float vmax8(const f32x4& lo, const f32x4& hi) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7\n\tv_max_f32 %0, %0, %8"
        : "=&v"(r)
        : "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]));
    return r;
}
// (volatile, with the temporary in a register the CALLER keeps alive over the whole slot - ScreenState::tmp: the statement then stays where
// the schedule puts it, in the shadow of the step's MFMAs, and its temporary can never be a register a queued MFMA still has to read.
// Non-volatile, hipcc sank all of a slot's looks behind its last MFMA: 16-20 vector instructions at a lone wave's 5.5-cycle cadence with
// the matrix pipe idle - tools/valu_rate_probe.hip, round 6.)
__device__ __forceinline__ void vmax8_into(float& m, float& t, const f32x4& lo, const f32x4& hi) {   // m = max(m, the eight)
    asm volatile("v_max3_f32 %1, %2, %3, %4\n\tv_max3_f32 %1, %1, %5, %6\n\tv_max3_f32 %1, %1, %7, %8\n\tv_max3_f32 %0, %0, %1, %9"
        : "+v"(m), "+v"(t)
        : "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]));
}

// Pass B, steady state: the eight-way maximum AND the threshold test in one statement, the outcome as a wave mask in SGPRs.
// The four masks of a slot are OR-ed and tested ONCE at the end of the slot (one scalar branch per slot instead of four
// exec-mask branches); a slot that has a candidate anywhere then runs screen_look() per column tile.
__device__ __forceinline__ uint64_t screen_peek(float& t, const f32x4& lo, const f32x4& hi, const float thr) {
    uint64_t mask;
    asm volatile("v_max3_f32 %1, %2, %3, %4\n\tv_max3_f32 %1, %1, %5, %6\n\tv_max3_f32 %1, %1, %7, %8\n\tv_max_f32 %1, %1, %9\n\t"
        "v_cmp_ge_f32 %0, %1, %10"
        : "=s"(mask), "+v"(t)
        : "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "v"(thr));
    return mask;
}

template <int D, int CT, int PASS, bool CHECK_N>
__device__ __forceinline__ void screen_look(const ScreenParams& p, char* cand, ScreenAcc<CT>& a, const int64_t n0,
                                            ScreenState<CT>& st, const int g, const int ct) {
    if (CHECK_N) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (n0 + 16 * rt + 4 * g + i >= p.N) a.acc[rt][ct][i] = -INFINITY;
    }
    if (PASS == 0) {
        vmax8_into(st.m[ct], st.tmp, a.acc[0][ct], a.acc[1][ct]);
    } else {
        // Every caller reaches this point with all 64 lanes active.  A lane whose best logit passes its threshold parks its
        // eight items (n0 + 16 rt + 4 g + i) as ONE entry with a bit per item that passed; the final phase rescores the marked
        // items exactly.  Each wave owns a quarter of the list and counts its entries in a register: an entry costs a ballot,
        // an mbcnt and one ds_write - no LDS atomic, no wait that would also drain the A fragments in flight.  A full quarter
        // raises the overflow flag: the (slow, exact) two-waves kernel then redoes pass B.
        const float v = vmax8(a.acc[0][ct], a.acc[1][ct]);
        const bool hit = v >= st.thr[ct];
        const uint64_t act = __ballot(hit);
        if (act != 0) {
            if (hit) {
                const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)act, 0u));
                const unsigned int slot = st.cnt + rank;
                unsigned int mask = 0;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) mask |= (a.acc[rt][ct][i] >= st.thr[ct] ? 1u : 0u) << (4 * rt + i);
                uint2* list = reinterpret_cast<uint2*>(cand) + (threadIdx.x >> 6) * (screen_pipe_cap<D>() / 4);
                if (slot < (unsigned)(screen_pipe_cap<D>() / 4))
                    list[slot] = make_uint2((unsigned int)st.row[ct], ((unsigned int)((n0 + 4 * g) >> 2) << 8) | mask);
                else *p.overflow = 1u;
                // everything this lane meets later only matters if it comes within 2 eps of what it has already seen
                st.thr[ct] = fmaxf(st.thr[ct], v - st.eps2[ct]);
            }
            st.cnt += (unsigned int)__popcll(act);
        }
    }
}

struct ScreenSeam {          // all wave-uniform
    const uint16_t* E;
    bool do_seam;            // this slot ends a ring chunk (runtime form, fenced slots only)
    int64_t n_stage;         // first item of the chunk to request
    char* stage_buf;
    unsigned stage_lds;      // the same ring buffer as an LDS byte address (steady-state seams: pipe_stage)
    unsigned next_lbase;     // LDS address (minus the immediate) of the NEXT slot's subtile
};

// L(t) with the look at subtile t-1 spread over it: column tile ct of the previous accumulators is released to the VALU
// after step (ct + 1) * NI / CT - 1 of this chain (>= 2 CT MFMAs after the MFMAs that wrote them).  In the middle of the
// chain: the seam (if this slot ends a ring chunk) and the first A fragments of the NEXT slot, so no slot starts on a cold
// LDS read.
template <int D, int CT, int PASS, int OFF, int OFFN, int I, bool SEAM, int VM, bool HAS_PREV, bool COLD>
__device__ __forceinline__ void screen_pipe_logits(const ScreenParams& p, char* cand, const unsigned lbase, const int a0,
                                                   bf16x8 (&af)[2 * FastGeo<D>::KS],
                                                   const bf16x8 (&xb)[CT][FastGeo<D>::KS], ScreenAcc<CT>& cur,
                                                   ScreenAcc<CT>& prev, const int64_t n_prev, ScreenState<CT>& st, const int g,
                                                   const ScreenSeam& sm, const int wave_u, const int (&lane_off)[4]) {
    constexpr int NI = 2 * FastGeo<D>::KS, MID = NI / 2;
    if constexpr (I < NI) {
        if constexpr (I == MID) {
            if constexpr (SEAM) {
                if constexpr (COLD) {
                    pipe_fence();
                    if (sm.do_seam) {
                        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                        fast_stage<D, 4>(sm.E, sm.n_stage, sm.stage_buf, wave_u, lane_off);
                    }
                    pipe_fence();
                } else {
#if defined(SCREEN_PROBE_NO_SEAMWAIT)     // probe builds only (with SCREEN_PROBE_NO_CAND): the steady-state seam without its wait + barrier
#elif defined(SCREEN_PROBE_NO_BARRIER)    // ... or without its barrier only
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");
#else
                    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(VM) : "memory");
#endif
                    pipe_stage<D>(sm.E, sm.n_stage, sm.stage_lds, wave_u, lane_off);
                }
            }
            static_assert(PIPE_AD <= MID, "af[0 .. PIPE_AD) must be consumed before the next slot's fragments land in them");
            pipe_a_prologue<D, OFFN, PIPE_AD>(sm.next_lbase, a0, af);   // first fragments of the next slot
        }
        // fragments younger than A(I) still in flight: the ones requested after it, and the next slot's first fragments if they
        // were requested while A(I) was already in flight
        constexpr int extra = (I >= MID && I < MID + PIPE_AD) ? PIPE_AD : 0;
        if constexpr (I + PIPE_AD < NI) pipe_a_issue<D, OFF, I + PIPE_AD>(lbase, a0, af[I + PIPE_AD]);
#ifdef SCREEN_PROBE_NO_LDSWAIT     // probe builds only (with SCREEN_PROBE_NO_CAND): the steady-state steps do not wait for their A fragments
        if constexpr (COLD) lgkm_wait<(I + PIPE_AD < NI ? PIPE_AD : NI - 1 - I) + extra>();
#else
        lgkm_wait<(I + PIPE_AD < NI ? PIPE_AD : NI - 1 - I) + extra>();
#endif
        constexpr int s = I >> 1, rt = I & 1;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            if constexpr (s == 0) mfma_v0<COLD>(cur.acc[rt][ct], af[I], xb[ct][s]);
            else mfma_v<COLD>(cur.acc[rt][ct], af[I], xb[ct][s]);
        }
        if constexpr (HAS_PREV && (I + 1) % (NI / CT) == 0) {
            constexpr int ct = (I + 1) / (NI / CT) - 1;
            asm volatile("" : "+v"(prev.acc[0][ct]), "+v"(prev.acc[1][ct]));   // not before this point of the chain
            if constexpr (PASS == 1 && !COLD) st.mk[ct] = screen_peek(st.tmp, prev.acc[0][ct], prev.acc[1][ct], st.thr[ct]);
            else screen_look<D, CT, PASS, false>(p, cand, prev, n_prev, st, g, ct);
        }
        screen_pipe_logits<D, CT, PASS, OFF, OFFN, I + 1, SEAM, VM, HAS_PREV, COLD>(p, cand, lbase, a0, af, xb, cur, prev,
                                                                                  n_prev, st, g, sm, wave_u, lane_off);
    } else if constexpr (HAS_PREV && PASS == 1 && !COLD) {
        uint64_t any = st.mk[0];
#pragma unroll
        for (int ct = 1; ct < CT; ++ct) any |= st.mk[ct];
        // rare: some lane of the wave has a candidate in the previous subtile (unlikely: hipcc then lays the ~250 instructions of the
        // block out of line and the steady-state trip falls through - 906 instructions for 16 slots instead of 4 781)
#ifdef SCREEN_PROBE_NO_BRANCH     // probe builds only (with SCREEN_PROBE_NO_CAND): the masks are computed, nothing looks at them
        asm volatile("" :: "s"(any));
        if (false) {
#else
        if (__builtin_expect(any != 0, 0)) {
#endif
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                if (st.mk[ct] != 0) screen_look<D, CT, PASS, false>(p, cand, prev, n_prev, st, g, ct);   // scalar test again
        }
    }
}

template <int D, int CT, int PASS>
__global__ void __launch_bounds__(256, 2) catalog_screen_pipe_kernel(ScreenParams p) {
    using G = FastGeo<D>;
    using PG = PipeGeo<D, CT>;
    constexpr int CB = 16384, NW = 4, ROWS = PG::ROWS, SUB = G::SUB, TR = PG::TR, NB = PG::NB, PF = PG::PF;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* cand = smem + NB * CB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int nrb = (int)((p.R + ROWS - 1) / ROWS);
    const int split = logical / nrb, rb = logical % nrb;
    const int t_beg = split * p.tiles_per_split;
    const int t_end = min(t_beg + p.tiles_per_split, p.ntiles);
    const int64_t nbase = (int64_t)t_beg * 32;
    int Cn = (int)min((int64_t)((t_end - t_beg) / SUB), (p.N - nbase) / G::BNF);
    Cn = max(Cn, 0);
    const int T = Cn * SUB;
    const int64_t rw = (int64_t)rb * ROWS + wave * 16 * CT;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int lane_off[4];
    fast_lane_off<D, NW>(lane, wave, lane_off);
    for (int c0 = 0; c0 <= PF; ++c0)   // requests beyond the last chunk repeat it (constant number of chunks in flight)
        if (Cn > 0) fast_stage<D, NW>(p.Eb, nbase + (int64_t)min(c0, Cn - 1) * G::BNF, smem + c0 * CB, wave_u, lane_off);

    bf16x8 xb[CT][G::KS];
    ScreenState<CT> st;
    st.cnt = 0u;
    st.tmp = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int64_t r = rw + 16 * ct + c;
        st.row[ct] = r < p.R ? r : p.R - 1;
        st.m[ct] = -INFINITY; st.thr[ct] = INFINITY; st.eps2[ct] = 0.f;
        if (PASS == 1 && r < p.R) { st.thr[ct] = p.thr[r]; st.eps2[ct] = p.eps2[r]; }  // padding rows: never a candidate
#pragma unroll
        for (int s = 0; s < G::KS; ++s) {
            const float4 v0 = *reinterpret_cast<const float4*>(p.x + st.row[ct] * D + 8 * fchunk<D>(s, g));
            const float4 v1 = *reinterpret_cast<const float4*>(p.x + st.row[ct] * D + 8 * fchunk<D>(s, g) + 4);
            xb[ct][s][0] = (__bf16)v0.x; xb[ct][s][1] = (__bf16)v0.y; xb[ct][s][2] = (__bf16)v0.z; xb[ct][s][3] = (__bf16)v0.w;
            xb[ct][s][4] = (__bf16)v1.x; xb[ct][s][5] = (__bf16)v1.y; xb[ct][s][6] = (__bf16)v1.z; xb[ct][s][7] = (__bf16)v1.w;
        }
    }
    const FastLane L = fast_lane<D>(lane);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    ScreenAcc<CT> A2[2];
    bf16x8 af[2 * G::KS];
    auto lds_of = [&](int t) { return lds0 + (unsigned)(((t / SUB) % NB) * CB + (t % SUB) * G::ST); };
    // the seam a slot carries in the middle of its chain (runtime form)
    auto seam_of = [&](int t) {
        ScreenSeam sm;
        sm.E = p.Eb;
        sm.do_seam = (t % SUB) == SUB - 1;
        const int cs = t / SUB + 1 + PF;   // beyond the last chunk: request the last one again (never read)
        sm.n_stage = nbase + (int64_t)min(cs, Cn - 1) * G::BNF;
        sm.stage_buf = smem + (cs % NB) * CB;
        sm.stage_lds = lds0 + (unsigned)((cs % NB) * CB);
        sm.next_lbase = lds0;
        return sm;
    };

    if (T > 0) {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PF * 4) : "memory");
        pipe_a_prologue<D, 0, PIPE_AD>(lds0, L.a0, af);
        {   // slot 0: nothing to look at yet
            ScreenSeam sm = seam_of(0);
            sm.next_lbase = lds_of(T > 1 ? 1 : 0);
            pipe_fence();
            screen_pipe_logits<D, CT, PASS, 0, 0, 0, true, 0, false, true>(p, cand, lds0, L.a0, af, xb, A2[0], A2[1], 0, st, g, sm,
                                                                         wave_u, lane_off);
            pipe_fence();
        }
        int t = 1;
        for (; t + TR <= T; t += TR) {   // steady state: every LDS offset an immediate
            pipe_fence();   // loop entry: hipcc's preheader copies of the loop-carried values (plain v_mov's) need wait states before the
                            // first asm MFMA reads them (catalog_x3.h, round 3: a stale fragment in the first MFMA of the first trip)
#define PCVAE_SP(UU)                                                                                                      \
            if constexpr (UU < TR) {                                                                                      \
                constexpr int TL = 1 + UU, TN = 2 + UU;                                                                   \
                constexpr int OL = ((TL / SUB) % NB) * CB + (TL % SUB) * G::ST;                                           \
                constexpr int ON = ((TN / SUB) % NB) * CB + (TN % SUB) * G::ST;                                           \
                constexpr bool SEAM = (TL % SUB) == SUB - 1;                                                              \
                ScreenSeam s2 = seam_of(t + UU);                                                                          \
                s2.stage_lds = lds0 + (((1 + UU) / SUB + 1 + PF) % NB) * CB;   /* t = 1 (mod TR): a constant */              \
                screen_pipe_logits<D, CT, PASS, OL, ON, 0, SEAM, (PF - 1) * 4, true, false>(                              \
                    p, cand, lds0, L.a0, af, xb, A2[TL & 1], A2[UU & 1], nbase + (int64_t)(t + UU - 1) * 32, st, g, s2,   \
                    wave_u, lane_off);                                                                                    \
            }
            PCVAE_SP(0) PCVAE_SP(1) PCVAE_SP(2) PCVAE_SP(3) PCVAE_SP(4) PCVAE_SP(5) PCVAE_SP(6) PCVAE_SP(7)
            PCVAE_SP(8) PCVAE_SP(9) PCVAE_SP(10) PCVAE_SP(11) PCVAE_SP(12) PCVAE_SP(13) PCVAE_SP(14) PCVAE_SP(15)
#undef PCVAE_SP
            pipe_fence();  // latch
        }
        for (; t < T; ++t) {   // the last slots: runtime ring offsets, fenced MFMAs
            ScreenSeam s2 = seam_of(t);
            s2.next_lbase = lds_of(t + 1 < T ? t + 1 : t);
            pipe_fence();
            if (t & 1) screen_pipe_logits<D, CT, PASS, 0, 0, 0, true, 0, true, true>(p, cand, lds_of(t), L.a0, af, xb, A2[1], A2[0], nbase + (int64_t)(t - 1) * 32, st, g, s2, wave_u, lane_off);
            else screen_pipe_logits<D, CT, PASS, 0, 0, 0, true, 0, true, true>(p, cand, lds_of(t), L.a0, af, xb, A2[0], A2[1], nbase + (int64_t)(t - 1) * 32, st, g, s2, wave_u, lane_off);
            pipe_fence();
        }
        pipe_fence();
        lgkm_wait<0>();
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {   // drain: the look at the last subtile
            if ((T - 1) & 1) screen_look<D, CT, PASS, false>(p, cand, A2[1], nbase + (int64_t)(T - 1) * 32, st, g, ct);
            else screen_look<D, CT, PASS, false>(p, cand, A2[0], nbase + (int64_t)(T - 1) * 32, st, g, ct);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- tail: short / ragged chunks, staged synchronously with clamped addresses, one subtile at a time
    for (int tt = t_beg + T; tt < t_end; tt += 4) {
        __syncthreads();
        fast_stage_tail<D, NW>(p.Eb, p.N, (int64_t)tt * 32, smem);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int nsub = min(4, t_end - tt);
        for (int sb = 0; sb < nsub; ++sb) {
            ScreenAcc<CT> a;
            pipe_fence();
            pipe2_cold_logits<D, CT>(lds0 + sb * G::ST, L.a0, af, xb, a.acc);
            pipe_fence();
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) screen_look<D, CT, PASS, true>(p, cand, a, (int64_t)(tt + sb) * 32, st, g, ct);
        }
    }
    if (PASS == 1) {  // rescore the parked candidates of the four waves' lists
        // One ENTRY per thread over all four lists at once (an entry = a lane's eight items with a bit per item that passed; nearly always
        // one bit): the ~10^2 entries of a workgroup are one round of exact chains.  (Until round 6: one ITEM SLOT per thread, list after
        // list - four dependent rounds of global round trips per workgroup with seven of eight threads idle; the order of the 64-bit
        // atomicMax folds does not matter, the ids are the same.)
        unsigned int* counts = reinterpret_cast<unsigned int*>(cand + screen_pipe_cap<D>() * SCREEN_PIPE_ENTRY_BYTES);
        if (lane == 0) counts[wave] = min(st.cnt, (unsigned)(screen_pipe_cap<D>() / 4));
        __syncthreads();
        const unsigned int c0 = counts[0], c1 = c0 + counts[1], c2 = c1 + counts[2], c3 = c2 + counts[3];
        for (unsigned int e = threadIdx.x; e < c3; e += 256) {
            const unsigned int w = e < c0 ? 0u : e < c1 ? 1u : e < c2 ? 2u : 3u;
            const unsigned int base = w == 0u ? 0u : w == 1u ? c0 : w == 2u ? c1 : c2;
            const uint2 en = (reinterpret_cast<const uint2*>(cand) + w * (screen_pipe_cap<D>() / 4))[e - base];
            const int64_t first = (int64_t)(en.y >> 8) << 2;
            unsigned int bits = en.y & 0xffu;
            while (bits) {
                const unsigned int b = (unsigned int)__builtin_ctz(bits);   // bit 4 rt + i <-> item n0 + 4 g + 16 rt + i
                bits &= bits - 1u;
                const int64_t n = first + 16 * ((b >> 2) & 1u) + (b & 3u);
                if (n < p.N) screen_rescore<D>(p, (int64_t)en.x, n);
            }
        }
    }
    if (PASS == 0) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            float v = st.m[ct];
            v = fmaxf(v, __shfl_xor(v, 16, 64));
            v = fmaxf(v, __shfl_xor(v, 32, 64));
            const int64_t r = rw + 16 * ct + c;
            if (g == 0 && r < p.R) p.pm[(int64_t)split * p.R + r] = v;
        }
    }
}

// after pass A: thr[r] = max_j pm[j][r] - 2 eps_r ; best_key[r] = 0.  pass A may have seen only a PREFIX of the catalog:
// any lower bound of the full approximate maximum m~ is a valid threshold base (more candidates, same answer).
__global__ void catalog_screen_threshold_kernel(ScreenParams p, int D) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= p.R) return;
    float mm = -INFINITY;
    for (int j = 0; j < p.nsplit; ++j) mm = fmaxf(mm, p.pm[(int64_t)j * p.R + r]);
    float ss = 0.f;
    for (int k = 0; k < D; ++k) ss = fmaf(p.x[r * D + k], p.x[r * D + k], ss);
    // two RNE roundings per product (2^-9 each), fp32 accumulation slack growing with the chain length
    const float eps = (0.00390625f * 1.02f + 2e-5f * (float)(D > 128 ? D / 128 : 1)) * sqrtf(ss) * p.e_max_norm;
#ifdef SCREEN_PROBE_NO_CAND   // probe builds only (tools/screen_loop_probe.sh): no item ever passes - pass B without its candidate handling (ids garbage)
    p.thr[r] = INFINITY;
#else
    p.thr[r] = mm - 2.f * eps - 1e-30f;
#endif
    p.eps2[r] = 2.f * eps + 1e-30f;
    p.best_key[r] = 0ull;
    if (r == 0) *p.overflow = 0u;
}

__global__ void catalog_screen_decode_kernel(ScreenParams p, int64_t* __restrict__ idx, float* __restrict__ best) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= p.R) return;
    const unsigned long long key = p.best_key[r];
#ifdef SCREEN_PROBE_NO_CAND   // (probe builds: no candidate was folded, the key is empty - a VALID id, the callers gather with it)
    idx[r] = 0;
#else
    idx[r] = (int64_t)(0xffffffffu - (unsigned int)(key & 0xffffffffull));
#endif
    if (best) best[r] = unordered_bits((unsigned int)(key >> 32));
}

#include "catalog_x3.h"

}  // namespace

namespace pcvae {

template <int D, int NC>
static int launch_ce_x3(const float* rx, int64_t R, const uint16_t* Ex, const float* Ef, int64_t N, float e_max_norm,
                        const int64_t* target, float* nll, float* lse, float* dx, float dx_scale, void* ws, hipStream_t st) {
    constexpr int CT = x3_ct(D);
    using XG = X3Geo<D, CT, NC>;
    const CatalogPlan pl = catalog_plan(R, N, D, NC == 3 ? PCVAE_PREC_BF16X6 : PCVAE_PREC_BF16X3);
    CatParamsB p{};
    p.rx = rx; p.E = Ex; p.target = target; p.R = R; p.N = N; p.dx_scale = dx_scale;
    p.nrb = pl.nrb; p.nsplit = pl.nsplit; p.tiles_per_split = pl.tiles_per_split; p.ntiles = pl.ntiles;
    const int64_t ns = pl.nsplit;
    p.pm = reinterpret_cast<float*>(ws);
    p.pl = p.pm + ns * R;
    p.pU = p.pl + ns * R;
    uint8_t* flags = reinterpret_cast<uint8_t*>(p.pU + ns * R * D);   // [nrb] behind the partials
    // row blocks whose logit bound allows raw exp2 run the max-free split-bf16 kernel; the others (flag 1) the exact f32 kernel
    hipLaunchKernelGGL((catalog_row_bound_kernel<D>), dim3((unsigned)p.nrb), dim3(256), 0, st, rx, R, e_max_norm, flags);
    p.safe_flags = flags;
    constexpr int lds_pipe = XG::NB * XG::CB;
    if (int rc_optin = lds_optin(reinterpret_cast<const void*>(&catalog_ce_x3_pipe_kernel<D, CT, NC>), lds_pipe)) return rc_optin;
    const dim3 grid((unsigned)(cdiv(R, XG::ROWS) * p.nsplit));
    hipLaunchKernelGGL((catalog_ce_x3_pipe_kernel<D, CT, NC>), grid, dim3(256), lds_pipe, st, p);
    int rc = check_launch(NC == 3 ? "catalog_ce_x6" : "catalog_ce_x3");
    if (rc != PCVAE_OK) return rc;
    hipLaunchKernelGGL((catalog_ce_merge_x3_kernel<D>), dim3((unsigned)cdiv(R, 4)), dim3(256), 0, st, p, Ef, nll, lse, dx);
    rc = check_launch("catalog_ce_merge_x3");
    if (rc != PCVAE_OK) return rc;
    // flagged row blocks (normally none): the exact f32 kernel on its own partials behind the flags, rows of flag 1 only
    char* ws2 = reinterpret_cast<char*>(flags) + ((p.nrb + 255) / 256) * 256;
    return catalog_ce_f32_flagged(rx, R, Ef, N, D, target, nll, lse, dx, dx_scale, ws2, flags, st);
}

// Ex: the split-bf16 table image.  ncomp = 2 (bf16x3, pcvae_split_bf16x2): D / 128 images of [N, 256] bf16, image i = c0 | c1 of
// dims 128 i .. 128 i + 127.  ncomp = 3 (bf16x6, pcvae_split_bf16x3): [N, 384] bf16, row n = c0 | c1 | c2 of E_n (D = 128)
int catalog_ce_x3(const float* rx, int64_t R, const uint16_t* Ex, const float* Ef, int64_t N, int D, int ncomp, float e_max_norm,
                  const int64_t* target, float* nll, float* lse, float* dx, float dx_scale, void* ws, hipStream_t st) {
    if (ncomp == 2 && D == 128) return launch_ce_x3<128, 2>(rx, R, Ex, Ef, N, e_max_norm, target, nll, lse, dx, dx_scale, ws, st);
    if (ncomp == 2 && D == 256) return launch_ce_x3<256, 2>(rx, R, Ex, Ef, N, e_max_norm, target, nll, lse, dx, dx_scale, ws, st);
    if (ncomp == 3 && D == 128) return launch_ce_x3<128, 3>(rx, R, Ex, Ef, N, e_max_norm, target, nll, lse, dx, dx_scale, ws, st);
    set_error("catalog_ce(bf16x%d): D=%d (bf16x3 exists for D = 128 and 256, bf16x6 for D = 128)", ncomp == 3 ? 6 : 3, D);
    return PCVAE_EINVAL;
}

int catalog_ce_bf16(const float* rx, int64_t R, const uint16_t* E, int64_t N, int D, float e_max_norm,
                    const int64_t* target, float keep_prob, uint64_t seed, uint64_t row_offset,
                    const uint8_t* keep_mask, float* nll, float* lse, float* dx, float dx_scale, void* ws, hipStream_t st) {
    const CatalogPlan pl = catalog_plan(R, N, D, PCVAE_PREC_BF16);
    CatParamsB p{};
    p.dx_scale = dx_scale;
    p.rx = rx; p.E = E; p.target = target; p.keep = keep_mask;
    p.seed = seed; p.row_offset = row_offset; p.R = R; p.N = N;
    p.nrb = pl.nrb; p.nsplit = pl.nsplit; p.tiles_per_split = pl.tiles_per_split; p.ntiles = pl.ntiles;
    const int64_t nsplit_ws = pl.nsplit;
    p.pm = reinterpret_cast<float*>(ws);
    p.pl = p.pm + nsplit_ws * R;
    p.pU = p.pl + nsplit_ws * R;
    // row-block flags live behind the partials (pcvae_catalog_ws_bytes reserves them)
    uint8_t* flags = reinterpret_cast<uint8_t*>(p.pU + (dx ? nsplit_ws * R * D : 0));
    int mask_mode = MASK_NONE;
    if (keep_mask) mask_mode = MASK_BYTES;
    else if (keep_prob < 1.0f) {
        mask_mode = MASK_PHILOX;
        const double th = (double)keep_prob * 4294967296.0;
        p.keep_thresh = th <= 0.0 ? 0u : (th >= 4294967295.0 ? 0xffffffffu : (uint32_t)th);
    }
    switch (D) {
        case 64: return launch_ce_b<64>(p, mask_mode, dx != nullptr, e_max_norm, flags, nll, lse, dx, st);
        case 128: return launch_ce_b<128>(p, mask_mode, dx != nullptr, e_max_norm, flags, nll, lse, dx, st);
        case 256: return launch_ce_b<256>(p, mask_mode, dx != nullptr, e_max_norm, flags, nll, lse, dx, st);
    }
    set_error("catalog_ce(bf16): unsupported D=%d (64, 128, 256; smaller tables use the f32 kernel)", D);
    return PCVAE_EINVAL;
}

// ranges of at least this many 32-item tiles take the software-pipelined screening kernels.  PCVAE_PIPE_MIN_TILES (read per launch) moves
// the threshold: the tests force the pipelined kernels onto small shapes with it.
// (D = 64, round 6: from 128 tiles per range - the quarter-catalog prefix pass of config 3's generate step takes the pipelined
// kernel too: 1.09 -> 1.05 ms per batch, same ids; then, with two workgroups per CU, from 96: the pivot stage's 4096 rows plan 512
// workgroups of 98 tiles - 70 -> 54 us; sweep 128 / 96 / 64 / 48: 0.770 / 0.753 / 0.762 / 0.760 ms per batch)
static int screen_pipe_min(int D) {
    const char* env_min = getenv("PCVAE_PIPE_MIN_TILES");
    return env_min ? atoi(env_min) : (D == 64 ? 96 : 512);
}

template <int D>
static int launch_screened(ScreenParams p, const CatalogPlan& pa, const CatalogPlan& pb, int64_t Ns, int64_t N, int64_t* idx,
                           float* best, int wgpc, hipStream_t st) {
    const size_t lds = SCREEN_LDS_BYTES;
    if (int rc_optin = lds_optin(reinterpret_cast<const void*>(&catalog_screen_bf16_kernel<D, 0>), (int)lds)) return rc_optin;
    if (int rc_optin = lds_optin(reinterpret_cast<const void*>(&catalog_screen_bf16_kernel<D, 1>), (int)lds)) return rc_optin;
    const dim3 block(512);
    // long ranges: the software-pipelined screening kernels (one wave per SIMD, 64 rows per wave for D <= 128)
    constexpr int CT = D == 256 ? 2 : 4;
    using PG = PipeGeo<D, CT>;
    // what the kernel uses; planned for one workgroup per CU it ASKS for 96 KB, so that the hardware does not co-schedule two anyway
    constexpr int lds_used = PG::NB * 16384 + screen_pipe_cap<D>() * SCREEN_PIPE_ENTRY_BYTES + 16;
    constexpr int lds_one = lds_used > 96 * 1024 ? lds_used : 96 * 1024;
    const int lds_pipe = wgpc >= 2 ? lds_used : lds_one;
    // (the opt-in is remembered per kernel, not per size: ask for the larger of the two once)
    if (int rc_optin = lds_optin(reinterpret_cast<const void*>(&catalog_screen_pipe_kernel<D, CT, 0>), lds_one)) return rc_optin;
    if (int rc_optin = lds_optin(reinterpret_cast<const void*>(&catalog_screen_pipe_kernel<D, CT, 1>), lds_one)) return rc_optin;
    const int pipe_min = N < SCREEN_PIPE_MAX_ITEMS ? screen_pipe_min(D) : INT_MAX;   // (the pipelined kernels' entries hold item / 4 in 24 bits)
    p.N = Ns; p.nrb = pa.nrb; p.nsplit = pa.nsplit; p.tiles_per_split = pa.tiles_per_split; p.ntiles = pa.ntiles;
    if (pa.tiles_per_split >= pipe_min)
        hipLaunchKernelGGL((catalog_screen_pipe_kernel<D, CT, 0>), dim3((unsigned)(cdiv(p.R, PG::ROWS) * pa.nsplit)), dim3(256),
                           lds_pipe, st, p);
    else
        hipLaunchKernelGGL((catalog_screen_bf16_kernel<D, 0>), dim3((unsigned)(pa.nrb * pa.nsplit)), block, lds, st, p);
    hipLaunchKernelGGL(catalog_screen_threshold_kernel, dim3((unsigned)cdiv(p.R, 256)), dim3(256), 0, st, p, D);
    p.N = N; p.nrb = pb.nrb; p.nsplit = pb.nsplit; p.tiles_per_split = pb.tiles_per_split; p.ntiles = pb.ntiles;
    if (pb.tiles_per_split >= pipe_min) {
        hipLaunchKernelGGL((catalog_screen_pipe_kernel<D, CT, 1>), dim3((unsigned)(cdiv(p.R, PG::ROWS) * pb.nsplit)), dim3(256),
                           lds_pipe, st, p);
        p.only_if_overflow = 1;   // exact redo, a no-op unless a candidate list overflowed
    }
    hipLaunchKernelGGL((catalog_screen_bf16_kernel<D, 1>), dim3((unsigned)(pb.nrb * pb.nsplit)), block, lds, st, p);
    hipLaunchKernelGGL(catalog_screen_decode_kernel, dim3((unsigned)cdiv(p.R, 256)), dim3(256), 0, st, p, idx, best);
    return check_launch("catalog_argmax_screened");
}

int catalog_argmax_screened(const float* x, int64_t R, const uint16_t* Eb, const float* Ef, int64_t N, int D, float e_max_norm,
                            int64_t* idx, float* best, void* ws, hipStream_t st) {
    // pass A over a prefix of the catalog is enough to seed the threshold: with N/16 items the expected number of
    // later items above the prefix maximum is ~16 per row (plus the lane-local running maximum in pass B), each
    // costing one D-term fmaf chain - far cheaper than a second full bf16 pass.
    // (Round 6: a prefix from N = 65536 up.  Until round 5 catalogs under 262144 items ran pass A over the WHOLE catalog - config 3's
    // generate step, N = 10^5, paid two full bf16 passes: 448 + 594 us of a 1.36 ms batch in the kernel trace,
    // profiles/r06_config3_generate_kernel_stats_before.csv.  The candidates per row, ~N / Ns, cost pass B rescoring time, and
    // relatively more so on a SHORT catalog pass: below 262144 items the prefix is N / 4 (sweep: profiles/r06_screen_prefix_sweep.txt).
    // PCVAE_SCREEN_PREFIX_MIN_ITEMS / PCVAE_SCREEN_PREFIX_DIV / PCVAE_SCREEN_PREFIX_DIV_LARGE move the switch / the short catalogs' divisor /
    // the long catalogs' divisor for A/B measurements.)
    static const int64_t prefix_min = [] { const char* e = getenv("PCVAE_SCREEN_PREFIX_MIN_ITEMS"); return e ? atoll(e) : 65536LL; }();
    static const int64_t prefix_div = [] { const char* e = getenv("PCVAE_SCREEN_PREFIX_DIV"); return e && atoll(e) > 0 ? atoll(e) : 4LL; }();
    // (long catalogs, config 4, round 6: N / 8 17.76, N / 12 17.35, N / 16 17.3, N / 24 17.2, N / 32 17.1-17.25 ms per batch - flat; 16 stays)
    static const int64_t prefix_div_large = [] { const char* e = getenv("PCVAE_SCREEN_PREFIX_DIV_LARGE"); return e && atoll(e) > 0 ? atoll(e) : 16LL; }();
    const int64_t Ns = N >= 262144 ? (N / prefix_div_large) / 128 * 128 : N >= prefix_min ? (N / prefix_div) / 128 * 128 : N;
    // the screen kernels always run 256-row workgroups: plan them as the D = 128 case
    // Two workgroups per CU at D = 64: a lone wave per SIMD issues one instruction per ~5.5 cycles and an MFMA per 17.5, two waves share
    // the SIMD at ~2.7 and 16.5 (tools/valu_rate_probe.hip) - and a D = 64 screening slot is mostly bookkeeping (16 MFMAs + 41 other
    // instructions).  The registers allow it (<= 202 VGPRs), the candidate list was what did not fit.
    // At D = 128 a slot is 32 MFMAs under the same bookkeeping and two workgroups per CU bought nothing (config 4: 17.19 vs 17.35 ms per
    // batch, tools/screen_wg_ab.sh): one, with the long candidate list.  PCVAE_SCREEN_WG_PER_CU=1 (read once) plans D = 64 for one as well
    // and asks for 96 KB of LDS so that the hardware does not co-schedule two anyway (A/B measurements).
    static const int wg_env = [] { const char* e = getenv("PCVAE_SCREEN_WG_PER_CU"); return e ? atoi(e) : 0; }();
    const int wgpc = (D == 64 && wg_env != 1) ? 2 : 1;
    // (few rows: twice the ranges can push a range under the pipelined kernels' minimum - then the plan for one workgroup per CU, whose
    // ranges are long enough, is the better one: config 3's pivot stage, R = 4096, 68 vs 97 us)
    const int pipe_min = N < SCREEN_PIPE_MAX_ITEMS ? screen_pipe_min(D) : INT_MAX;
    auto plan = [&](int64_t n) {
        const CatalogPlan p2 = catalog_plan(R, n, 128, PCVAE_PREC_BF16, wgpc);
        if (wgpc > 1 && p2.tiles_per_split < pipe_min) {
            const CatalogPlan p1 = catalog_plan(R, n, 128, PCVAE_PREC_BF16, 1);
            if (p1.tiles_per_split >= pipe_min) return p1;
        }
        return p2;
    };
    const CatalogPlan pa = plan(Ns), pb = plan(N);
    ScreenParams p{};
    p.x = x; p.Eb = Eb; p.Ef = Ef; p.R = R; p.e_max_norm = e_max_norm;
    p.best_key = reinterpret_cast<unsigned long long*>(ws);  // 8-byte atomics: keep first (ws is 16-byte aligned)
    p.thr = reinterpret_cast<float*>(p.best_key + R);
    p.eps2 = p.thr + R;
    p.pm = p.eps2 + R;   // [nsplit(A) <= 64][R]
    p.overflow = reinterpret_cast<unsigned int*>(p.pm + 64 * R);   // (pcvae_catalog_ws_bytes reserves 64 bytes behind pm)
    switch (D) {
        case 64: return launch_screened<64>(p, pa, pb, Ns, N, idx, best, wgpc, st);
        case 128: return launch_screened<128>(p, pa, pb, Ns, N, idx, best, wgpc, st);
        case 256: return launch_screened<256>(p, pa, pb, Ns, N, idx, best, wgpc, st);
    }
    set_error("catalog_argmax(screened): unsupported D=%d (64, 128, 256)", D);
    return PCVAE_EINVAL;
}

}  // namespace pcvae
