// K3: the MLP stacks of the PivotCVAE (encoder / pivot-selection / slate-completion / prior; reference
// models/pivotcvae.py:159-240) as LDS-tiled fp32 MFMA GEMMs on gfx950: bias + LeakyReLU fused into the forward epilogue, the
// LeakyReLU derivative into the input-gradient GEMM, the bias gradient into the (batch-split) weight-gradient GEMM.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate, exact fmaf chain), because the parity contract is 1e-4
// relative against an fp32 reference.  One 256-thread workgroup = 4 waves (2 x 2), each wave owns one 32 x 32 accumulator
// tile of a 64 x 64 output tile.  One kernel body covers the three layouts of a Linear layer:
//    forward      Y[M,N]  = X[M,K]  . W[N,K]^T    A k-contiguous,   B k-contiguous
//    input grad   dX[M,K] = dY[M,N] . W[N,K]      A k-contiguous,   B row-contiguous
//    weight grad  dW[N,K] = dY[M,N]^T . X[M,K]    A row-contiguous, B row-contiguous, split over the batch
//
// How the operands reach LDS (round 2; the round-1 kernel staged tiles through registers: 32 ds_write_b32 per thread between
// two barriers per K chunk, and with the global loads removed the [8192 x 256 x 1419] layer still took 57 us of an ideal 38):
//   * tiles stream global -> LDS by LDS-DMA (global_load_lds_dword: no staging registers, no ds_write, no store phase), two
//     stages of 32 k, ONE barrier per chunk; 32 KB of LDS and ~110 VGPRs leave four workgroups per CU (a deeper ring was
//     measured: it costs occupancy and is slower on every layer of the model);
//   * each lane's DMA source address is free, so the LDS image is whatever the MFMA operand fetch wants, with no padding:
//       k-contiguous operand  -> image [64 rows][32 k] (128-byte rows), 16-byte chunks XOR-swizzled by (row >> 1) & 7; with the
//                                k order  k(g, h, j) = 8 g + 4 h + j  (g = 0..3, h = lane >> 5, j = 0..3) a lane's four
//                                consecutive k-steps are ONE conflict-free ds_read_b128 (any bijection of k is a valid order
//                                as long as both operands use it);
//       row-contiguous operand -> image [32 k][64 rows] (256-byte lines), the lines of the h = 1 half (k bit 2) rotated by 32
//                                rows, so the two lane halves of a ds_read_b32 hit disjoint banks;
//   * ragged K: the last, partial chunk takes per-lane 64-bit addresses, lanes past the end read a zero page; ragged M / N:
//     row indices are clamped (those rows are computed and never stored).
//
// Grouped launches: INDEPENDENT GEMMs (a layer's weight- and input-gradient; the same layer of the encoder and of the prior
// stack) are one launch - every kernel boundary costs a drain + ramp of 3-4 us, a 25-GEMM step spent ~100 us there, and the
// prior's small layers (256 workgroups, half a chip) now ride along with the encoder's.  A group's workgroups are dealt to
// the XCDs so that each XCD holds a CONTIGUOUS run of every problem's tiles (column tiles of one row block share the A rows
// in one private L2; all tiles of one batch split of a weight gradient share both operands).
#include "common.h"
#include <cstdint>
#include <cstdlib>
#include <algorithm>

using namespace pcvae;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {

// two fp32 values -> their bf16 roundings (RNE) in one register, element 0 in the low half (v_cvt_pk_bf16_f32); and back, exactly
__device__ __forceinline__ unsigned pack_bf16x2(const f32x2 x) {
    const bf16x2 h = __builtin_convertvector(x, bf16x2);
    return *reinterpret_cast<const unsigned*>(&h);
}
__device__ __forceinline__ f32x2 widen_bf16x2(const unsigned w) {
    f32x2 r;
    r[0] = __uint_as_float(w << 16);
    r[1] = __uint_as_float(w & 0xffff0000u);
    return r;
}

constexpr int BM = 64, BN = 64;                                       // output tile of the DMA body
constexpr int DK = 32, NSTAGE = 2, STAGE_BYTES = 2 * 64 * DK * 4, PCS = 8;   // PCS: 256-byte pieces per wave per operand per chunk
constexpr int SM = 32, SBK = 64, SLD = 33, STPT = SM * SBK / 256;     // small-tile body (below)
constexpr int SMALL_LDS = (2 * SBK * SLD + 4 * SM * SM) * 4;
constexpr int MAXG = 6;

enum { EPI_FWD = 0, EPI_DX = 1, EPI_DW = 2 };
enum { KIND_FWD = 0, KIND_DX = 1, KIND_DW = 2, KIND_FWD_S = 3, KIND_DX_S = 4 };

struct GemmParams {
    const float* A; int64_t lda;   // logical A(m, k)
    const float* B; int64_t ldb;   // logical B(n, k)
    float* C; int64_t ldc;         // C(m, n)
    int64_t M, N, K;
    const float* bias;             // EPI_FWD: [N] or null
    float* bias_grad;              // EPI_DW: [M of this GEMM = layer outputs] or null: += sum over the reduction index
    const float* aux; int64_t ldaux;  // EPI_DX: activated input [M,N] or null
    int act;
    int accumulate;                // EPI_DX: C = (C + A.B) * act'(aux) - the second of two layers that share an input
    int64_t k_per_split;           // EPI_DW: reduction range per z
    int kind, nx, ny, nz, slots;   // tile grid of this problem; slots = cdiv(nx ny nz, 8): workgroups it takes on each XCD
    // EPI_DW with a workspace: the nz batch splits of an output tile store their partial tiles (and bias partials), the LAST one to
    // arrive (a counter per tile, self-resetting) sums them in split order and adds the sum to C - no atomics on C, bitwise
    // reproducible.  Null: the problem runs as ONE split.
    float* ws_part;                // [ny nx][nz][16][256]
    float* ws_bias;                // [ny][nz][64]
    unsigned* ws_cnt;              // [ny nx], zero between launches
};

struct GroupParams {
    GemmParams g[MAXG];
    int n;
    int dma_x4;   // 16-byte LDS-DMA where the operands allow it (PCVAE_GEMM_DMA16, default on)
};

__device__ float g_zero_page[64];

// source offset (floats, relative to the tile origin P(row0, k0)) of the 4 bytes that land at LDS position (piece pc, lane)
template <bool KC>
__device__ __forceinline__ int64_t dma_src(int pc, int lane, int64_t ld, int rows_left, int* k_of_lane) {
    if (KC) {   // piece = 2 image rows of 32 k
        const int row = 2 * pc + (lane >> 5), f = lane & 31;
        const int k = 4 * ((f >> 2) ^ ((row >> 1) & 7)) + (f & 3);
        *k_of_lane = k;
        const int rc = row < rows_left ? row : rows_left - 1;
        return (int64_t)rc * ld + k;
    } else {    // piece = one k line of 64 rows
        const int k = pc, row = lane ^ (((k >> 2) & 1) << 5);
        *k_of_lane = k;
        const int rc = row < rows_left ? row : rows_left - 1;
        return (int64_t)k * ld + rc;
    }
}

// A wave's 8 pieces of one operand: the instruction's immediate offset steps BOTH the LDS and the global address, so a piece's
// lane offset is its source offset minus that step (plus a 2 KB bias folded into the scalar base to keep it non-negative).
// Invisible to hipcc's vmcnt bookkeeping: the loop waits by hand.  (m0 on the clobber list: stated, not assumed - hipcc never
// keeps a value in M0 across statements.)
__device__ __forceinline__ void dma8(const float* base, const int (&off)[PCS], unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %9\n\ts_nop 0\n\t"
                 "global_load_lds_dword %0, %8\n\t"
                 "global_load_lds_dword %1, %8 offset:256\n\t"
                 "global_load_lds_dword %2, %8 offset:512\n\t"
                 "global_load_lds_dword %3, %8 offset:768\n\t"
                 "global_load_lds_dword %4, %8 offset:1024\n\t"
                 "global_load_lds_dword %5, %8 offset:1280\n\t"
                 "global_load_lds_dword %6, %8 offset:1536\n\t"
                 "global_load_lds_dword %7, %8 offset:1792"
                 ::"v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]), "v"(off[7]),
                   "s"(base), "s"(lds_dst)
                 : "memory", "m0");
}

// 16-byte LDS-DMA (round 6): the same LDS images filled by global_load_lds_dwordx4 - a lane brings one 16-byte chunk (4 consecutive k
// of a k-contiguous row, or 4 consecutive rows of a k line), a wave 1 KB per instruction, so an operand's 8 KB chunk is 2 instructions
// per wave instead of 8.  The image is unchanged (the swizzles above act on whole 16-byte chunks), so results are bitwise the
// dword path's.  Source offset (floats, relative to the tile origin) of the 16 bytes that land at LDS position (piece pc of 1 KB, lane):
template <bool KC>
__device__ __forceinline__ int64_t dma_src4(int pc, int lane, int64_t ld, int rows_left) {
    if (KC) {   // piece = 8 image rows of 32 k
        const int row = 8 * pc + (lane >> 3), slot = lane & 7;
        const int k = 4 * (slot ^ ((row >> 1) & 7));
        const int rc = row < rows_left ? row : rows_left - 1;
        return (int64_t)rc * ld + k;
    } else {    // piece = 4 k lines of 64 rows (only used when all 64 rows exist: a chunk of 4 rows cannot be clamped row by row)
        const int k = 4 * pc + (lane >> 4), slot = lane & 15;
        const int row = 4 * (slot ^ (((k >> 2) & 1) << 3));
        return (int64_t)k * ld + row;
    }
}
__device__ __forceinline__ void dma2x4(const float* base, const int (&off)[2], unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024"
                 ::"v"(off[0]), "v"(off[1]), "s"(base), "s"(lds_dst)
                 : "memory", "m0");
}

__device__ __forceinline__ void dma1_addr(const float* src, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(lds_dst) : "memory", "m0");
}

template <int EPI>
__device__ __forceinline__ void store_c(const GemmParams& p, int64_t m, int64_t n, float v, float bias) {
    if (EPI == EPI_FWD) {
        v += bias;
        if (p.act == PCVAE_ACT_LEAKY) v = leaky(v);
        else if (p.act == PCVAE_ACT_RELU) v = fmaxf(v, 0.f);
        p.C[m * p.ldc + n] = v;
    } else if (EPI == EPI_DX) {
        if (p.accumulate) v += p.C[m * p.ldc + n];
        if (p.aux && !(p.aux[m * p.ldaux + n] > 0.f)) v *= kLeakySlope;
        p.C[m * p.ldc + n] = v;
    }
}

#ifdef GEMM_STAMPS   // probe builds only: shader-clock stamps of workgroup 0 (slots 0..31; the weight gradient's last arriver of tile 0: slots 32..)
__device__ unsigned long long g_gemm_stamps[64];
#define GSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_gemm_stamps[i] = clock64(); } while (0)
#define GSTAMP_IF(c, i) do { if ((c) && threadIdx.x == 0) g_gemm_stamps[i] = clock64(); } while (0)
extern "C" int pcvae_gemm_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_stamps), sizeof(g_gemm_stamps)); }
#else
#define GSTAMP(i) do { } while (0)
#define GSTAMP_IF(c, i) do { } while (0)
#endif

// ---- 64 x 64 tile, LDS-DMA --------------------------------------------------------------------------------------------------
// X3 (round 3): the same tile, staging and epilogues, but the contraction runs on the bf16 matrix cores at fp32-EQUIVALENT precision:
// every operand value v is split in registers into hi = RNE bf16(v), lo = RNE bf16(v - hi) and a product is three
// v_mfma_f32_16x16x32_bf16 (hi*hi + lo*hi + hi*lo, fp32 accumulate; lo*lo dropped: 2^-18 relative) - the catalog kernel's bf16x3
// arithmetic (catalog_x3.h) applied to the MLP stacks.  The fp32 LDS images are the SAME ones: with the k order
// kset(g) = {4g .. 4g+3} u {16+4g .. 16+4g+3} (g = lane >> 4; any bijection of k is a valid order as long as both operands use
// it) a lane's eight k values of a k-contiguous image are the two 16-byte chunks g and g+4 of its row, and the XOR swizzle
// (row >> 1) & 7 makes those ds_read_b128 conflict-free for 16-row operand tiles as well (searched over the guide's b128 service
// groups); a row-contiguous image is read with 8 ds_read_b32 whose two 32-lane halves hit disjoint bank halves (the 32-row
// rotation of the lines with k bit 2 set).  A wave's 32 x 32 tile = 2 x 2 MFMA tiles of 16 x 16: 12 MFMAs of 16 cycles per
// 32-deep chunk against 16 of 64 cycles for v_mfma_f32_32x32x2_f32 - the loop turns from matrix-pipe-bound to VALU-bound (the
// splits: ~96 vector ops per chunk), about 1.75x faster on the large layers.  (Measured and dropped, round 3: splitting each staged tile
// ONCE per workgroup into bf16 LDS images - half the vector ops, plain 16-byte operand reads - needs a second barrier per chunk and
// 16 KB more LDS (three workgroups per CU instead of five): enc_1 forward 49.0 us against 36.8 us for the in-register split.)
// X6 (round 6, XM = 2): the same body with every operand value as THREE bf16 components, c0 = RNE bf16(v), c1 = RNE bf16(v - c0),
// c2 = v - c0 - c1 (exactly representable: 3 x 8 significand bits hold the 24 of an fp32), and SIX MFMAs per product - c2 c0, c0 c2,
// c1 c1, c1 c0, c0 c1, c0 c0, smallest first; the dropped c1 c2, c2 c1, c2 c2 are <= 2^-25 relative, below the rounding of an fp32
// product - every partial product exact in the fp32 accumulator's input: the reference's fp32 arithmetic on the bf16 matrix cores
// (the catalog kernel's bf16x6, catalog_x3.h, applied to K3).  XM: 0 exact f32 MFMA, 1 bf16x3, 2 bf16x6.
template <bool A_KC, bool B_KC, int EPI, int XM>
__device__ __forceinline__ void gemm_tile_dma(const GemmParams& p, const int bx, const int by, const int bz, char* smem, const bool dma_x4) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, h = lane >> 5;
    constexpr bool X3 = XM != 0;                 // (the name of the round-3 code: "runs on the bf16 pipe, 16 x 16 x 32 tiles")
    const int c16 = lane & 15, gq = lane >> 4;   // X3 / X6: MFMA 16x16x32 lane coordinates
    const int64_t m0 = (int64_t)by * BM, n0 = (int64_t)bx * BN;

    int64_t kbeg = 0, kend = p.K;
    if (EPI == EPI_DW) {
        kbeg = (int64_t)bz * p.k_per_split;
        kend = kbeg + p.k_per_split < p.K ? kbeg + p.k_per_split : p.K;
    }
    const int nch = (int)((kend - kbeg + DK - 1) / DK);
    const bool ragged = ((kend - kbeg) % DK) != 0;

    // one accumulator of 16 floats per lane either way: a 32 x 32 MFMA tile, or 2 x 2 tiles of 16 x 16 (acc4[rt][ct])
    f32x16 acc;
    f32x4 acc4[2][2];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc4[0][0][i] = 0.f; acc4[0][1][i] = 0.f; acc4[1][0][i] = 0.f; acc4[1][1][i] = 0.f; }
    // EPI_DW: A(m, k) = dY[k][m]; the column sums of dY (bias gradient) are the k-sums of the A images this workgroup streams
    // anyway.  Only the workgroups of the first output column do it, 64 threads each.
    const bool do_bias = (EPI == EPI_DW) && p.bias_grad != nullptr && bx == 0 && threadIdx.x < BM;
    float bsum = 0.f;

    // tile origins, per-chunk steps (floats) and the lane offsets of this wave's 8 pieces per operand (bytes, + 2048 - 256 i)
    const int rowsA = (int)(p.M - m0 < BM ? p.M - m0 : BM), rowsB = (int)(p.N - n0 < BN ? p.N - n0 : BN);
    const float* A0 = A_KC ? p.A + m0 * p.lda + kbeg : p.A + kbeg * p.lda + m0;
    const float* B0 = B_KC ? p.B + n0 * p.ldb + kbeg : p.B + kbeg * p.ldb + n0;
    const int64_t stepA = A_KC ? DK : DK * p.lda, stepB = B_KC ? DK : DK * p.ldb;
    int offA[PCS], offB[PCS];
#pragma unroll
    for (int i = 0; i < PCS; ++i) {
        int kk;
        offA[i] = (int)(dma_src<A_KC>(wave * PCS + i, lane, p.lda, rowsA, &kk) * 4) + 2048 - 256 * i;
        offB[i] = (int)(dma_src<B_KC>(wave * PCS + i, lane, p.ldb, rowsB, &kk) * 4) + 2048 - 256 * i;
    }
    // 16-byte DMA where the operand allows it (workgroup-uniform): k-contiguous operands always (a chunk lies inside one row, rows are
    // clamped whole); row-contiguous operands only when all 64 rows of the tile exist.  Lane offsets: bytes, + 1024 - 1024 i.
    const bool a4 = dma_x4 && (A_KC || rowsA == BM), b4 = dma_x4 && (B_KC || rowsB == BN);
    int offA4[2], offB4[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        offA4[i] = (int)(dma_src4<A_KC>(wave * 2 + i, lane, p.lda, rowsA) * 4) + 1024 - 1024 * i;
        offB4[i] = (int)(dma_src4<B_KC>(wave * 2 + i, lane, p.ldb, rowsB) * 4) + 1024 - 1024 * i;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;   // LDS byte address of the ring

    auto issue = [&](int c) {   // chunk c -> stage c % NSTAGE (A image, then B image); uniform branch on the ragged last chunk
        const unsigned dstA = lds0 + (unsigned)(c % NSTAGE) * STAGE_BYTES + (unsigned)wave * 2048u, dstB = dstA + 64 * DK * 4;
        if (c == nch - 1 && ragged) {
            const int64_t k0 = kbeg + (int64_t)c * DK;
#pragma unroll 1   // once per workgroup: rolled, or its sixteen 64-bit addresses set the register count of the whole kernel
            for (int i = 0; i < PCS; ++i) {
                int ka, kb;
                const int64_t sa = dma_src<A_KC>(wave * PCS + i, lane, p.lda, rowsA, &ka);
                const int64_t sb = dma_src<B_KC>(wave * PCS + i, lane, p.ldb, rowsB, &kb);
                const float* pa = k0 + ka < kend ? A0 + (int64_t)c * stepA + sa : g_zero_page + lane;
                const float* pb = k0 + kb < kend ? B0 + (int64_t)c * stepB + sb : g_zero_page + lane;
                dma1_addr(pa, dstA + 256u * i);
                dma1_addr(pb, dstB + 256u * i);
            }
        } else {
            if (a4) dma2x4(reinterpret_cast<const float*>(reinterpret_cast<const char*>(A0 + (int64_t)c * stepA) - 1024), offA4, dstA);
            else dma8(reinterpret_cast<const float*>(reinterpret_cast<const char*>(A0 + (int64_t)c * stepA) - 2048), offA, dstA);
            if (b4) dma2x4(reinterpret_cast<const float*>(reinterpret_cast<const char*>(B0 + (int64_t)c * stepB) - 1024), offB4, dstB);
            else dma8(reinterpret_cast<const float*>(reinterpret_cast<const char*>(B0 + (int64_t)c * stepB) - 2048), offB, dstB);
        }
    };

    if (EPI == EPI_DW) GSTAMP(0);
    // (four stages for weight-gradient launches of one workgroup per CU were measured, round 3: the K loop stayed at 0.93 us per
    // 32-deep chunk - it is bound by the 32 ds_read_b32 + 16 MFMAs of a row-contiguous chunk, not by the chunk's round trip)
    if (nch > 0) issue(0);

    // operand fetch addresses inside a stage (bytes)
    const int rowA = wm * 32 + li, rowB = wn * 32 + li;
    const unsigned rdA = A_KC ? (unsigned)(rowA * 128) : (unsigned)((4 * h) * 256 + ((rowA ^ (h << 5)) << 2));
    const unsigned rdB = B_KC ? (unsigned)(rowB * 128) : (unsigned)((4 * h) * 256 + ((rowB ^ (h << 5)) << 2));
    const unsigned swA = (unsigned)((rowA >> 1) & 7), swB = (unsigned)((rowB >> 1) & 7);

    for (int c = 0; c < nch; ++c) {
        // chunk c has landed for this wave, then for every wave; after the barrier nobody still reads the other stage
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (c + 1 < nch) issue(c + 1);
        const char* sA = smem + (c % NSTAGE) * STAGE_BYTES;
        const char* sB = sA + 64 * DK * 4;
        if (do_bias) {
#pragma unroll
            for (int k = 0; k < DK; ++k)
                bsum += *reinterpret_cast<const float*>(sA + k * 256 + ((threadIdx.x ^ (((k >> 2) & 1) << 5)) << 2));
        }
        if constexpr (X3) {
            // operand tile t (16 rows) of this wave: eight k values kset(gq) of row 16 t + c16, then the split into bf16 components
            bf16x8 ah[2], al[2], bh[2], bl[2], a2[2], b2[2];
            auto ld = [&](const char* img, const bool kc, const int row, float (&v)[8]) {
                if (kc) {
                    const unsigned sw = (unsigned)((row >> 1) & 7);
                    const f32x4 q0 = *reinterpret_cast<const f32x4*>(img + row * 128 + ((((unsigned)gq) ^ sw) << 4));
                    const f32x4 q1 = *reinterpret_cast<const f32x4*>(img + row * 128 + ((((unsigned)gq + 4u) ^ sw) << 4));
                    v[0] = q0[0]; v[1] = q0[1]; v[2] = q0[2]; v[3] = q0[3];
                    v[4] = q1[0]; v[5] = q1[1]; v[6] = q1[2]; v[7] = q1[3];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int k = (e < 4 ? 4 * gq + e : 16 + 4 * gq + (e - 4));
                        v[e] = *reinterpret_cast<const float*>(img + k * 256 + ((row ^ (((k >> 2) & 1) << 5)) << 2));
                    }
                }
            };
            auto sp = [&](const float (&v)[8], bf16x8& hi, bf16x8& lo, bf16x8& lo2) {
#ifdef GEMM_PROBE_NO_SPLIT   // probe builds only (tools/gemm_loop_probe.sh): the fragments without the split's vector work (results garbage)
                {
                    const unsigned* w = reinterpret_cast<const unsigned*>(v);
                    unsigned* ph = reinterpret_cast<unsigned*>(&hi);
                    unsigned* pl = reinterpret_cast<unsigned*>(&lo);
                    unsigned* p2 = reinterpret_cast<unsigned*>(&lo2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ph[e] = w[e]; pl[e] = w[e + 4]; p2[e] = w[e] ^ w[e + 4]; }
                }
#else
                // the split, two values at a time: v_cvt_pk_bf16_f32 rounds a PAIR (RNE) into one register - already the fragment's
                // layout -, one shift and one mask widen it back, one v_pk_add_f32 takes the (exact) remainders: 4 vector
                // instructions per pair and level.  Written on explicit 2-vectors: left to itself hipcc packed only every other pair
                // and split the rest value by value (7 instructions per pair and level; ISA, round 6).  Same roundings, same bits.
                unsigned* ph = reinterpret_cast<unsigned*>(&hi);
                unsigned* pl = reinterpret_cast<unsigned*>(&lo);
                unsigned* p2 = reinterpret_cast<unsigned*>(&lo2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x2 x = {v[2 * e], v[2 * e + 1]};
                    const unsigned w0 = pack_bf16x2(x);
                    ph[e] = w0;
                    const f32x2 r1 = x - widen_bf16x2(w0);    // exact
                    const unsigned w1 = pack_bf16x2(r1);
                    pl[e] = w1;
                    if constexpr (XM == 2) p2[e] = pack_bf16x2(r1 - widen_bf16x2(w1));   // exact: at most 8 significant bits are left
                }
#endif
            };
            auto fetch = [&](const char* img, const bool kc, const int row, bf16x8& hi, bf16x8& lo, bf16x8& lo2) {
                float v[8];
                ld(img, kc, row, v);
                sp(v, hi, lo, lo2);
            };
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fetch(sA, A_KC, wm * 32 + 16 * t + c16, ah[t], al[t], a2[t]);
                fetch(sB, B_KC, wn * 32 + 16 * t + c16, bh[t], bl[t], b2[t]);
            }
#ifdef GEMM_PROBE_NO_MFMA    // probe builds only: the loop without its MFMAs (the fragments are folded into the accumulators by four adds)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const f32x4 fa = *reinterpret_cast<const f32x4*>(&ah[t2]), fb = *reinterpret_cast<const f32x4*>(&bl[t2]);
                const f32x4 fc = *reinterpret_cast<const f32x4*>(&al[t2]), fd = *reinterpret_cast<const f32x4*>(&bh[t2]);
                acc4[t2][0] += fa + fb;
                acc4[t2][1] += fc + fd;
                if constexpr (XM == 2) acc4[t2][0] += *reinterpret_cast<const f32x4*>(&a2[t2]) + *reinterpret_cast<const f32x4*>(&b2[t2]);
            }
#else
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    if constexpr (XM == 2) {
                        acc4[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[rt], bh[ct], acc4[rt][ct], 0, 0, 0);
                        acc4[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], b2[ct], acc4[rt][ct], 0, 0, 0);
                        acc4[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt], bl[ct], acc4[rt][ct], 0, 0, 0);
                    }
                    acc4[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt], bh[ct], acc4[rt][ct], 0, 0, 0);
                    acc4[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], bl[ct], acc4[rt][ct], 0, 0, 0);
                    acc4[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], bh[ct], acc4[rt][ct], 0, 0, 0);
                }
#endif
        } else {
        float a[4][4], b[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (A_KC) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(sA + rdA + ((((unsigned)(2 * g + h)) ^ swA) << 4));
                a[g][0] = q[0]; a[g][1] = q[1]; a[g][2] = q[2]; a[g][3] = q[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) a[g][j] = *reinterpret_cast<const float*>(sA + rdA + (8 * g + j) * 256);
            }
            if (B_KC) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(sB + rdB + ((((unsigned)(2 * g + h)) ^ swB) << 4));
                b[g][0] = q[0]; b[g][1] = q[1]; b[g][2] = q[2]; b[g][3] = q[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) b[g][j] = *reinterpret_cast<const float*>(sB + rdB + (8 * g + j) * 256);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][j], b[g][j], acc, 0, 0, 0);
        }
    }

    // Where a lane's 16 accumulator values sit in the wave's 32 x 32 tile (row dm, column dn):
    //   32x32 MFMA : register r -> dm = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), dn = lane & 31
    //   X3         : register r = 8 rt + 4 ct + i -> dm = 16 rt + 4 (lane >> 4) + i, dn = 16 ct + (lane & 15)
    if constexpr (X3) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = acc4[r >> 3][(r >> 2) & 1][r & 3];
    }
    auto dm_of = [&](int r) { return X3 ? 16 * (r >> 3) + 4 * gq + (r & 3) : (r & 3) + 8 * (r >> 2) + 4 * h; };
    auto dn_of = [&](int r) { return X3 ? 16 * ((r >> 2) & 1) + c16 : li; };

    const int64_t nw = n0 + wn * 32, mw = m0 + wm * 32;   // origin of this wave's 32 x 32 tile
    if (EPI == EPI_DW) {
        GSTAMP(1);
        // NO fp32 atomics on the output.  HIP's atomicAdd(float*) is an agent-scope global_atomic_add_f32 that the issuing XCD's L2
        // executes, and workgroups of different XCDs adding into one cache line lose updates (common.h: atomic_add_f32; the batch
        // splits of a tile run on different XCDs by construction).  The nz splits of an output tile store their partial tiles in
        // the launch's workspace instead, and the LAST one to arrive (a counter per tile, integer atomics are coherent) sums them
        // in split order and adds the sum to C: one writer per element, bitwise reproducible.  One split (no workspace): plain
        // read-modify-write.
        //   * partials travel with device-scope (sc1) stores / loads: written through to, and read from, the memory side.  NOT a
        //     device-scope fence: that writes back and invalidates the whole L2 per workgroup (measured: +80 us per launch).  An
        //     explicit s_waitcnt vmcnt(0) in every thread, in front of the barrier that precedes the arrival, holds the arrival
        //     back until the stores are acknowledged.
        //   * the last workgroup requests RB splits together (the first version walked them one by one: +12 .. 45 us per launch).
        const int tile = by * p.nx + bx, tid = threadIdx.x;
        if (p.nz > 1) {
            float* part = p.ws_part + ((int64_t)tile * p.nz + bz) * 4096;
#pragma unroll
            for (int r = 0; r < 16; ++r) __hip_atomic_store(&part[r * 256 + tid], acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (do_bias)
                __hip_atomic_store(&p.ws_bias[((int64_t)by * p.nz + bz) * 64 + tid], bsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // EVERY thread holds here until its own sc1 stores are acknowledged by the memory side; only then may the barrier be
            // passed and the arrival be counted.  The workgroup-scope release alone does NOT do that on gfx950 (outside tgsplit
            // mode it emits no vmcnt wait: the ISA had the stores, s_barrier and the global_atomic_add back to back, so the last
            // workgroup of another XCD could read a stale partial).  tools/isa_loop_check.py asserts the wait is in the ISA.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GSTAMP(2);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            int* s_last = reinterpret_cast<int*>(smem);
            if (tid == 0) {
                const unsigned old = atomicAdd(&p.ws_cnt[tile], 1u);
                *s_last = old == (unsigned)(p.nz - 1);
                if (old == (unsigned)(p.nz - 1)) atomicExch(&p.ws_cnt[tile], 0u);   // every split has arrived: ready for the next launch
            }
            __syncthreads();
            GSTAMP(3);
            if (!*s_last) return;
            GSTAMP_IF(tile == 0, 32);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            float* all = p.ws_part + (int64_t)tile * p.nz * 4096 + tid;
            // what the final add reads - this tile of C and the bias gradient - is requested FIRST, and the bias partials of every
            // split together: behind the reduction they were one exposed round trip each (shader-clock stamps, round 3: 5.9 us of
            // a 17.5 us launch between the last partial tile and the end of the workgroup)
            float cold[16], bold = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = mw + dm_of(r), n = nw + dn_of(r);
                cold[r] = (m < p.M && n < p.N) ? p.C[m * p.ldc + n] : 0.f;
            }
            if (do_bias && m0 + tid < p.M) bold = p.bias_grad[m0 + tid];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#ifndef PCVAE_GEMM_RB
#define PCVAE_GEMM_RB 4
#endif
            constexpr int RB = PCVAE_GEMM_RB;   // splits requested together: 64 loads in flight per thread instead of 16
            for (int z0 = 0; z0 < p.nz; z0 += RB) {   // fixed order: bitwise reproducible
                float v[RB][16];
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    const int z = z0 + u < p.nz ? z0 + u : p.nz - 1;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        v[u][r] = __hip_atomic_load(&all[(int64_t)z * 4096 + r * 256], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int u = 0; u < RB; ++u)
                    if (z0 + u < p.nz)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[r] += v[u][r];
            }
            GSTAMP_IF(tile == 0, 33);
            if (do_bias) {
                bsum = 0.f;
                constexpr int BB = 8;   // bias partials requested together (fixed order of the sum: bitwise reproducible)
                for (int z0 = 0; z0 < p.nz; z0 += BB) {
                    float bv[BB];
#pragma unroll
                    for (int u = 0; u < BB; ++u) {
                        const int z = z0 + u < p.nz ? z0 + u : p.nz - 1;
                        bv[u] = __hip_atomic_load(&p.ws_bias[((int64_t)by * p.nz + z) * 64 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int u = 0; u < BB; ++u)
                        if (z0 + u < p.nz) bsum += bv[u];
                }
            }
            if (do_bias && m0 + tid < p.M) p.bias_grad[m0 + tid] = bold + bsum;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = mw + dm_of(r), n = nw + dn_of(r);
                if (m < p.M && n < p.N) p.C[m * p.ldc + n] = cold[r] + acc[r];
            }
            GSTAMP_IF(tile == 0, 34);
            return;
        }
        if (do_bias && m0 + tid < p.M) p.bias_grad[m0 + tid] += bsum;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = mw + dm_of(r), n = nw + dn_of(r);
            if (m < p.M && n < p.N) p.C[m * p.ldc + n] += acc[r];
        }
        GSTAMP_IF(tile == 0, 34);
        return;
    }

    // (a lane's columns: one for the 32 x 32 MFMA tile, two - registers 0..3 | 8..11 and 4..7 | 12..15 - for X3)
    float bias[2] = {0.f, 0.f};
    if (EPI == EPI_FWD && p.bias) {
        if (nw + dn_of(0) < p.N) bias[0] = p.bias[nw + dn_of(0)];
        if (X3 && nw + dn_of(4) < p.N) bias[1] = p.bias[nw + dn_of(4)];
    }
    if (rowsA == BM && rowsB == BN) {   // interior tile (workgroup-uniform): no per-row bounds
#pragma unroll
        for (int r = 0; r < 16; ++r) store_c<EPI>(p, mw + dm_of(r), nw + dn_of(r), acc[r], bias[X3 ? (r >> 2) & 1 : 0]);
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t m = mw + dm_of(r), n = nw + dn_of(r);
        if (m < p.M && n < p.N) store_c<EPI>(p, m, n, acc[r], bias[X3 ? (r >> 2) & 1 : 0]);
    }
}


// ---- small-M variant ------------------------------------------------------------------------------------------------------
// A rank of the data-parallel job holds B/8 = 1024 slates: a [1024 x 256] layer is only 64 tiles of 64 x 64, a quarter
// of the chip, and each of those workgroups is bound by its own MFMA chain (K = 1419: 45 rounds of 16 MFMAs per wave).
// Here a workgroup owns a 32 x 32 tile and its four waves split every 64-deep K chunk between them (wave w multiplies
// k in [16w, 16w + 16)), so the same layer is 256 workgroups with a 4x shorter chain each; the four partial tiles are
// summed through LDS in a fixed order (wave 0 + 1 + 2 + 3: deterministic).  Tiles are staged through registers into
// K-major LDS images (row stride 33 floats: conflict-free ds_read_b32 / ds_write_b32).
template <bool KC>
__device__ __forceinline__ void load_tile_s(const float* __restrict__ P, int64_t ld, int64_t row0, int64_t rows,
                                            int64_t k0, int64_t kend, float (&v)[STPT]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < STPT; ++i) {
        int row, k;
        if (KC) { k = t & 63; row = (t >> 6) + 4 * i; } else { row = t & 31; k = (t >> 5) + 8 * i; }
        const int64_t gr = row0 + row, gk = k0 + k;
        const bool ok = gr < rows && gk < kend;
        v[i] = ok ? (KC ? P[gr * ld + gk] : P[gk * ld + gr]) : 0.f;
    }
}

template <bool KC>
__device__ __forceinline__ void store_tile_s(float* S, const float (&v)[STPT]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < STPT; ++i) {
        int row, k;
        if (KC) { k = t & 63; row = (t >> 6) + 4 * i; } else { row = t & 31; k = (t >> 5) + 8 * i; }
        S[k * SLD + row] = v[i];
    }
}

// 16-byte loads for interior tiles (row starts need only be 4-byte aligned, which global_load_dwordx4 accepts):
//   KC: row = 16 i + (t >> 4), k = 4 (t & 15) + j;   !KC: k = 32 i + (t >> 3), row = 4 (t & 7) + j
template <bool KC>
__device__ __forceinline__ void load_tile_s_v4(const float* __restrict__ P, int64_t ld, int64_t row0, int64_t k0, float (&v)[STPT]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < STPT / 4; ++i) {
        const int major = KC ? 16 * i + (t >> 4) : 32 * i + (t >> 3), minor = KC ? 4 * (t & 15) : 4 * (t & 7);
        const float* src = KC ? P + (row0 + major) * ld + (k0 + minor) : P + (k0 + major) * ld + (row0 + minor);
        typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
        const f32x4u q = *reinterpret_cast<const f32x4u*>(src);
        v[4 * i] = q[0]; v[4 * i + 1] = q[1]; v[4 * i + 2] = q[2]; v[4 * i + 3] = q[3];
    }
}

template <bool KC>
__device__ __forceinline__ void store_tile_s_v4(float* S, const float (&v)[STPT]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < STPT / 4; ++i) {
        const int major = KC ? 16 * i + (t >> 4) : 32 * i + (t >> 3), minor = KC ? 4 * (t & 15) : 4 * (t & 7);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (KC) S[(minor + j) * SLD + major] = v[4 * i + j];
            else S[major * SLD + minor + j] = v[4 * i + j];
        }
    }
}

template <bool A_KC, bool B_KC, int EPI>
__device__ __forceinline__ void gemm_tile_small(const GemmParams& p, const int bx, const int by, char* smem) {
    static_assert(EPI == EPI_FWD || EPI == EPI_DX, "the weight-gradient GEMM is already split over workgroups");
    float* As = reinterpret_cast<float*>(smem);
    float* Bs = As + SBK * SLD;
    float* Red = Bs + SBK * SLD;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int64_t m0 = (int64_t)by * SM, n0 = (int64_t)bx * SM;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    const bool rows_in = m0 + SM <= p.M && n0 + SM <= p.N;   // workgroup-uniform
    GSTAMP(0);
    int stamp_i = 1;

    // What the epilogue reads from memory - the bias (forward), the activated input and, for the second of two layers sharing an
    // input, the running sum (input gradient) - is requested HERE, ahead of the K loop: read at the end it was a fully exposed
    // round trip, 0.5 us of a 4 us workgroup (shader-clock stamps, round 3).  Element i of this thread: row (tid >> 5) + 8 i,
    // column tid & 31.
    float epi_bias = 0.f, epi_aux[4] = {1.f, 1.f, 1.f, 1.f}, epi_c[4] = {0.f, 0.f, 0.f, 0.f};
    {
        const int64_t n = n0 + (threadIdx.x & 31);
        if (EPI == EPI_FWD) {
            if (p.bias && n < p.N) epi_bias = p.bias[n];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t m = m0 + (threadIdx.x >> 5) + 8 * i;
                if (m < p.M && n < p.N) {
                    if (p.aux) epi_aux[i] = p.aux[m * p.ldaux + n];
                    if (p.accumulate) epi_c[i] = p.C[m * p.ldc + n];
                }
            }
        }
    }

    auto mfma_chunk = [&](const float* As, const float* Bs) {
        // operand reads run PRE k-steps ahead of the MFMAs: hipcc's schedule of the plain loop was read, read,
        // s_waitcnt lgkmcnt(0), mfma - a fully exposed LDS latency per MFMA
        constexpr int NS = SBK / 8, PRE = 4;
        float ar[NS], br[NS];
#pragma unroll
        for (int s = 0; s < PRE && s < NS; ++s) {
            const int k = wave * (SBK / 4) + 2 * s + h;
            ar[s] = As[k * SLD + li];
            br[s] = Bs[k * SLD + li];
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + PRE < NS) {
                const int k = wave * (SBK / 4) + 2 * (s + PRE) + h;
                ar[s + PRE] = As[k * SLD + li];
                br[s + PRE] = Bs[k * SLD + li];
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[s], br[s], acc, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
    };

    constexpr int NPRE = 4;
    const int nchunks = (int)((p.K + SBK - 1) / SBK);
    if (nchunks <= NPRE) {
        // short reductions (K <= 256: every layer of the model but the encoder's first): ALL chunks are requested up front, one memory
        // round trip for the tile instead of one per chunk (each chunk waited ~0.6 us for the loads issued one chunk earlier)
        float va[NPRE][STPT], vb[NPRE][STPT];
#pragma unroll
        for (int c = 0; c < NPRE; ++c) {
            if (c < nchunks) {   // workgroup-uniform
                if (rows_in && (int64_t)(c + 1) * SBK <= p.K) {
                    load_tile_s_v4<A_KC>(p.A, p.lda, m0, (int64_t)c * SBK, va[c]);
                    load_tile_s_v4<B_KC>(p.B, p.ldb, n0, (int64_t)c * SBK, vb[c]);
                } else {
                    load_tile_s<A_KC>(p.A, p.lda, m0, p.M, (int64_t)c * SBK, p.K, va[c]);
                    load_tile_s<B_KC>(p.B, p.ldb, n0, p.N, (int64_t)c * SBK, p.K, vb[c]);
                }
            }
        }
        // (two LDS stages - chunk c + 1 written while chunk c is multiplied, one barrier per chunk - were measured: 1100 against 1250
        // ticks per chunk, the launch unchanged at 5.7 us, a workgroup less per CU: not kept)
#pragma unroll
        for (int c = 0; c < NPRE; ++c) {
            if (c < nchunks) {
                if (c) __syncthreads();
                GSTAMP(stamp_i++);
                if (rows_in && (int64_t)(c + 1) * SBK <= p.K) {
                    store_tile_s_v4<A_KC>(As, va[c]);
                    store_tile_s_v4<B_KC>(Bs, vb[c]);
                } else {
                    store_tile_s<A_KC>(As, va[c]);
                    store_tile_s<B_KC>(Bs, vb[c]);
                }
                __syncthreads();
                mfma_chunk(As, Bs);
            }
        }
    } else {
        float va[STPT], vb[STPT];
        bool vec = rows_in && SBK <= p.K;
        if (vec) {
            load_tile_s_v4<A_KC>(p.A, p.lda, m0, 0, va);
            load_tile_s_v4<B_KC>(p.B, p.ldb, n0, 0, vb);
        } else {
            load_tile_s<A_KC>(p.A, p.lda, m0, p.M, 0, p.K, va);
            load_tile_s<B_KC>(p.B, p.ldb, n0, p.N, 0, p.K, vb);
        }
        for (int64_t k0 = 0; k0 < p.K; k0 += SBK) {
            __syncthreads();
            GSTAMP(stamp_i++);
            if (vec) {
                store_tile_s_v4<A_KC>(As, va);
                store_tile_s_v4<B_KC>(Bs, vb);
            } else {
                store_tile_s<A_KC>(As, va);
                store_tile_s<B_KC>(Bs, vb);
            }
            __syncthreads();
            if (k0 + SBK < p.K) {
                vec = rows_in && k0 + 2 * SBK <= p.K;
                if (vec) {
                    load_tile_s_v4<A_KC>(p.A, p.lda, m0, k0 + SBK, va);
                    load_tile_s_v4<B_KC>(p.B, p.ldb, n0, k0 + SBK, vb);
                } else {
                    load_tile_s<A_KC>(p.A, p.lda, m0, p.M, k0 + SBK, p.K, va);
                    load_tile_s<B_KC>(p.B, p.ldb, n0, p.N, k0 + SBK, p.K, vb);
                }
            }
            mfma_chunk(As, Bs);
        }
    }
    GSTAMP(stamp_i++);
    // partial tiles -> LDS as [wave][row][col]
#pragma unroll
    for (int r = 0; r < 16; ++r) Red[wave * SM * SM + ((r & 3) + 8 * (r >> 2) + 4 * h) * SM + li] = acc[r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = threadIdx.x + 256 * i, row = e >> 5, col = e & 31;
        const int64_t m = m0 + row, n = n0 + col;
        if (m >= p.M || n >= p.N) continue;
        float v = ((Red[e] + Red[SM * SM + e]) + Red[2 * SM * SM + e]) + Red[3 * SM * SM + e];
        if (EPI == EPI_FWD) {   // as store_c, on the operands requested ahead of the loop
            v += epi_bias;
            if (p.act == PCVAE_ACT_LEAKY) v = leaky(v);
            else if (p.act == PCVAE_ACT_RELU) v = fmaxf(v, 0.f);
        } else {
            v += epi_c[i];
            if (p.aux && !(epi_aux[i] > 0.f)) v *= kLeakySlope;
        }
        p.C[m * p.ldc + n] = v;
    }
    GSTAMP(stamp_i++);
}

// ---- the kernel: one or several independent problems ---------------------------------------------------------------------
// Hardware workgroup b runs on XCD b & 7 (round-robin dispatch).  Slot s = b >> 3 of an XCD walks the problems in order; inside
// problem j (slots_j = cdiv(tiles_j, 8) slots per XCD) XCD x takes the logical tiles [x slots_j, (x + 1) slots_j): contiguous
// runs, equal shares of every problem on every XCD, the first (largest) problem dispatched first.
// (Two instantiations - a launch is all 64 x 64 DMA tiles or all small tiles - so that the DMA body's ~110 registers, not
// the register-staged small body's 150, set the occupancy of the launches that matter.)
// (... and launches without a weight gradient - every forward and input-gradient launch - take an instantiation without that body:
// its split reduction holds 64 more registers, the difference between five and four workgroups per CU.)
template <bool SMALL, bool HAS_DW, int X3 = 0>   // X3: the arithmetic XM of the 64 x 64 DMA body (0 f32, 1 bf16x3, 2 bf16x6)
__global__ void __launch_bounds__(256) gemm_group_kernel(const GroupParams gp) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int xcd = blockIdx.x & 7;
    int slot = blockIdx.x >> 3, j = 0;
    while (j + 1 < gp.n && slot >= gp.g[j].slots) {
        slot -= gp.g[j].slots;
        ++j;
    }
    const GemmParams p = gp.g[j];
    const int l = xcd * p.slots + slot;
    if (l >= p.nx * p.ny * p.nz) return;
    const int bx = l % p.nx, by = (l / p.nx) % p.ny, bz = l / (p.nx * p.ny);
    if (SMALL) {
        if (p.kind == KIND_FWD_S) gemm_tile_small<true, true, EPI_FWD>(p, bx, by, smem);
        else gemm_tile_small<true, false, EPI_DX>(p, bx, by, smem);
    } else {
        const bool x4 = gp.dma_x4 != 0;
        if (p.kind == KIND_FWD) gemm_tile_dma<true, true, EPI_FWD, X3>(p, bx, by, bz, smem, x4);
        else if (p.kind == KIND_DX || !HAS_DW) gemm_tile_dma<true, false, EPI_DX, X3>(p, bx, by, bz, smem, x4);
        else gemm_tile_dma<false, false, EPI_DW, X3>(p, bx, by, bz, smem, x4);
    }
}

// 64 x 64 tiles once the launch fills the chip, else 32 x 32 tiles with the K chunk split over the waves
// (read at every launch, ~50 ns: PCVAE_GEMM_SMALL_BELOW=0 makes every launch use the 64 x 64 tiles, whose k order per output row
// does not depend on M - the data-parallel tests use it to get forward activations that are BITWISE independent of the sharding)
static inline int64_t small_below() {
    const char* e = getenv("PCVAE_GEMM_SMALL_BELOW");
    return e ? atoll(e) : 256LL;
}

static int check_desc(const pcvae_gemm_desc& d) {
    PCVAE_REQUIRE(d.a && d.b && d.c, "linear: null pointer");
    PCVAE_REQUIRE(d.M >= 0 && d.N > 0 && d.K > 0, "linear: bad shape M=%lld N=%lld K=%lld", (long long)d.M, (long long)d.N,
                  (long long)d.K);
    // a tile's DMA lane offsets are 32-bit byte offsets from the tile origin: 64 rows (or 32 k lines) of ld floats
    PCVAE_REQUIRE(d.lda < (1LL << 22) && d.ldb < (1LL << 22), "linear: leading dimension too large");
    switch (d.kind & ~(PCVAE_GEMM_X3 | PCVAE_GEMM_X6)) {
        case PCVAE_GEMM_FWD:
            PCVAE_REQUIRE(d.lda >= d.K && d.ldb >= d.K && d.ldc >= d.N, "linear_fwd: bad leading dimension");
            PCVAE_REQUIRE(d.act == PCVAE_ACT_NONE || d.act == PCVAE_ACT_LEAKY || d.act == PCVAE_ACT_RELU,
                          "linear_fwd: unknown activation %d", d.act);
            break;
        case PCVAE_GEMM_DX:
        case PCVAE_GEMM_DX_ACC:
            PCVAE_REQUIRE(d.lda >= d.N && d.ldb >= d.K && d.ldc >= d.K && (!d.aux || d.ldaux >= d.K), "linear_bwd_input: bad shape");
            break;
        case PCVAE_GEMM_DW:
            PCVAE_REQUIRE(d.lda >= d.N && d.ldb >= d.K && d.ldc >= d.K, "linear_bwd_weight: bad shape");
            break;
        default:
            PCVAE_REQUIRE(false, "linear: unknown kind %d", d.kind);
    }
    return PCVAE_OK;
}

// Weight gradient C(n, kk) = sum_m dY[m, n] * X[m, kk]: both operands row-contiguous, the reduction over the batch split across z
// so that a [256 x 1419] gradient still fills the chip.  The split count minimises (waves of workgroups) x (K rounds per
// workgroup) for ~512 resident workgroups (sweeps over forced split counts on the model's layers found nothing better).
// Without a workspace there is nothing to combine splits with (no fp32 atomics on the output: gemm_tile_dma): one split.
struct DwPlan {
    int64_t kps;      // batch rows per split (a multiple of 64)
    int nx, ny, nz;
};
static DwPlan dw_plan(const pcvae_gemm_desc& d, bool have_ws) {
    const bool one_split = !have_ws;   // without a workspace there is no way to combine splits (no fp32 atomics: see below)
    const int64_t tiles = cdiv(d.N, BM) * cdiv(d.K, BN), rounds_total = cdiv(d.M, 64);
    int64_t splits = 1, best = INT64_MAX;
    for (int64_t sp = 1; sp <= (one_split ? 1 : std::min<int64_t>(64, rounds_total)); ++sp) {
        const int64_t rounds = cdiv(rounds_total, sp), nsp = cdiv(rounds_total, rounds);   // splits actually launched
        // +1: prologue / epilogue; the last term is the last workgroup's reduction of the nsp partial tiles, four per memory
        // round trip (a round of 64 batch rows ~ 0.5 us ~ 64 units; a round trip ~ 1.5-2 us)
        // red: 96 .. 384 measured within 3 % of each other on config 4's layers (round 2); round 3 took the serial loads out of the last
        // workgroup's tail (5.9 -> 0.85 us) and a batch of four partials now costs 1.6 us = 55 units: the low end, which at M = 1024
        // picks 8 splits of 4 chunks instead of 4 of 8
#ifndef PCVAE_DW_RED
#define PCVAE_DW_RED 96
#endif
        constexpr int64_t red = PCVAE_DW_RED;
        const int64_t cost = cdiv(tiles * nsp, 512) * (rounds + 1) * 64 + nsp + (nsp > 1 ? red * cdiv(nsp, 4) : 0);
        if (cost < best) { best = cost; splits = nsp; }
    }
    const int64_t kps = cdiv(rounds_total, splits) * 64;
    return DwPlan{kps, (int)cdiv(d.K, BN), (int)cdiv(d.N, BM), (int)cdiv(d.M, kps)};
}

// workspace of a launch: [tile counters of every weight-gradient problem: a FIXED 64 KB region, so that launches of different
// shapes sharing one buffer never write partial tiles over counters][problem: partial tiles, bias partials]...
constexpr size_t CNT_REGION = 65536;
static size_t group_ws_bytes(const pcvae_gemm_desc* descs, int n) {
    size_t cnt = 0, part = 0;
    for (int i = 0; i < n; ++i) {
        if ((descs[i].kind & ~(PCVAE_GEMM_X3 | PCVAE_GEMM_X6)) != PCVAE_GEMM_DW || descs[i].M <= 0) continue;
        const DwPlan pl = dw_plan(descs[i], true);
        cnt += (size_t)pl.nx * pl.ny;
        if (pl.nz > 1) part += ((size_t)pl.nx * pl.ny * pl.nz * 4096 + (size_t)pl.ny * pl.nz * 64) * sizeof(float);
    }
    if (cnt * sizeof(unsigned) > CNT_REGION) return 0;   // more output tiles than counters: the caller gets the one-split path
    return cnt == 0 ? 0 : CNT_REGION + part;
}

static int launch_group(const pcvae_gemm_desc* descs, int n, void* ws, size_t ws_bytes, pcvae_stream_t stream) {
    PCVAE_REQUIRE(descs && n >= 1 && n <= MAXG, "linear_group: 1..%d problems per launch", MAXG);
    GroupParams gp;
    gp.n = 0;
    static const int dma16 = [] { const char* e = getenv("PCVAE_GEMM_DMA16"); return e ? atoi(e) : 1; }();
    gp.dma_x4 = dma16;
    int64_t tiles64 = 0;
    // split-bf16 arithmetic: only if EVERY problem of the launch asks for it (one kernel per launch); bf16x6 only if every problem asks
    // for bf16x6 (a mixed launch runs the narrower bf16x3 - a group is built under ONE ops.mlp_arith, so this does not happen)
    bool has_dw = false, x3 = true, x6 = true;
    pcvae_gemm_desc local[MAXG];
    for (int i = 0; i < n; ++i) {
        if (int rc = check_desc(descs[i])) return rc;
        local[i] = descs[i];
        x3 = x3 && (local[i].kind & (PCVAE_GEMM_X3 | PCVAE_GEMM_X6)) != 0;
        x6 = x6 && (local[i].kind & PCVAE_GEMM_X6) != 0;
        local[i].kind &= ~(PCVAE_GEMM_X3 | PCVAE_GEMM_X6);
    }
    descs = local;
    for (int i = 0; i < n; ++i) {
        const pcvae_gemm_desc& d = descs[i];
        if (d.M == 0) continue;
        has_dw |= d.kind == PCVAE_GEMM_DW;
        // output tiles of 64 x 64 over the whole launch decide the tile size of the forward / input-gradient problems
        tiles64 += d.kind == PCVAE_GEMM_FWD ? cdiv(d.M, BM) * cdiv(d.N, BN)
                 : d.kind == PCVAE_GEMM_DW ? 256 : cdiv(d.M, BM) * cdiv(d.K, BN);
    }
    // The weight gradient exists for 64 x 64 tiles only.  A launch whose every problem asks for bf16x3 takes the 64 x 64 tiles at
    // ANY size: the arithmetic of a layer is then a property of the model (set_mlp_precision), not of the per-rank batch - the
    // same model at world = 1 and world = 8, or a ragged last batch, runs every GEMM in the same arithmetic (the 32 x 32 K-split
    // tiles have no bf16x3 body, and launches this small are launch-bound either way).
    const bool small = tiles64 < small_below() && !has_dw && !x3;
    const size_t ws_need = ws ? group_ws_bytes(descs, n) : 0;
    const bool have_ws = ws != nullptr && ws_need > 0;
    if (have_ws) PCVAE_REQUIRE(ws_bytes >= ws_need, "linear_group: workspace too small (pcvae_linear_group_ws_bytes)");
    size_t cnt_off = 0, part_off = CNT_REGION;
    int64_t total = 0;
    // longest problem first: slot order is dispatch order, and a launch ends with its last workgroup
    int order[MAXG];
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order, order + n, [&](int a, int b) {
        return (double)descs[a].M * descs[a].N * descs[a].K > (double)descs[b].M * descs[b].N * descs[b].K;
    });
    for (int oi = 0; oi < n; ++oi) {
        const pcvae_gemm_desc& d = descs[order[oi]];
        if (d.M == 0) continue;
        GemmParams& g = gp.g[gp.n];
        g = GemmParams{};
        if (d.kind == PCVAE_GEMM_FWD) {
            g.A = d.a; g.lda = d.lda; g.B = d.b; g.ldb = d.ldb; g.C = d.c; g.ldc = d.ldc;
            g.M = d.M; g.N = d.N; g.K = d.K; g.bias = d.aux; g.act = d.act;
            g.kind = small ? KIND_FWD_S : KIND_FWD;
            g.nx = (int)cdiv(d.N, small ? SM : BN); g.ny = (int)cdiv(d.M, small ? SM : BM); g.nz = 1;
        } else if (d.kind == PCVAE_GEMM_DX || d.kind == PCVAE_GEMM_DX_ACC) {
            // C(m, kk) = sum_n dY[m, n] * W[n, kk]:  A = dY (reduction index contiguous), B(kk, n) = W[n * ldw + kk]
            g.A = d.a; g.lda = d.lda; g.B = d.b; g.ldb = d.ldb; g.C = d.c; g.ldc = d.ldc;
            g.M = d.M; g.N = d.K; g.K = d.N; g.aux = d.aux; g.ldaux = d.ldaux; g.accumulate = d.kind == PCVAE_GEMM_DX_ACC;
            g.kind = small ? KIND_DX_S : KIND_DX;
            g.nx = (int)cdiv(d.K, small ? SM : BN); g.ny = (int)cdiv(d.M, small ? SM : BM); g.nz = 1;
        } else {
            const DwPlan pl = dw_plan(d, have_ws);
            g.A = d.a; g.lda = d.lda; g.B = d.b; g.ldb = d.ldb; g.C = d.c; g.ldc = d.ldc;
            g.M = d.N; g.N = d.K; g.K = d.M; g.bias_grad = d.aux_out; g.k_per_split = pl.kps;
            g.kind = KIND_DW;
            g.nx = pl.nx; g.ny = pl.ny; g.nz = pl.nz;
            if (have_ws) {
                char* base = static_cast<char*>(ws);
                g.ws_cnt = reinterpret_cast<unsigned*>(base + cnt_off);
                cnt_off += (size_t)pl.nx * pl.ny * sizeof(unsigned);
                g.ws_part = reinterpret_cast<float*>(base + part_off);
                g.ws_bias = g.ws_part + (pl.nz > 1 ? (size_t)pl.nx * pl.ny * pl.nz * 4096 : 0);
                if (pl.nz > 1) part_off += ((size_t)pl.nx * pl.ny * pl.nz * 4096 + (size_t)pl.ny * pl.nz * 64) * sizeof(float);
            }
        }
        const int64_t wgs = (int64_t)g.nx * g.ny * g.nz;
        PCVAE_REQUIRE(wgs < (1LL << 28), "linear: problem too large");
        g.slots = (int)cdiv(wgs, 8);
        total += g.slots;
        ++gp.n;
    }
    if (gp.n == 0) return PCVAE_OK;
    PCVAE_REQUIRE(total * 8 < (1LL << 31), "linear_group: launch too large");
    if (small)   // (exact f32 only: a launch that asks for bf16x3 never takes the K-split 32 x 32 tiles, see above)
        hipLaunchKernelGGL((gemm_group_kernel<true, false>), dim3((unsigned)(total * 8)), dim3(256), SMALL_LDS, as_stream(stream), gp);
    else if (has_dw && x6)
        hipLaunchKernelGGL((gemm_group_kernel<false, true, 2>), dim3((unsigned)(total * 8)), dim3(256), NSTAGE * STAGE_BYTES,
                           as_stream(stream), gp);
    else if (x6)
        hipLaunchKernelGGL((gemm_group_kernel<false, false, 2>), dim3((unsigned)(total * 8)), dim3(256), NSTAGE * STAGE_BYTES,
                           as_stream(stream), gp);
    else if (has_dw && x3)
        hipLaunchKernelGGL((gemm_group_kernel<false, true, 1>), dim3((unsigned)(total * 8)), dim3(256), NSTAGE * STAGE_BYTES,
                           as_stream(stream), gp);
    else if (has_dw)
        hipLaunchKernelGGL((gemm_group_kernel<false, true>), dim3((unsigned)(total * 8)), dim3(256), NSTAGE * STAGE_BYTES,
                           as_stream(stream), gp);
    else if (x3)
        hipLaunchKernelGGL((gemm_group_kernel<false, false, 1>), dim3((unsigned)(total * 8)), dim3(256), NSTAGE * STAGE_BYTES,
                           as_stream(stream), gp);
    else
        hipLaunchKernelGGL((gemm_group_kernel<false, false>), dim3((unsigned)(total * 8)), dim3(256), NSTAGE * STAGE_BYTES,
                           as_stream(stream), gp);
    return check_launch("linear_group");
}

}  // namespace

extern "C" size_t pcvae_linear_group_ws_bytes(const pcvae_gemm_desc* descs, int n) {
    if (!descs || n < 1 || n > MAXG) return 0;
    return group_ws_bytes(descs, n);
}

extern "C" int pcvae_linear_group(const pcvae_gemm_desc* descs, int n, void* ws, size_t ws_bytes, pcvae_stream_t stream) {
    return launch_group(descs, n, ws, ws_bytes, stream);
}

extern "C" int pcvae_linear_fwd(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, float* Y,
                                int64_t ldy, int64_t M, int64_t N, int64_t K, int act, pcvae_stream_t stream) {
    const pcvae_gemm_desc d{PCVAE_GEMM_FWD, act, X, ldx, W, ldw, Y, ldy, bias, 0, nullptr, M, N, K};
    return launch_group(&d, 1, nullptr, 0, stream);
}

extern "C" int pcvae_linear_bwd_input(const float* dY, int64_t lddy, const float* W, int64_t ldw, const float* Xact,
                                      int64_t ldxa, float* dX, int64_t lddx, int64_t M, int64_t N, int64_t K,
                                      pcvae_stream_t stream) {
    const pcvae_gemm_desc d{PCVAE_GEMM_DX, 0, dY, lddy, W, ldw, dX, lddx, Xact, ldxa, nullptr, M, N, K};
    return launch_group(&d, 1, nullptr, 0, stream);
}

// dX = (dX + dY . W) * LeakyReLU'(Xact): the second of two layers fed by the same activated input (the mu / logvar heads of
// the encoder and of the prior) - replaces a GEMM into a temporary, autograd's add, a copy and a separate LeakyReLU' kernel
extern "C" int pcvae_linear_bwd_input_acc(const float* dY, int64_t lddy, const float* W, int64_t ldw, const float* Xact,
                                          int64_t ldxa, float* dX, int64_t lddx, int64_t M, int64_t N, int64_t K,
                                          pcvae_stream_t stream) {
    const pcvae_gemm_desc d{PCVAE_GEMM_DX_ACC, 0, dY, lddy, W, ldw, dX, lddx, Xact, ldxa, nullptr, M, N, K};
    return launch_group(&d, 1, nullptr, 0, stream);
}

extern "C" int pcvae_linear_bwd_weight(const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW,
                                       int64_t lddw, float* db, int64_t M, int64_t N, int64_t K,
                                       pcvae_stream_t stream) {
    const pcvae_gemm_desc d{PCVAE_GEMM_DW, 0, dY, lddy, X, ldx, dW, lddw, nullptr, 0, db, M, N, K};
    return launch_group(&d, 1, nullptr, 0, stream);
}
