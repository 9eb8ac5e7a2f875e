// K3s: a whole MLP stack (encoder / prior / slate completion; reference models/pivotcvae.py:159-174, 205-227, 229-240) as ONE
// launch per direction, for SMALL batches.
//
// Why: at BASELINE config 2 (B = 1024) a Linear layer is ~0.1 GFLOP: its GEMM launch is 7-9 us of ramp, operand latency and drain
// around < 1 us of arithmetic, and the optimisation step is a chain of 16 such launches (profiles/r03_f32_config2_step_launches.txt).
// The rows of a batch never mix inside a stack (y_l = act(W_l y_{l-1} + b_l) row by row, and the input gradient likewise), so a
// workgroup can carry a tile of 16 batch rows through EVERY layer with the activations resident in LDS - one launch, one ramp, no
// inter-layer round trip through HBM / L2.  What stays outside: the weight gradients (a reduction over the batch: the grouped
// split GEMM of gemm_f32.hip, which reads the activations and layer gradients this kernel stores).
//
// Arithmetic: exact fp32, v_mfma_f32_16x16x4_f32 (16 batch rows x 16 outputs x 4 k per instruction, 32 cycles).  A tile of 16 rows
// fills the instruction's M side exactly; the four waves of a workgroup split a layer's OUTPUT columns (16-column tiles), or -
// layers narrower than four tiles - the reduction, with the four partial tiles summed in wave order (deterministic).  The k order
// of an output element depends on the layer shape only - never on the batch size or on where a row sits in the batch - so a row's
// activations are bitwise independent of how a batch is sharded over ranks.
//
// Operands: the activation tile is the MFMA A operand, read from LDS with one ds_read_b128 per 16 k (k order k(g, q, j) = 16 g + 4 q
// + j, q = lane >> 4: both operands use it, any bijection of k is a valid order); the weights are the B operand, read straight from
// L2 into registers (each wave owns its columns, nothing to share through LDS), four groups ahead of the MFMAs that use them.
// Forward: W[n][k .. k+3] is one 16-byte load; input gradient: W[n .. n+3][col] are four 4-byte loads of 64 contiguous bytes per
// 16 lanes.
#include "common.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

using namespace pcvae;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

namespace {

constexpr int MT = 16;                          // batch rows per workgroup
constexpr int PF = 3;                           // weight groups in flight per wave
constexpr int MAXL = PCVAE_STACK_MAX_LAYERS;    // layers per stack
constexpr int MAXS = PCVAE_STACK_MAX_STACKS;    // independent stacks per launch
constexpr int RED_TILES = 3;                    // layers of fewer than 4 column tiles split the reduction over the waves
constexpr int RED_COLS = 16 * RED_TILES;

struct FwdStack {
    const float* x; int64_t ldx; int64_t M;
    int K0, nl, wg0;
    const float* W[MAXL]; const float* b[MAXL];
    float* y[MAXL]; int64_t ldy[MAXL];
    int N[MAXL], act[MAXL];
};
struct FwdParams {
    FwdStack s[MAXS];
    int n, pitch_in, pitch_h;
};

struct BwdStack {
    const float* g; int64_t ldg; int64_t M;       // gradient wrt the last layer's (linear) output [M, N[nl-1]]
    int K0, nl, wg0, dx_cols;
    const float* W[MAXL];
    const float* yin[MAXL]; int64_t ldyin[MAXL];   // layer l's input (= layer l-1's activated output), l >= 1
    float* gout[MAXL]; int64_t ldgout[MAXL];       // gout[l] = gradient wrt layer l's INPUT, masked by act'(yin[l]); l = 0: dx
    int N[MAXL];
};
struct BwdParams {
    BwdStack s[MAXS];
    int n, pitch_h;
};

// every LDS access indexes THIS array by offset: a float* that may point at one of several tiles loses its address space and the
// operand reads become flat loads, which tie the LDS reads to the vmcnt of the weight loads in flight
extern __shared__ __attribute__((aligned(16))) float stack_lds[];

#ifdef STACK_STAMPS   // probe builds only (tools/stack_probe.py): shader-clock stamps of workgroup 0, wave 0
__device__ unsigned long long g_stamps[64];
#define STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps[i] = clock64(); } while (0)
extern "C" int pcvae_stack_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps)); }
#else
#define STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ---- forward: acc[t] += in[16][k groups of this wave] . W[tile t][same k]^T -----------------------------------------------------
// groups gstart, gstart + gstep, ... of the K / 16 whole groups, plus the ragged last group if it falls to this wave
template <int NT>
__device__ __forceinline__ void fwd_pass(const int in, const int pin, const int K, const float* __restrict__ W, const int N,
                                         const int tile0, f32x4 (&acc)[NT], const int gstart, const int gstep) {
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const float* wrow[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = 16 * (tile0 + t) + c;
        wrow[t] = W + (int64_t)(n < N ? n : N - 1) * K;
    }
    const float* arow = stack_lds + in + c * pin;
    // Whole groups of 32 k: lane (column c, slot q) takes W[c][32 g + 8 q .. + 7] as TWO adjacent 16-byte loads, so that the four lanes
    // of a column cover one whole 128-byte line with back-to-back instructions (k order k(g, q, h, j) = 32 g + 8 q + 4 h + j, the same
    // for the A operand).  With 16-k groups every instruction asked for HALF lines and the other halves came a group later, after the
    // other waves' traffic had pushed the line out of the vector L1: twice the L2 requests, and the L2 request rate - not the matrix
    // pipe - set the pace (round 3: 1500 cycles per group of 16 MFMAs = 512).
    const int k32 = K >> 5;
    const int cnt = k32 > gstart ? (k32 - gstart + gstep - 1) / gstep : 0;
    // The ring is BRANCH-FREE (group indices clamp to the last one, a round's surplus groups multiply a zero A operand): with a
    // conditional reload hipcc's s_waitcnt insertion falls back to vmcnt(0) at every use.
    if (cnt > 0) {
        const int glast = gstart + (cnt - 1) * gstep;
        f32x4 wq[PF][NT][2];
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int g = min(gstart + p * gstep, glast);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                wq[p][t][0] = *reinterpret_cast<const f32x4u*>(wrow[t] + 32 * g + 8 * q);
                wq[p][t][1] = *reinterpret_cast<const f32x4u*>(wrow[t] + 32 * g + 8 * q + 4);
            }
            __builtin_amdgcn_sched_barrier(0);   // ring order from the first load on: vmcnt counts in issue order
        }
        for (int i0 = 0; i0 < cnt; i0 += PF) {
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                const int i = i0 + p;
                const int g = min(gstart + i * gstep, glast);
                f32x4 a[2];
                a[0] = *reinterpret_cast<const f32x4*>(arow + 32 * g + 8 * q);
                a[1] = *reinterpret_cast<const f32x4*>(arow + 32 * g + 8 * q + 4);
                if (i >= cnt) a[0] = a[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[t] = mfma4(a[h][j], wq[p][t][h][j], acc[t]);
                const int gn = min(g + PF * gstep, glast);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    wq[p][t][0] = *reinterpret_cast<const f32x4u*>(wrow[t] + 32 * gn + 8 * q);
                    wq[p][t][1] = *reinterpret_cast<const f32x4u*>(wrow[t] + 32 * gn + 8 * q + 4);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // what is left of K (< 32) in steps of 16 k, k order 16 s + 4 q + j: on ONE wave when the waves split the reduction.  The LDS
    // image is zero past K; W is not read there.
    if (gstep == 1 || gstart == k32 % gstep) {
        for (int sg = 2 * k32; 16 * sg < K; ++sg) {
            const int kb = 16 * sg + 4 * q;
            const f32x4 a = *reinterpret_cast<const f32x4*>(arow + kb);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4 w;
                if (kb + 4 <= K) w = *reinterpret_cast<const f32x4u*>(wrow[t] + kb);
                else
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = kb + j < K ? wrow[t][kb + j] : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t] = mfma4(a[j], w[j], acc[t]);
            }
        }
    }
}

// ---- input gradient: acc[t] += gin[16][n groups of this wave] . W[same n][columns of tile t] ------------------------------------
template <int NT>
__device__ __forceinline__ void bwd_pass(const int gin, const int pin, const int N, const float* __restrict__ W, const int64_t ldw,
                                         const int Kc, const int tile0, f32x4 (&acc)[NT], const int gstart, const int gstep) {
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    int col[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int k = 16 * (tile0 + t) + c;
        col[t] = k < Kc ? k : Kc - 1;
    }
    const float* arow = stack_lds + gin + c * pin + 4 * q;
    const float* wbase = W + (int64_t)(4 * q) * ldw;
    const int nfull = N >> 4;
    const int cnt = nfull > gstart ? (nfull - gstart + gstep - 1) / gstep : 0;
    auto load = [&](f32x4 (&dst)[NT], int g) {
        const float* wg = wbase + (int64_t)(16 * g) * ldw;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[t][j] = wg[(int64_t)j * ldw + col[t]];
    };
    if (cnt > 0) {   // branch-free ring: see fwd_pass
        const int glast = gstart + (cnt - 1) * gstep;
        f32x4 wq[PF][NT];
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            load(wq[p], min(gstart + p * gstep, glast));
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int i0 = 0; i0 < cnt; i0 += PF) {
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                const int i = i0 + p;
                const int g = min(gstart + i * gstep, glast);
                f32x4 a = *reinterpret_cast<const f32x4*>(arow + 16 * g);
                if (i >= cnt) a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = mfma4(a[j], wq[p][t][j], acc[t]);
                load(wq[p], min(g + PF * gstep, glast));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if ((N & 15) && nfull >= gstart && (nfull - gstart) % gstep == 0) {   // ragged N: gin is zero past N in LDS; W rows are clamped
        const f32x4 a = *reinterpret_cast<const f32x4*>(arow + 16 * nfull);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = 16 * nfull + 4 * q + j;
                acc[t] = mfma4(a[j], W[(int64_t)(n < N ? n : N - 1) * ldw + col[t]], acc[t]);
            }
    }
}

// One layer of a 16-row tile, either direction.  FWD: out = act(in . W^T + bias), NOUT = layer outputs, NRED = its inputs.
// !FWD: out = (in . W[:, :NOUT]) * act'(aux), NRED = layer outputs, NOUT = the input columns wanted.  `out` gets zeros in the columns
// [NOUT, 16 ceil(NOUT / 16)) - it is the next layer's zero-padded A image.
template <bool FWD>
__device__ __forceinline__ void layer_tile(const int in, const int pin, const int NRED, const float* __restrict__ W, const int64_t ldw,
                                           const int NOUT, const float* __restrict__ bias, const int act,
                                           const float* __restrict__ aux, const int64_t ldaux, const int64_t row0, const int64_t M,
                                           const int out_off, const int pout, const int red_off) {
    float* out = stack_lds + out_off;
    float* red = stack_lds + red_off;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    const int ntiles = (NOUT + 15) >> 4;
    auto finish = [&](float v, float ax, int colx) -> float {   // bias / activation (forward: ax = the bias) or act' (backward)
        if (FWD) {
            v += ax;
            if (act == PCVAE_ACT_LEAKY) v = leaky(v);
            else if (act == PCVAE_ACT_RELU) v = fmaxf(v, 0.f);
        } else if (aux && !(ax > 0.f)) {
            v *= kLeakySlope;
        }
        return v;
    };
    auto aux_at = [&](int row, int colx) -> float {   // clamped: rows past M / columns past NOUT are computed and never stored
        const int64_t m = row0 + row < M ? row0 + row : M - 1;
        return aux[m * ldaux + (colx < NOUT ? colx : NOUT - 1)];
    };
    if (ntiles > RED_TILES) {   // the waves split the columns
        const int tpw = (ntiles + 3) >> 2;
        for (int p0 = 0; p0 < tpw; p0 += 4) {
            const int tile0 = wave * tpw + p0;
            int nt = tpw - p0 < 4 ? tpw - p0 : 4;
            if (tile0 + nt > ntiles) nt = ntiles - tile0;
            if (nt <= 0) break;
            f32x4 acc[4];
            float ax[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) ax[t][r] = (!FWD && aux && t < nt) ? aux_at(4 * q + r, 16 * (tile0 + t) + c) : 1.f;   // ahead of the loop
                if (FWD) ax[t][0] = (bias && t < nt && 16 * (tile0 + t) + c < NOUT) ? bias[16 * (tile0 + t) + c] : 0.f;
            }
            auto run = [&](auto tag) {
                constexpr int NT = decltype(tag)::value;
                f32x4(&a)[NT] = reinterpret_cast<f32x4(&)[NT]>(acc);
                if (FWD) fwd_pass<NT>(in, pin, NRED, W, NOUT, tile0, a, 0, 1);
                else bwd_pass<NT>(in, pin, NRED, W, ldw, NOUT, tile0, a, 0, 1);
            };
            if (nt == 4) run(std::integral_constant<int, 4>{});
            else if (nt == 3) run(std::integral_constant<int, 3>{});
            else if (nt == 2) run(std::integral_constant<int, 2>{});
            else run(std::integral_constant<int, 1>{});
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t >= nt) break;
                const int colx = 16 * (tile0 + t) + c;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * q + r;
                    out[row * pout + colx] = colx < NOUT ? finish(acc[t][r], ax[t][FWD ? 0 : r], colx) : 0.f;
                }
            }
        }
    } else {   // narrow layer: the waves split the reduction, partial tiles meet in LDS
        f32x4 acc[RED_TILES];
#pragma unroll
        for (int t = 0; t < RED_TILES; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto run = [&](auto tag) {
            constexpr int NT = decltype(tag)::value;
            f32x4(&a)[NT] = reinterpret_cast<f32x4(&)[NT]>(acc);
            if (FWD) fwd_pass<NT>(in, pin, NRED, W, NOUT, 0, a, wave, 4);
            else bwd_pass<NT>(in, pin, NRED, W, ldw, NOUT, 0, a, wave, 4);
        };
        if (ntiles == 3) run(std::integral_constant<int, 3>{});
        else if (ntiles == 2) run(std::integral_constant<int, 2>{});
        else run(std::integral_constant<int, 1>{});
#pragma unroll
        for (int t = 0; t < RED_TILES; ++t) {
            if (t >= ntiles) break;
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * MT + 4 * q + r) * RED_COLS + 16 * t + c] = acc[t][r];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < MT * 16 * ntiles; e += 256) {
            const int row = e / (16 * ntiles), colx = e - row * 16 * ntiles;
            const float* pr = red + row * RED_COLS + colx;
            const float v = ((pr[0] + pr[MT * RED_COLS]) + pr[2 * MT * RED_COLS]) + pr[3 * MT * RED_COLS];
            out[row * pout + colx] = colx < NOUT ? finish(v, FWD ? (bias ? bias[colx] : 0.f) : (aux ? aux_at(row, colx) : 1.f), colx) : 0.f;
        }
    }
}

// rows [row0, row0 + 16) x [0, ncols) of a global matrix -> LDS tile, zero past M and from ncols to 16 ceil(ncols / 16).
// By LDS-DMA (global_load_lds_dword, 64 consecutive floats of one row per wave instruction; lanes past the matrix read a zero page):
// every request of the tile is in flight at once and nothing is staged through registers - a load / ds_write loop pays a memory
// round trip per iteration (measured: 13 iterations, 9 us for the [16 x 198] encoder tile of config 2).  Invisible to hipcc's vmcnt
// bookkeeping: the caller waits by hand (tile_in_wait) before its barrier.
__device__ float g_stack_zero[64];

__device__ __forceinline__ void tile_in(const float* __restrict__ src, const int64_t ld, const int64_t row0, const int64_t M,
                                        const int ncols, const int dst_off, const int pitch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int npad = (ncols + 15) & ~15, segs = (npad + 63) >> 6;
    for (int e = __builtin_amdgcn_readfirstlane(wave); e < MT * segs; e += 4) {   // wave-uniform
        const int row = e / segs, k = 64 * (e - row * segs) + lane;
        if (k < npad) {
            const float* p = (row0 + row < M && k < ncols) ? src + (row0 + row) * ld + k : g_stack_zero + lane;
            const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(dst_off + row * pitch + 64 * (e - row * segs)) * 4u);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(p), "s"(dst) : "memory", "m0");
        }
    }
}
__device__ __forceinline__ void tile_in_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// LDS tile -> rows [row0, row0 + 16) x [0, ncols) of a global matrix
__device__ __forceinline__ void tile_out(const int src_off, const int pitch, float* __restrict__ dst, const int64_t ld, const int64_t row0,
                                         const int64_t M, const int ncols) {
    const float* src = stack_lds + src_off;
    if ((ld & 3) == 0 && (ncols & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        const int n4 = ncols >> 2;
        for (int e = threadIdx.x; e < MT * n4; e += 256) {
            const int row = e / n4, k = 4 * (e - row * n4);
            if (row0 + row < M) *reinterpret_cast<f32x4*>(dst + (row0 + row) * ld + k) = *reinterpret_cast<const f32x4*>(src + row * pitch + k);
        }
    } else {
        for (int e = threadIdx.x; e < MT * ncols; e += 256) {
            const int row = e / ncols, k = e - row * ncols;
            if (row0 + row < M) dst[(row0 + row) * ld + k] = src[row * pitch + k];
        }
    }
}

__global__ void __launch_bounds__(256) stack_fwd_kernel(const FwdParams fp) {
    int j = 0;
    while (j + 1 < fp.n && (int)blockIdx.x >= fp.s[j + 1].wg0) ++j;
    const FwdStack& s = fp.s[j];
    const int64_t row0 = (int64_t)(blockIdx.x - s.wg0) * MT;
    const int tin = 0, h0 = MT * fp.pitch_in, red = h0 + 2 * MT * fp.pitch_h;   // LDS offsets (floats)
    STAMP(0);
    tile_in(s.x, s.ldx, row0, s.M, s.K0, tin, fp.pitch_in);
    tile_in_wait();
    STAMP(1);
    __syncthreads();
    STAMP(2);
    int in = tin, pin = fp.pitch_in, K = s.K0;
    for (int l = 0; l < s.nl; ++l) {
        const int out = h0 + (l & 1) * MT * fp.pitch_h;
        STAMP(3 + 4 * l);
        layer_tile<true>(in, pin, K, s.W[l], K, s.N[l], s.b[l], s.act[l], nullptr, 0, row0, s.M, out, fp.pitch_h, red);
        STAMP(4 + 4 * l);
        __syncthreads();
        STAMP(5 + 4 * l);
        tile_out(out, fp.pitch_h, s.y[l], s.ldy[l], row0, s.M, s.N[l]);   // the backward pass reads every layer's output
        STAMP(6 + 4 * l);
        in = out; pin = fp.pitch_h; K = s.N[l];
    }
}

__global__ void __launch_bounds__(256) stack_bwd_kernel(const BwdParams bp) {
    int j = 0;
    while (j + 1 < bp.n && (int)blockIdx.x >= bp.s[j + 1].wg0) ++j;
    const BwdStack& s = bp.s[j];
    const int64_t row0 = (int64_t)(blockIdx.x - s.wg0) * MT;
    const int red = 2 * MT * bp.pitch_h;
    tile_in(s.g, s.ldg, row0, s.M, s.N[s.nl - 1], 0, bp.pitch_h);
    tile_in_wait();
    __syncthreads();
    int cur = 0;
    for (int l = s.nl - 1; l >= 0; --l) {
        const int ncols = l > 0 ? s.N[l - 1] : s.dx_cols;
        if (ncols <= 0) break;
        const int64_t ldw = l > 0 ? s.N[l - 1] : s.K0;
        const int out = (cur ^ 1) * MT * bp.pitch_h;
        layer_tile<false>(cur * MT * bp.pitch_h, bp.pitch_h, s.N[l], s.W[l], ldw, ncols, nullptr, 0, l > 0 ? s.yin[l] : nullptr, s.ldyin[l],
                          row0, s.M, out, bp.pitch_h, red);
        __syncthreads();
        tile_out(out, bp.pitch_h, s.gout[l], s.ldgout[l], row0, s.M, ncols);
        cur ^= 1;
    }
}

inline int pad16(int v) { return (v + 15) & ~15; }

// PCVAE_STACK_FUSED=0 (environment, read per call) reports every stack as not eligible: the tests use it to compare the two routes
bool fused_enabled() {
    const char* e = getenv("PCVAE_STACK_FUSED");
    return !(e && atoi(e) == 0);
}

constexpr int LDS_MAX = 160 * 1024;

bool shape_ok(int64_t M, int K0, int nl, const int32_t* N, size_t* lds_in_floats, int* hmax) {
    if (M <= 0 || K0 <= 0 || nl < 1 || nl > MAXL) return false;
    int hm = 0;
    for (int l = 0; l < nl; ++l) {
        if (N[l] <= 0) return false;
        hm = std::max(hm, pad16(N[l]));
    }
    *lds_in_floats = (size_t)MT * (pad16(K0) + 4);
    *hmax = hm;
    return true;
}

}  // namespace

extern "C" int pcvae_stack_eligible(const pcvae_stack_desc* st, int n) {
    if (!st || n < 1 || n > MAXS || !fused_enabled()) return 0;
    int64_t wgs = 0;
    size_t in_max = 0;
    int hmax = 0;
    for (int i = 0; i < n; ++i) {
        size_t in_f;
        int hm;
        if (!shape_ok(st[i].M, st[i].K0, st[i].n_layers, st[i].N, &in_f, &hm)) return 0;
        in_max = std::max(in_max, in_f);
        hmax = std::max(hmax, hm);
        wgs += cdiv(st[i].M, MT);
    }
    // one round of workgroups: past that the tiled GEMMs (every CU on every layer) are the faster route
    if (wgs > 256) return 0;
    const size_t lds = (in_max + 2 * (size_t)MT * (hmax + 4) + 4 * MT * RED_COLS) * sizeof(float);
    return lds <= (size_t)LDS_MAX ? 1 : 0;
}

extern "C" int pcvae_stack_fwd(const pcvae_stack_desc* st, int n, pcvae_stream_t stream) {
    PCVAE_REQUIRE(st && n >= 1 && n <= MAXS, "stack_fwd: 1..%d stacks per launch", MAXS);
    FwdParams fp{};
    fp.n = n;
    int64_t wg = 0;
    size_t in_max = 0;
    int hmax = 0;
    for (int i = 0; i < n; ++i) {
        const pcvae_stack_desc& d = st[i];
        size_t in_f;
        int hm;
        PCVAE_REQUIRE(shape_ok(d.M, d.K0, d.n_layers, d.N, &in_f, &hm), "stack_fwd: bad shape (stack %d)", i);
        PCVAE_REQUIRE(d.x && d.ldx >= d.K0, "stack_fwd: bad input (stack %d)", i);
        FwdStack& s = fp.s[i];
        s.x = d.x; s.ldx = d.ldx; s.M = d.M; s.K0 = d.K0; s.nl = d.n_layers; s.wg0 = (int)wg;
        for (int l = 0; l < d.n_layers; ++l) {
            PCVAE_REQUIRE(d.W[l] && d.y[l] && d.ldy[l] >= d.N[l], "stack_fwd: bad layer %d of stack %d", l, i);
            PCVAE_REQUIRE(d.act[l] == PCVAE_ACT_NONE || d.act[l] == PCVAE_ACT_LEAKY || d.act[l] == PCVAE_ACT_RELU,
                          "stack_fwd: unknown activation %d", d.act[l]);
            s.W[l] = d.W[l]; s.b[l] = d.bias[l]; s.y[l] = d.y[l]; s.ldy[l] = d.ldy[l]; s.N[l] = d.N[l]; s.act[l] = d.act[l];
        }
        in_max = std::max(in_max, in_f);
        hmax = std::max(hmax, hm);
        wg += cdiv(d.M, MT);
    }
    PCVAE_REQUIRE(wg < (1LL << 30), "stack_fwd: batch too large");
    fp.pitch_in = (int)(in_max / MT);
    fp.pitch_h = hmax + 4;
    const size_t lds = (in_max + 2 * (size_t)MT * fp.pitch_h + 4 * MT * RED_COLS) * sizeof(float);
    PCVAE_REQUIRE(lds <= (size_t)LDS_MAX, "stack_fwd: layers too wide for LDS (%zu bytes): use pcvae_linear_group", lds);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stack_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_set = lds;
    }
    hipLaunchKernelGGL(stack_fwd_kernel, dim3((unsigned)wg), dim3(256), lds, as_stream(stream), fp);
    return check_launch("stack_fwd");
}

extern "C" int pcvae_stack_bwd(const pcvae_stack_desc* st, int n, pcvae_stream_t stream) {
    PCVAE_REQUIRE(st && n >= 1 && n <= MAXS, "stack_bwd: 1..%d stacks per launch", MAXS);
    BwdParams bp{};
    bp.n = n;
    int64_t wg = 0;
    int hmax = 0;
    for (int i = 0; i < n; ++i) {
        const pcvae_stack_desc& d = st[i];
        size_t in_f;
        int hm;
        PCVAE_REQUIRE(shape_ok(d.M, d.K0, d.n_layers, d.N, &in_f, &hm), "stack_bwd: bad shape (stack %d)", i);
        const int L = d.n_layers;
        PCVAE_REQUIRE(d.g && d.ldg >= d.N[L - 1], "stack_bwd: bad upstream gradient (stack %d)", i);
        PCVAE_REQUIRE(d.dx_cols >= 0 && d.dx_cols <= d.K0 && (d.dx_cols == 0 || (d.gout[0] && d.ldgout[0] >= d.dx_cols)),
                      "stack_bwd: bad dx window (stack %d)", i);
        BwdStack& s = bp.s[i];
        s.g = d.g; s.ldg = d.ldg; s.M = d.M; s.K0 = d.K0; s.nl = L; s.wg0 = (int)wg; s.dx_cols = d.dx_cols;
        for (int l = 0; l < L; ++l) {
            PCVAE_REQUIRE(d.W[l], "stack_bwd: bad layer %d of stack %d", l, i);
            s.W[l] = d.W[l]; s.N[l] = d.N[l];
            s.gout[l] = d.gout[l]; s.ldgout[l] = d.ldgout[l];
            if (l > 0) {
                // the input of layer l is layer l-1's output y[l-1]; a NULL yin means "layer l-1 has no activation"
                s.yin[l] = d.act[l - 1] == PCVAE_ACT_LEAKY ? d.y[l - 1] : nullptr;
                s.ldyin[l] = d.ldy[l - 1];
                PCVAE_REQUIRE(d.act[l - 1] == PCVAE_ACT_LEAKY || d.act[l - 1] == PCVAE_ACT_NONE, "stack_bwd: LeakyReLU or linear layers only");
                PCVAE_REQUIRE(d.act[l - 1] != PCVAE_ACT_LEAKY || (d.y[l - 1] && d.ldy[l - 1] >= d.N[l - 1]), "stack_bwd: layer %d needs its input", l);
                PCVAE_REQUIRE(d.gout[l] && d.ldgout[l] >= d.N[l - 1], "stack_bwd: bad gradient buffer of layer %d", l);
            }
        }
        hmax = std::max(hmax, std::max(hm, pad16(d.dx_cols)));
        wg += cdiv(d.M, MT);
    }
    PCVAE_REQUIRE(wg < (1LL << 30), "stack_bwd: batch too large");
    bp.pitch_h = hmax + 4;
    const size_t lds = (2 * (size_t)MT * bp.pitch_h + 4 * MT * RED_COLS) * sizeof(float);
    PCVAE_REQUIRE(lds <= (size_t)LDS_MAX, "stack_bwd: layers too wide for LDS (%zu bytes): use pcvae_linear_group", lds);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stack_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_set = lds;
    }
    hipLaunchKernelGGL(stack_bwd_kernel, dim3((unsigned)wg), dim3(256), lds, as_stream(stream), bp);
    return check_launch("stack_bwd");
}
