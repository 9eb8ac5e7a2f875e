// HBM-bound kernels of the PivotCVAE hot path (gfx950): embedding gather (K1), click-count
// condition (K2), reparameterisation with wavefront Philox (K4), Gaussian-prior KL (K7), Adam over
// a flat buffer (K8), candidate-set scores (K9) and small helpers.  All are sized for 64-wide
// wavefronts and 16 B/lane accesses where alignment allows.
#include "common.h"

using namespace pcvae;

// =============================================================================================
// K1: row gather.  One 16-byte chunk per lane; LPR lanes cover one row, 64/LPR rows per
// wave-instruction, UNROLL row groups in flight per wave so the dependent idx->row loads overlap.
// =============================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: what the nontemporal builtins accept
#ifndef GATHER_UNROLL
#define GATHER_UNROLL 16  // rows in flight per lane group: 16 measured best of {2,4,8,16} (tools/bench_gather.py)
#endif
#ifndef GATHER_NT_LOAD
#define GATHER_NT_LOAD 1
#endif
#ifndef GATHER_NT_STORE
#define GATHER_NT_STORE 1
#endif
#ifndef GATHER_COAL
#define GATHER_COAL 1   // one coalesced index load per wave and batch (gather_rows_coal_kernel); 0: sixteen broadcast loads per lane
#endif
#ifndef GATHER_MAXBLOCKS
#define GATHER_MAXBLOCKS (256 * 8)
#endif

template <int UNROLL, bool CONTIG>
__global__ void __launch_bounds__(256) gather_rows_vec4_kernel(const f32x4* __restrict__ table, int chunks, int lpr,
                                                               const int64_t* __restrict__ idx, int64_t n_idx,
                                                               int group, float* __restrict__ out, int64_t out_ld,
                                                               int D) {
    const int lane = threadIdx.x & 63;
    const int rows_per_wave = 64 / lpr;
    const int sub = lane / lpr;     // which row of the wave's row group
    const int chunk0 = lane % lpr;  // first 16 B chunk of the row handled by this lane
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t n_groups = (n_idx + rows_per_wave - 1) / rows_per_wave;
    for (int64_t g0 = wave * UNROLL; g0 < n_groups; g0 += n_waves * UNROLL) {
        int64_t src[UNROLL];
        int64_t dst[UNROLL];  // float offset of the destination row
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t i = (g0 + u) * rows_per_wave + sub;
            const bool ok = g0 + u < n_groups && i < n_idx;
            src[u] = ok ? idx[i] : -1;
            // CONTIG: out_ld == group * D, i.e. output row i starts at i * D (no 64-bit division on the hot path)
            dst[u] = CONTIG ? i * (int64_t)D : (i / group) * out_ld + (i % group) * (int64_t)D;
        }
        for (int c = chunk0; c < chunks; c += lpr) {
            f32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                if (src[u] >= 0) v[u] = GATHER_NT_LOAD ? __builtin_nontemporal_load(&table[src[u] * chunks + c]) : table[src[u] * chunks + c];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                if (src[u] >= 0) {
                    if (GATHER_NT_STORE) __builtin_nontemporal_store(v[u], reinterpret_cast<f32x4*>(out + dst[u] + c * 4));
                    else *reinterpret_cast<f32x4*>(out + dst[u] + c * 4) = v[u];
                }
        }
    }
}

// Round 4: the same batch with the wave's row indices fetched by ONE coalesced load - lane l < RPB loads idx[first + l] (RPB = the
// 16 * 64 / LPR rows of a batch: 16 / 32 / 64 for rows of 64 / 32 / 16 chunks), and every lane then reads the index of each of its
// 16 rows from the lane that holds it (ds_bpermute: no LDS memory) - instead of 16 loads per lane of which only 64 / LPR addresses
// per instruction are distinct.  One index round trip in front of the row loads instead of sixteen queued ones, and a sixteenth
// of the address-unit work: in tools/gather_probe.hip (bare HIP, same shape, same box) 21.7 us against 34.6 us event to event.
template <int LPR, bool CONTIG>
__global__ void __launch_bounds__(256) gather_rows_coal_kernel(const f32x4* __restrict__ table, const int64_t* __restrict__ idx,
                                                               int64_t n_idx, int group, float* __restrict__ out, int64_t out_ld,
                                                               int D) {
    constexpr int U = 16, RPW = 64 / LPR, RPB = U * RPW, CH = LPR;   // LPR lanes per row = 16-byte chunks per row (D = 4 LPR)
    static_assert(RPB <= 64, "a batch's indices fit one wave-wide load");
    const int lane = threadIdx.x & 63, sub = lane / LPR, c = lane % LPR;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t first = wave * RPB; first < n_idx; first += n_waves * RPB) {
        const int64_t mine = (lane < RPB && first + lane < n_idx) ? idx[first + lane] : -1;
        const int lo = (int)(mine & 0xffffffff), hi = (int)(mine >> 32);
        int64_t src[U];
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int from = RPW * u + sub;
            src[u] = ((int64_t)__shfl(hi, from, 64) << 32) | (uint32_t)__shfl(lo, from, 64);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (src[u] >= 0) v[u] = GATHER_NT_LOAD ? __builtin_nontemporal_load(&table[src[u] * CH + c]) : table[src[u] * CH + c];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (src[u] >= 0) {
                const int64_t i = first + RPW * u + sub;
                // CONTIG: out_ld == group * D, i.e. output row i starts at i * D (no 64-bit division on the hot path)
                float* o = out + (CONTIG ? i * (int64_t)D : (i / group) * out_ld + (i % group) * (int64_t)D) + c * 4;
                if (GATHER_NT_STORE) __builtin_nontemporal_store(v[u], reinterpret_cast<f32x4*>(o));
                else *reinterpret_cast<f32x4*>(o) = v[u];
            }
    }
}

// (Round 3, measured and dropped: the wave's row indices through the SCALAR cache - 4 s_load_dwordx16 instead of 16 broadcast vector
// loads.  23.3 us against 21.5 us for this kernel on the same box: every row load then waits for ALL indices (one lgkmcnt), while
// here a lane's row load follows its own index load.)

// generic fallback (any D / alignment): one wave per row, 4 B per lane
__global__ void __launch_bounds__(256) gather_rows_scalar_kernel(const float* __restrict__ table, int D,
                                                                 const int64_t* __restrict__ idx, int64_t n_idx,
                                                                 int group, float* __restrict__ out, int64_t out_ld) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t i = wave; i < n_idx; i += n_waves) {
        const float* s = table + idx[i] * D;
        float* o = out + (i / group) * out_ld + (i % group) * (int64_t)D;
        for (int d = lane; d < D; d += 64) o[d] = s[d];
    }
}

extern "C" int pcvae_gather_rows(const float* table, int64_t n_rows, int D, const int64_t* idx, int64_t n_idx,
                                 int group, float* out, int64_t out_ld, pcvae_stream_t stream) {
    if (n_idx == 0) return PCVAE_OK;  // empty batch: nothing to do (pointers may be null)
    PCVAE_REQUIRE(table && idx && out, "gather_rows: null pointer");
    PCVAE_REQUIRE(D > 0 && n_rows > 0 && group > 0, "gather_rows: bad D/n_rows/group (%d, %lld, %d)", D,
                  (long long)n_rows, group);
    PCVAE_REQUIRE(out_ld >= (int64_t)group * D, "gather_rows: out_ld %lld < group*D %lld", (long long)out_ld,
                  (long long)group * D);
    const bool vec = (D % 4 == 0) && (out_ld % 4 == 0) && (((uintptr_t)out) % 16 == 0) && (((uintptr_t)table) % 16 == 0);
    if (vec) {
        const int chunks = D / 4;
        int lpr = 1;
        while (lpr < chunks && lpr < 64) lpr <<= 1;
        const int rows_per_wave = 64 / lpr;
        const int64_t n_groups = cdiv(n_idx, rows_per_wave);
        const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(cdiv(cdiv(n_groups, GATHER_UNROLL), 4), GATHER_MAXBLOCKS));
        const bool contig = out_ld == (int64_t)group * D;
        const f32x4* tab4 = reinterpret_cast<const f32x4*>(table);
#define PCVAE_GATHER_COAL(L, C)                                                                                          \
        PCVAE_LAUNCH_TIMED(PCVAE_TIMER_GATHER, (gather_rows_coal_kernel<L, C>), dim3((unsigned)blocks), dim3(256), 0,   \
                           as_stream(stream), tab4, idx, n_idx, group, out, out_ld, D)
        if (GATHER_COAL && GATHER_UNROLL == 16 && chunks == lpr && (lpr == 16 || lpr == 32 || lpr == 64)) {   // D = 64 / 128 / 256
            if (lpr == 16) { if (contig) PCVAE_GATHER_COAL(16, true); else PCVAE_GATHER_COAL(16, false); }
            else if (lpr == 32) { if (contig) PCVAE_GATHER_COAL(32, true); else PCVAE_GATHER_COAL(32, false); }
            else { if (contig) PCVAE_GATHER_COAL(64, true); else PCVAE_GATHER_COAL(64, false); }
            return check_launch("gather_rows");
        }
#undef PCVAE_GATHER_COAL
        if (out_ld == (int64_t)group * D)
            PCVAE_LAUNCH_TIMED(PCVAE_TIMER_GATHER, (gather_rows_vec4_kernel<GATHER_UNROLL, true>), dim3((unsigned)blocks), dim3(256), 0,
                               as_stream(stream), reinterpret_cast<const f32x4*>(table), chunks, lpr, idx, n_idx, group, out, out_ld, D);
        else
            PCVAE_LAUNCH_TIMED(PCVAE_TIMER_GATHER, (gather_rows_vec4_kernel<GATHER_UNROLL, false>), dim3((unsigned)blocks), dim3(256), 0,
                               as_stream(stream), reinterpret_cast<const f32x4*>(table), chunks, lpr, idx, n_idx, group, out, out_ld, D);
    } else {
        const int64_t blocks = std::min<int64_t>(cdiv(n_idx, 4), 256 * 8);
        hipLaunchKernelGGL(gather_rows_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), table,
                           D, idx, n_idx, group, out, out_ld);
    }
    return check_launch("gather_rows");
}

// which kernel pcvae_gather_rows launches for a shape with 16-byte aligned table / out (bench.py and tools/ label their
// measurements with the name rocprofv3's kernel trace shows): 0 = gather_rows_scalar_kernel, 1 = gather_rows_vec4_kernel,
// 2 = gather_rows_coal_kernel
extern "C" int pcvae_gather_rows_variant(int D, int group, int64_t out_ld) {
    if (D <= 0 || D % 4 != 0 || out_ld % 4 != 0) return 0;
    const int chunks = D / 4;
    int lpr = 1;
    while (lpr < chunks && lpr < 64) lpr <<= 1;
    (void)group;
    return (GATHER_COAL && GATHER_UNROLL == 16 && chunks == lpr && (lpr == 16 || lpr == 32 || lpr == 64)) ? 2 : 1;
}

// =============================================================================================
// K2: condition one-hot
// =============================================================================================
__global__ void condition_kernel(const float* __restrict__ r, int64_t B, int ncols, int S, float* __restrict__ out,
                                 int64_t out_ld) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float cnt = 0.f;
    for (int s = 0; s < ncols; ++s) cnt += r[b * ncols + s];
    const int c = (int)cnt;  // torch: sum(r).to(long) truncates
    for (int j = 0; j <= S; ++j) out[b * out_ld + j] = (j == c) ? 1.f : 0.f;
}

extern "C" int pcvae_condition(const float* r, int64_t B, int ncols, int S, float* out, int64_t out_ld,
                               pcvae_stream_t stream) {
    PCVAE_REQUIRE(r && out && S > 0 && ncols > 0 && out_ld >= S + 1, "condition: bad arguments");
    if (B == 0) return PCVAE_OK;
    hipLaunchKernelGGL(condition_kernel, dim3((unsigned)cdiv(B, 256)), dim3(256), 0, as_stream(stream), r, B, ncols, S,
                       out, out_ld);
    return check_launch("condition");
}

// =============================================================================================
// strided 2-D copy / LeakyReLU backward
// =============================================================================================
__global__ void copy2d_kernel(const float* __restrict__ src, int64_t src_ld, float* __restrict__ dst, int64_t dst_ld,
                              int64_t rows, int cols) {
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i % cols);
        dst[r * dst_ld + c] = src[r * src_ld + c];
    }
}

extern "C" int pcvae_copy2d(const float* src, int64_t src_ld, float* dst, int64_t dst_ld, int64_t rows, int cols,
                            pcvae_stream_t stream) {
    PCVAE_REQUIRE(src && dst && cols > 0 && src_ld >= cols && dst_ld >= cols, "copy2d: bad arguments");
    if (rows == 0) return PCVAE_OK;
    const int64_t blocks = std::min<int64_t>(cdiv(rows * cols, 256), 2048);
    hipLaunchKernelGGL(copy2d_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), src, src_ld, dst, dst_ld,
                       rows, cols);
    return check_launch("copy2d");
}

// torch.cat(parts, 1) of up to four column blocks in ONE launch (a concat of k parts was k copy2d launches; at B/8 slates
// per rank every launch is ~4.6 us of floor).  Part i is [rows, cols[i]] with leading dimension ld[i]; unused parts: cols = 0.
struct ConcatParams {
    const float* src[4];
    int64_t ld[4];
    int cols[4];
};

__global__ void concat_kernel(ConcatParams p, float* __restrict__ dst, int64_t dst_ld, int64_t rows, int total) {
    const int64_t n = rows * total;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / total;
        int c = (int)(i % total);
        const int c_out = c;
        int part = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (part == k && c >= p.cols[k]) { c -= p.cols[k]; part = k + 1; }
        dst[r * dst_ld + c_out] = p.src[part][r * p.ld[part] + c];
    }
}

extern "C" int pcvae_concat(const float* s0, int64_t ld0, int c0, const float* s1, int64_t ld1, int c1, const float* s2,
                            int64_t ld2, int c2, const float* s3, int64_t ld3, int c3, float* dst, int64_t dst_ld,
                            int64_t rows, pcvae_stream_t stream) {
    const int total = c0 + c1 + c2 + c3;
    PCVAE_REQUIRE(dst && c0 > 0 && c1 >= 0 && c2 >= 0 && c3 >= 0 && dst_ld >= total, "concat: bad arguments");
    PCVAE_REQUIRE(s0 && ld0 >= c0 && (c1 == 0 || (s1 && ld1 >= c1)) && (c2 == 0 || (s2 && ld2 >= c2)) &&
                  (c3 == 0 || (s3 && ld3 >= c3)), "concat: bad part");
    PCVAE_REQUIRE((c2 == 0 || c1 > 0) && (c3 == 0 || c2 > 0), "concat: parts must be packed from the front");
    if (rows == 0) return PCVAE_OK;
    ConcatParams p{{s0, s1 ? s1 : s0, s2 ? s2 : s0, s3 ? s3 : s0}, {ld0, ld1, ld2, ld3}, {c0, c1, c2, c3}};
    const int64_t blocks = std::min<int64_t>(cdiv(rows * total, 256), 4096);
    hipLaunchKernelGGL(concat_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p, dst, dst_ld, rows, total);
    return check_launch("concat");
}

__global__ void scale_rows_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ out, int64_t ldo,
                                  int64_t rows, int cols, const float* __restrict__ scale_dev, float scale_host) {
    const float sc = scale_host * (scale_dev ? scale_dev[0] : 1.f);
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i % cols);
        out[r * ldo + c] = x[r * ldx + c] * sc;
    }
}

extern "C" int pcvae_scale_rows(const float* x, int64_t ldx, float* out, int64_t ldo, int64_t rows, int cols,
                                const float* scale_dev, float scale_host, pcvae_stream_t stream) {
    PCVAE_REQUIRE(x && out && cols > 0 && ldx >= cols && ldo >= cols, "scale_rows: bad arguments");
    if (rows == 0) return PCVAE_OK;
    const int64_t blocks = std::min<int64_t>(cdiv(rows * cols, 256), 4096);
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, ldx, out, ldo,
                       rows, cols, scale_dev, scale_host);
    return check_launch("scale_rows");
}

__global__ void leaky_bwd_kernel(float* __restrict__ g, int64_t ldg, const float* __restrict__ y, int64_t ldy,
                                 int64_t rows, int cols) {
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i % cols);
        if (!(y[r * ldy + c] > 0.f)) g[r * ldg + c] *= kLeakySlope;
    }
}

extern "C" int pcvae_leaky_bwd(float* g, int64_t ldg, const float* y, int64_t ldy, int64_t rows, int cols,
                               pcvae_stream_t stream) {
    PCVAE_REQUIRE(g && y && cols > 0 && ldg >= cols && ldy >= cols, "leaky_bwd: bad arguments");
    if (rows == 0) return PCVAE_OK;
    const int64_t blocks = std::min<int64_t>(cdiv(rows * cols, 256), 2048);
    hipLaunchKernelGGL(leaky_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), g, ldg, y, ldy, rows,
                       cols);
    return check_launch("leaky_bwd");
}

// =============================================================================================
// K4: reparameterisation.  eps from Philox4x32-10 + Box-Muller; element e uses Philox counter
// (offset + e) / 4, lane (offset + e) % 4 -> independent of launch geometry and of rank sharding.
// =============================================================================================
__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t e) {
    const uint64_t ctr = e >> 2;
    const Philox4 p = philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0x5043564Au /*"PCVJ"*/, 0u, (uint32_t)seed,
                                    (uint32_t)(seed >> 32));
    const int j = (int)(e & 3);
    const uint32_t a = (j < 2) ? p.x : p.z;
    const uint32_t b = (j < 2) ? p.y : p.w;
    const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);           // [0, 1)
    const float rad = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincosf(6.28318530717958647692f * u2, &s, &c);
    return rad * ((j & 1) ? s : c);
}

__global__ void reparam_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                                   const float* __restrict__ eps_in, uint64_t seed, uint64_t offset,
                                   float* __restrict__ z, int64_t ldz, float* __restrict__ eps_out, int64_t B, int Z) {
    const int64_t n = B * Z;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float eps = eps_in ? eps_in[i] : philox_normal(seed, offset + (uint64_t)i);
        const float sd = expf(0.5f * logvar[i]);
        z[(i / Z) * ldz + (i % Z)] = eps * sd + mu[i];
        if (eps_out) eps_out[i] = eps;
    }
}

extern "C" int pcvae_reparam_fwd(const float* mu, const float* logvar, const float* eps_in, uint64_t seed,
                                 uint64_t offset, float* z, int64_t ldz, float* eps_out, int64_t B, int Z,
                                 pcvae_stream_t stream) {
    PCVAE_REQUIRE(mu && logvar && z && Z > 0 && ldz >= Z, "reparam_fwd: bad arguments");
    if (B == 0) return PCVAE_OK;
    const int64_t blocks = std::min<int64_t>(cdiv(B * Z, 256), 2048);
    hipLaunchKernelGGL(reparam_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), mu, logvar, eps_in,
                       seed, offset, z, ldz, eps_out, B, Z);
    return check_launch("reparam_fwd");
}

__global__ void philox_normal_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = philox_normal(seed, offset + (uint64_t)i);
}

extern "C" int pcvae_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, pcvae_stream_t stream) {
    PCVAE_REQUIRE(n >= 0 && (out || n == 0), "philox_normal: bad arguments");
    if (n == 0) return PCVAE_OK;
    const int64_t blocks = std::min<int64_t>(cdiv(n, 256), 2048);
    hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), out, n, seed, offset);
    return check_launch("philox_normal");
}

__global__ void reparam_bwd_kernel(const float* __restrict__ dz, int64_t lddz, const float* __restrict__ eps,
                                   const float* __restrict__ logvar, float* __restrict__ dmu,
                                   float* __restrict__ dlogvar, int64_t B, int Z) {
    const int64_t n = B * Z;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float g = dz[(i / Z) * lddz + (i % Z)];
        dmu[i] += g;
        dlogvar[i] += g * eps[i] * 0.5f * expf(0.5f * logvar[i]);
    }
}

extern "C" int pcvae_reparam_bwd(const float* dz, int64_t lddz, const float* eps, const float* logvar, float* dmu,
                                 float* dlogvar, int64_t B, int Z, pcvae_stream_t stream) {
    PCVAE_REQUIRE(dz && eps && logvar && dmu && dlogvar && Z > 0 && lddz >= Z, "reparam_bwd: bad arguments");
    if (B == 0) return PCVAE_OK;
    const int64_t blocks = std::min<int64_t>(cdiv(B * Z, 256), 2048);
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), dz, lddz, eps,
                       logvar, dmu, dlogvar, B, Z);
    return check_launch("reparam_bwd");
}

// =============================================================================================
// deterministic single-block reductions (K7 forward, CE mean)
// =============================================================================================
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float block_sum_1024(float v) {
    __shared__ float part[16];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x < 64) {
        t = threadIdx.x < (blockDim.x >> 6) ? part[threadIdx.x] : 0.f;
        t = wave_sum(t);
    }
    return t;  // valid in wave 0
}

// Deterministic in (n, grid) - a fixed slice per block, partials combined in block order by whichever block finishes last - and
// spread over up to 64 CUs: one block streaming the four [B, Z] arrays alone (2 MB at B = 8192) took 45 us, a latency-bound
// 46 GB/s.  The partials live in a small device-global scratch: calls on different streams must not overlap (the trainer issues
// everything on one stream).
constexpr int KLD_MAX_BLOCKS = 64;
__device__ float g_kld_partial[KLD_MAX_BLOCKS];
__device__ unsigned int g_kld_done = 0;

// block partial t (valid in thread 0) -> out[0] = -0.5 * sum over blocks, combined in block order by the block that finishes last
__device__ __forceinline__ void kld_combine(const float t, float* __restrict__ out) {
    if (gridDim.x == 1) {   // one block (small batches): nothing to combine - no partial, no arrival, no second round trip
        if (threadIdx.x == 0) out[0] = -0.5f * t;
        return;
    }
    __shared__ bool last;
    // device-scope (sc1) store / loads of the partials instead of __threadfence(): a device-scope fence writes back and invalidates
    // the whole L2 of the XCD, once per block - 10 of this kernel's 16 us at config 4.  The explicit s_waitcnt vmcnt(0) holds the
    // arrival count back until the store is acknowledged at the memory side (a workgroup-scope release emits no such wait on
    // gfx950: without it the last block could read the partial of a previous call).
    if (threadIdx.x == 0) {
        __hip_atomic_store(&g_kld_partial[blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        last = atomicAdd(&g_kld_done, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last && threadIdx.x < 64) {   // one wave: partial b in lane b, a fixed butterfly - deterministic, one memory round trip
        static_assert(KLD_MAX_BLOCKS <= 64, "one partial per lane");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        float tot = threadIdx.x < gridDim.x ? __hip_atomic_load(&g_kld_partial[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                            : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
        if (threadIdx.x == 0) {
            out[0] = -0.5f * tot;
            atomicExch(&g_kld_done, 0u);
        }
    }
}

__global__ void __launch_bounds__(1024) kld_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                       const float* __restrict__ pmu, const float* __restrict__ plv,
                                                       int64_t n, float* __restrict__ out) {
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(mu) | reinterpret_cast<uintptr_t>(lv) | reinterpret_cast<uintptr_t>(pmu) |
                         reinterpret_cast<uintptr_t>(plv)) & 15) == 0 ? n / 4 : 0;
    const float4* mu4 = reinterpret_cast<const float4*>(mu);
    const float4* lv4 = reinterpret_cast<const float4*>(lv);
    const float4* pmu4 = reinterpret_cast<const float4*>(pmu);
    const float4* plv4 = reinterpret_cast<const float4*>(plv);
    // block b owns float4 elements [b * per, (b + 1) * per); the scalar tail belongs to the last block
    const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
    float a4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const float4 m = mu4[i], l = lv4[i], pm = pmu4[i], pl = plv4[i];
        const float d0 = m.x - pm.x, d1 = m.y - pm.y, d2 = m.z - pm.z, d3 = m.w - pm.w;
        a4[0] += 1.f + l.x - pl.x - (expf(l.x) + d0 * d0) / expf(pl.x);
        a4[1] += 1.f + l.y - pl.y - (expf(l.y) + d1 * d1) / expf(pl.y);
        a4[2] += 1.f + l.z - pl.z - (expf(l.z) + d2 * d2) / expf(pl.z);
        a4[3] += 1.f + l.w - pl.w - (expf(l.w) + d3 * d3) / expf(pl.w);
    }
    float acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    if (blockIdx.x == gridDim.x - 1)
        for (int64_t i = 4 * n4 + threadIdx.x; i < n; i += blockDim.x) {
            const float d = mu[i] - pmu[i];
            acc += 1.f + lv[i] - plv[i] - (expf(lv[i]) + d * d) / expf(plv[i]);
        }
    kld_combine(block_sum_1024(acc), out);
}

extern "C" int pcvae_kld_fwd(const float* mu, const float* lv, const float* pmu, const float* plv, int64_t n,
                             float* kld_out, pcvae_stream_t stream) {
    PCVAE_REQUIRE(mu && lv && pmu && plv && kld_out && n >= 0, "kld_fwd: bad arguments");
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(KLD_MAX_BLOCKS, cdiv(n / 4, 1024)));
    hipLaunchKernelGGL(kld_fwd_kernel, dim3((unsigned)blocks), dim3(1024), 0, as_stream(stream), mu, lv, pmu, plv, n, kld_out);
    return check_launch("kld_fwd");
}

__global__ void kld_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                               const float* __restrict__ pmu, const float* __restrict__ plv, int64_t n,
                               const float* __restrict__ scale_dev, float scale_host, float* __restrict__ dmu,
                               float* __restrict__ dlv, float* __restrict__ dpmu, float* __restrict__ dplv) {
    const float sc = scale_host * (scale_dev ? scale_dev[0] : 1.f);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float ip = expf(-plv[i]);
        const float d = mu[i] - pmu[i];
        const float ev = expf(lv[i]);
        if (dmu) dmu[i] += sc * d * ip;
        if (dlv) dlv[i] += sc * (-0.5f) * (1.f - ev * ip);
        if (dpmu) dpmu[i] += sc * (-d * ip);
        if (dplv) dplv[i] += sc * (-0.5f) * (-1.f + (ev + d * d) * ip);
    }
}

extern "C" int pcvae_kld_bwd(const float* mu, const float* lv, const float* pmu, const float* plv, int64_t n,
                             const float* scale_dev, float scale_host, float* dmu, float* dlv, float* dpmu,
                             float* dplv, pcvae_stream_t stream) {
    PCVAE_REQUIRE(mu && lv && pmu && plv && n >= 0, "kld_bwd: bad arguments");
    if (n == 0) return PCVAE_OK;
    const int64_t blocks = std::min<int64_t>(cdiv(n, 256), 2048);
    hipLaunchKernelGGL(kld_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), mu, lv, pmu, plv, n,
                       scale_dev, scale_host, dmu, dlv, dpmu, dplv);
    return check_launch("kld_bwd");
}

// K4 + K7 backward in one launch: the posterior's (mu, logvar) receive a gradient from the reparameterisation and from the KL
// term; this kernel WRITES the four gradients (no zero-fill, no accumulation, no autograd add afterwards).
//   dmu   = dz + s d / exp(plv)                 dlv  = dz eps exp(lv / 2) / 2 - s (1 - exp(lv) / exp(plv)) / 2
//   dpmu  = -s d / exp(plv)                     dplv = -s (-1 + (exp(lv) + d^2) / exp(plv)) / 2        d = mu - pmu
__global__ void latent_bwd_kernel(const float* __restrict__ dz, int64_t lddz, const float* __restrict__ eps,
                                  const float* __restrict__ mu, const float* __restrict__ lv, const float* __restrict__ pmu,
                                  const float* __restrict__ plv, const float* __restrict__ dkld_dev, float dkld_host,
                                  float* __restrict__ dmu, float* __restrict__ dlv, float* __restrict__ dpmu,
                                  float* __restrict__ dplv, int64_t B, int Z) {
    const float sc = dkld_host * (dkld_dev ? dkld_dev[0] : 1.f);
    const int64_t n = B * Z;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float g = dz[(i / Z) * lddz + (i % Z)];
        const float ip = expf(-plv[i]);
        const float d = mu[i] - pmu[i];
        const float ev = expf(lv[i]);
        dmu[i] = g + sc * d * ip;
        dlv[i] = g * eps[i] * 0.5f * expf(0.5f * lv[i]) + sc * (-0.5f) * (1.f - ev * ip);
        dpmu[i] = sc * (-d * ip);
        dplv[i] = sc * (-0.5f) * (-1.f + (ev + d * d) * ip);
    }
}

extern "C" int pcvae_latent_bwd(const float* dz, int64_t lddz, const float* eps, const float* mu, const float* lv,
                                const float* pmu, const float* plv, const float* dkld_dev, float dkld_host, float* dmu,
                                float* dlv, float* dpmu, float* dplv, int64_t B, int Z, pcvae_stream_t stream) {
    PCVAE_REQUIRE(dz && eps && mu && lv && pmu && plv && dmu && dlv && dpmu && dplv && Z > 0 && lddz >= Z && B >= 0,
                  "latent_bwd: bad arguments");
    if (B == 0) return PCVAE_OK;
    const int64_t blocks = std::min<int64_t>(cdiv(B * Z, 256), 2048);
    hipLaunchKernelGGL(latent_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), dz, lddz, eps, mu, lv, pmu,
                       plv, dkld_dev, dkld_host, dmu, dlv, dpmu, dplv, B, Z);
    return check_launch("latent_bwd");
}

// ---- the same two operations on PACKED head outputs: y_enc = [mu | logvar], y_prior = [pmu | plogvar], both [B, 2 Z] with leading
// dimension ld - what ONE N = 2 Z GEMM per stack produces when the two heads' weights are adjacent (models/pivotcvae.py:
// 170-173, 236-239 are two nn.Linear each).  Forward: z (into a column window of the slate-completion input) + eps + the KL sum in
// one launch; backward: the packed gradients [dmu | dlogvar], [dpmu | dplogvar] in one launch.
__global__ void __launch_bounds__(1024) latent_fwd_packed_kernel(const float* __restrict__ y_enc, const float* __restrict__ y_prior,
                                                                 int64_t ld, const float* __restrict__ eps_in, uint64_t seed,
                                                                 uint64_t offset, float* __restrict__ z, int64_t ldz,
                                                                 float* __restrict__ eps_out, float* __restrict__ kld_out,
                                                                 int64_t B, int Z) {
    const int64_t n = B * Z;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    float acc = 0.f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const int64_t b = i / Z;
        const int k = (int)(i - b * Z);
        const float mu = y_enc[b * ld + k], lv = y_enc[b * ld + Z + k];
        const float pmu = y_prior[b * ld + k], plv = y_prior[b * ld + Z + k];
        const float eps = eps_in ? eps_in[i] : philox_normal(seed, offset + (uint64_t)i);
        z[b * ldz + k] = eps * expf(0.5f * lv) + mu;
        eps_out[i] = eps;
        const float d = mu - pmu;
        acc += 1.f + lv - plv - (expf(lv) + d * d) / expf(plv);
    }
    kld_combine(block_sum_1024(acc), kld_out);
}

extern "C" int pcvae_latent_fwd_packed(const float* y_enc, const float* y_prior, int64_t ld, const float* eps_in, uint64_t seed,
                                       uint64_t offset, float* z, int64_t ldz, float* eps_out, float* kld_out, int64_t B, int Z,
                                       pcvae_stream_t stream) {
    PCVAE_REQUIRE(y_enc && y_prior && z && eps_out && kld_out && Z > 0 && ld >= 2 * Z && ldz >= Z && B > 0,
                  "latent_fwd_packed: bad arguments");
    // one element per thread up to 64 blocks: the kernel's loop pays a memory round trip per iteration (a single block walking the
    // 16 384 elements of config 2 in 16 iterations: 30 us; 8 blocks of 2 iterations: 9.2 us), the cross-block combine about 3 us once
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(KLD_MAX_BLOCKS, cdiv(B * Z, 1024)));
    hipLaunchKernelGGL(latent_fwd_packed_kernel, dim3((unsigned)blocks), dim3(1024), 0, as_stream(stream), y_enc, y_prior, ld, eps_in,
                       seed, offset, z, ldz, eps_out, kld_out, B, Z);
    return check_launch("latent_fwd_packed");
}

__global__ void latent_bwd_packed_kernel(const float* __restrict__ dz, int64_t lddz, const float* __restrict__ eps,
                                         const float* __restrict__ y_enc, const float* __restrict__ y_prior, int64_t ld,
                                         const float* __restrict__ dkld_dev, float dkld_host, float* __restrict__ g_enc,
                                         float* __restrict__ g_prior, int64_t ldg, int64_t B, int Z) {
    const float sc = dkld_host * (dkld_dev ? dkld_dev[0] : 1.f);
    const int64_t n = B * Z;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / Z;
        const int k = (int)(i - b * Z);
        const float mu = y_enc[b * ld + k], lv = y_enc[b * ld + Z + k];
        const float pmu = y_prior[b * ld + k], plv = y_prior[b * ld + Z + k];
        const float g = dz ? dz[b * lddz + k] : 0.f;
        const float ip = expf(-plv), d = mu - pmu, ev = expf(lv);
        g_enc[b * ldg + k] = g + sc * d * ip;
        g_enc[b * ldg + Z + k] = g * eps[i] * 0.5f * expf(0.5f * lv) + sc * (-0.5f) * (1.f - ev * ip);
        g_prior[b * ldg + k] = sc * (-d * ip);
        g_prior[b * ldg + Z + k] = sc * (-0.5f) * (-1.f + (ev + d * d) * ip);
    }
}

extern "C" int pcvae_latent_bwd_packed(const float* dz, int64_t lddz, const float* eps, const float* y_enc, const float* y_prior,
                                       int64_t ld, const float* dkld_dev, float dkld_host, float* g_enc, float* g_prior,
                                       int64_t ldg, int64_t B, int Z, pcvae_stream_t stream) {
    PCVAE_REQUIRE(eps && y_enc && y_prior && g_enc && g_prior && Z > 0 && ld >= 2 * Z && ldg >= 2 * Z && (!dz || lddz >= Z) && B > 0,
                  "latent_bwd_packed: bad arguments");
    const int64_t blocks = std::min<int64_t>(cdiv(B * Z, 256), 2048);
    hipLaunchKernelGGL(latent_bwd_packed_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), dz, lddz, eps, y_enc, y_prior,
                       ld, dkld_dev, dkld_host, g_enc, g_prior, ldg, B, Z);
    return check_launch("latent_bwd_packed");
}

// =============================================================================================
// K1 + K2 + the reference's torch.cat's in one launch: everything the three MLP stacks read that does not depend on a weight
// (models/pivotcvae.py:250-258 get_condition / docEmbed / userEmbed, :166, :201, :213, :231 the concatenations, :194 the
// ground-truth pivot row) written straight into the stacks' input buffers - one wave per slate:
//     enc_in  [B, S D + C (+ D)] = item rows | one-hot click count | user row
//     prior_in[B, C (+ D)]       =             one-hot click count | user row
//     scm_in  [B, Z + C + D (+D)]= (z: left for the latent kernel) | click count | pivot row E[s[b, 0]] | user row
//     rx      [B, S D]           = pivot row | (slate-completion output: written by that stack's last GEMM)
// replaces condition + two gathers + four concat launches and the copies they make.
// =============================================================================================
__global__ void __launch_bounds__(256) assemble_inputs_kernel(const float* __restrict__ E, const float* __restrict__ U,
                                                              const int64_t* __restrict__ s, const float* __restrict__ r,
                                                              const int64_t* __restrict__ u, int64_t B, int S, int D, int ncols, int Z,
                                                              float* __restrict__ enc_in, int64_t ld_enc, float* __restrict__ prior_in,
                                                              int64_t ld_prior, float* __restrict__ scm_in, int64_t ld_scm,
                                                              float* __restrict__ rx, int64_t ld_rx) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int C = S + 1;
    float cnt = 0.f;
    for (int j = 0; j < ncols; ++j) cnt += r[b * ncols + j];
    const int c1 = (int)cnt;   // torch: sum(r).to(long) truncates
    float* enc = enc_in + b * ld_enc;
    float* pri = prior_in + b * ld_prior;
    float* scm = scm_in + b * ld_scm;
    for (int j = 0; j < S; ++j) {
        const float* src = E + s[b * S + j] * (int64_t)D;
        for (int d = lane; d < D; d += 64) {
            const float v = src[d];
            enc[j * D + d] = v;
            if (j == 0) { scm[Z + C + d] = v; rx[b * ld_rx + d] = v; }
        }
    }
    if (lane < C) {
        const float v = lane == c1 ? 1.f : 0.f;
        enc[S * D + lane] = v;
        pri[lane] = v;
        scm[Z + lane] = v;
    }
    if (U) {
        const float* src = U + u[b] * (int64_t)D;
        for (int d = lane; d < D; d += 64) {
            const float v = src[d];
            enc[S * D + C + d] = v;
            pri[C + d] = v;
            scm[Z + C + D + d] = v;
        }
    }
}

// D / 4 a divisor of 256: the same work as 16-byte row chunks, one chunk per thread and UNROLL of them in flight (a slate is S + 1 table rows
// of D / 4 chunks; chunk g of the launch belongs to slate g / (rows * cpr)), so that the launch is many short, independent index ->
// row -> store chains instead of one wave walking a slate: the scalar kernel above keeps one 256-byte request per wave in flight
// and runs at 0.42 of the HBM peak on config 4; this one at 0.48: 110.9 MB in 28.8 us - non-temporal accesses, a wave per slate
// with six chunks per lane in flight and 16-byte aligned row starts were measured and change nothing.  Destination rows start on 4-byte boundaries only (row widths like 1419):
// dword-aligned 16-byte accesses, which gfx950 takes.  The lanes that hold a slate's first row also write its one-hot
// click count.
#ifndef ASSEMBLE_NT_LOAD
#define ASSEMBLE_NT_LOAD 1
#endif
template <int UNROLL>
__global__ void __launch_bounds__(256) assemble_inputs_vec_kernel(const float* __restrict__ E, const float* __restrict__ U,
                                                                  const int64_t* __restrict__ s, const float* __restrict__ r,
                                                                  const int64_t* __restrict__ u, int64_t B, int S, int D, int ncols,
                                                                  int Z, float* __restrict__ enc_in, int64_t ld_enc,
                                                                  float* __restrict__ prior_in, int64_t ld_prior,
                                                                  float* __restrict__ scm_in, int64_t ld_scm, float* __restrict__ rx,
                                                                  int64_t ld_rx) {
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    // a workgroup copies 256 / cpr x UNROLL consecutive table rows (row id = slate * rows-per-slate + j); a thread keeps one
    // chunk column c4 and walks rows: 32-bit index arithmetic only (64-bit divisions per chunk cost more than the copy)
    const int C = S + 1, cpr = D >> 2, rps = S + (U ? 1 : 0), rpw = 256 / cpr;   // rows per slate, rows per workgroup pass
    const unsigned total_rows = (unsigned)(B * rps);
    const int c4 = ((int)threadIdx.x % cpr) << 2;
    const unsigned row0 = blockIdx.x * (unsigned)(rpw * UNROLL) + threadIdx.x / cpr;
    f32x4u v[UNROLL];
    int bb[UNROLL], jj[UNROLL], jc[UNROLL];
    int64_t id[UNROLL];
    float r0[UNROLL];
    // three branch-free phases, so that the UNROLL index loads, then the UNROLL row loads, are in flight together (rows past the
    // end are clamped for the loads and skipped by the stores): as one loop hipcc serialises index -> row -> index -> row
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
        const unsigned row = row0 + k * rpw, rc = row < total_rows ? row : total_rows - 1;
        const int b = (int)(rc / (unsigned)rps), j = (int)(rc - (unsigned)b * rps);
        id[k] = j < S ? s[(int64_t)b * S + j] : u[b];
        // the first click flags of the slate, with the index loads (behind the stores they were a third exposed round trip)
        r0[k] = (j == 0 && (c4 >> 2) < ncols) ? r[(int64_t)b * ncols + (c4 >> 2)] : 0.f;
        bb[k] = b;
        jc[k] = j;
        jj[k] = row < total_rows ? j : -1;
    }
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
        const float* src = (jc[k] < S ? E : U) + id[k] * (int64_t)D;   // (rows past the end re-read the last row)
        // table rows are read once and are 16-byte aligned (D % 4 == 0, checked by the host): a non-temporal load - the standalone
        // gather measures 21.5 us with it against 31.5 us without (round 3)
        const f32x4 q = ASSEMBLE_NT_LOAD ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + c4))
                                         : *reinterpret_cast<const f32x4*>(src + c4);
        v[k] = q;
    }
#ifndef ASSEMBLE_NT_STORE
#define ASSEMBLE_NT_STORE 0
#endif
#define ASM_ST(ptr_, val_)                                                                                   \
    do {                                                                                                     \
        if (ASSEMBLE_NT_STORE) __builtin_nontemporal_store((val_), reinterpret_cast<f32x4u*>(ptr_));         \
        else *reinterpret_cast<f32x4u*>(ptr_) = (val_);                                                      \
    } while (0)
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
        const int j = jj[k];
        if (j < 0) continue;
        const int64_t b = bb[k];
        float* enc = enc_in + b * ld_enc;
        float* pri = prior_in + b * ld_prior;
        float* scm = scm_in + b * ld_scm;
        if (j < S) {
            ASM_ST(enc + (int64_t)j * D + c4, v[k]);
            if (j == 0) {
                ASM_ST(scm + Z + C + c4, v[k]);
                ASM_ST(rx + b * ld_rx + c4, v[k]);
            }
        } else {
            ASM_ST(enc + (int64_t)S * D + C + c4, v[k]);
            ASM_ST(pri + C + c4, v[k]);
            ASM_ST(scm + Z + C + D + c4, v[k]);
        }
        if (j == 0) {   // this slate's one-hot click count, by the cpr lanes that hold its first row (a lane walking ncols
                        // dependent loads alone held its wave for ten memory round trips)
            const int c = c4 >> 2;
            float cnt = r0[k];
            for (int i = c + cpr; i < ncols; i += cpr) cnt += r[b * ncols + i];
            for (int o = cpr >> 1; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
            const int c1 = (int)cnt;   // torch: sum(r).to(long) truncates
            for (int i = c; i < C; i += cpr) {
                const float o = i == c1 ? 1.f : 0.f;
                enc[S * D + i] = o;
                pri[i] = o;
                scm[Z + i] = o;
            }
        }
    }
}

// (Round 4, measured and dropped: the batch shape of gather_rows_coal_kernel here - 16 rows in flight per lane behind ONE coalesced
// index load per wave: 28.1 us against 27.4 us for this kernel on the same box.  The launch is bound by its stores - 7.8 KB written per
// slate against 5.6 KB read, into rows that start on 4-byte boundaries (row widths like 1419 floats) - not by its index round trips.)
extern "C" int pcvae_assemble_inputs(const float* E, int64_t n_items, const float* U, int64_t n_users, const int64_t* s, const float* r,
                                     const int64_t* u, int64_t B, int S, int D, int ncols, int Z, float* enc_in, int64_t ld_enc,
                                     float* prior_in, int64_t ld_prior, float* scm_in, int64_t ld_scm, float* rx, int64_t ld_rx,
                                     pcvae_stream_t stream) {
    PCVAE_REQUIRE(E && s && r && enc_in && prior_in && scm_in && rx && (!U || u), "assemble_inputs: null pointer");
    const int C = S + 1, ud = U ? D : 0;
    PCVAE_REQUIRE(B >= 0 && S > 0 && S < 64 && D > 0 && ncols > 0 && Z > 0 && n_items > 0 && (!U || n_users > 0),
                  "assemble_inputs: bad shape B=%lld S=%d D=%d", (long long)B, S, D);
    PCVAE_REQUIRE(ld_enc >= (int64_t)S * D + C + ud && ld_prior >= C + ud && ld_scm >= (int64_t)Z + C + D + ud && ld_rx >= (int64_t)S * D,
                  "assemble_inputs: a leading dimension is narrower than its row");
    if (B == 0) return PCVAE_OK;
    const int cpr = D / 4;
    if (D % 4 == 0 && cpr <= 64 && 64 % cpr == 0 && B * (S + 1) < (1LL << 31) && ((uintptr_t)E % 16 == 0) &&
        (!U || (uintptr_t)U % 16 == 0)) {   // a row = an aligned group of <= 64 lanes, 16-byte aligned in its table
        PCVAE_LAUNCH_TIMED(PCVAE_TIMER_ASSEMBLE, assemble_inputs_vec_kernel<4>, dim3((unsigned)cdiv(B * (S + (U ? 1 : 0)), (256 / cpr) * 4)),
                           dim3(256), 0, as_stream(stream), E, U, s, r, u, B, S, D, ncols, Z, enc_in, ld_enc, prior_in, ld_prior, scm_in,
                           ld_scm, rx, ld_rx);
    } else
        hipLaunchKernelGGL(assemble_inputs_kernel, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, as_stream(stream), E, U, s, r, u, B, S, D,
                           ncols, Z, enc_in, ld_enc, prior_in, ld_prior, scm_in, ld_scm, rx, ld_rx);
    return check_launch("assemble_inputs");
}

__global__ void __launch_bounds__(1024) sum_kernel(const float* __restrict__ x, int64_t n, float scale,
                                                   float* __restrict__ out) {
    float acc = 0.f;
    const int64_t n4 = (reinterpret_cast<uintptr_t>(x) & 15) == 0 ? n / 4 : 0;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int64_t i = threadIdx.x; i < n4; i += blockDim.x) {
        const float4 v = x4[i];
        a4[0] += v.x; a4[1] += v.y; a4[2] += v.z; a4[3] += v.w;
    }
    acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    for (int64_t i = 4 * n4 + threadIdx.x; i < n; i += blockDim.x) acc += x[i];
    const float t = block_sum_1024(acc);
    if (threadIdx.x == 0) out[0] = t * scale;
}

extern "C" int pcvae_sum(const float* x, int64_t n, float scale, float* out, pcvae_stream_t stream) {
    PCVAE_REQUIRE(x && out && n >= 0, "sum: bad arguments");
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, as_stream(stream), x, n, scale, out);
    return check_launch("sum");
}

// the logged ELBO terms of one step as one 3-float record: out = (rec + beta * kld, rec, kld).  A data-parallel rank writes its
// record into the tail of the flat gradient buffer, so the gradient all-reduce sums the statistics too (train_generative.py:62).
__global__ void elbo_pack_kernel(const float* __restrict__ rec, const float* __restrict__ kld, float beta, float* __restrict__ out) {
    if (threadIdx.x == 0) {
        const float r = rec[0], k = kld[0];
        out[0] = r + beta * k;
        out[1] = r;
        out[2] = k;
    }
}

extern "C" int pcvae_elbo_pack(const float* rec, const float* kld, float beta, float* out, pcvae_stream_t stream) {
    PCVAE_REQUIRE(rec && kld && out, "elbo_pack: bad arguments");
    hipLaunchKernelGGL(elbo_pack_kernel, dim3(1), dim3(64), 0, as_stream(stream), rec, kld, beta, out);
    return check_launch("elbo_pack");
}

// zero a buffer on the stream (optimizer.zero_grad(), train_generative.py:124).  A KERNEL, not hipMemsetAsync: captured into a hipGraph
// the memset node is not ordered against its neighbours on ROCm 7.2 - replayed right behind an eager Adam launch on the same stream it
// cleared gradients that Adam was still reading, or a later node's writes (round 3: hipGraph-replayed training at config 4 drifted
// off the eager trajectory, KLD 611.33 against 533.36 after six steps, only when the host was NOT running ahead of the GPU;
// tests/test_hip_model_golden.py::test_graph_replay_after_a_host_sync_follows_the_eager_trajectory).
__global__ void __launch_bounds__(256) zero_kernel(uint4* __restrict__ p16, size_t n16, unsigned char* __restrict__ tail, size_t ntail) {
    const uint4 z = {0u, 0u, 0u, 0u};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p16[i] = z;
    if (blockIdx.x == 0 && threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

// the step-dependent words (mask / candidate seed, sampler stream position) of a hipGraph-replayed step: written by this one-thread
// kernel BEFORE the replay, read by the captured kernels through their seed_dev / row_offset_dev arguments
__global__ void set_words_kernel(uint64_t* dst, uint64_t a, uint64_t b) {
    dst[0] = a;
    dst[1] = b;
}

extern "C" int pcvae_set_words(uint64_t* dst, uint64_t a, uint64_t b, pcvae_stream_t stream) {
    PCVAE_REQUIRE(dst && ((uintptr_t)dst % 8 == 0), "set_words: null or misaligned pointer");
    hipLaunchKernelGGL(set_words_kernel, dim3(1), dim3(1), 0, as_stream(stream), dst, a, b);
    return check_launch("set_words");
}

extern "C" int pcvae_zero(void* p, size_t nbytes, pcvae_stream_t stream) {
    PCVAE_REQUIRE(p || nbytes == 0, "zero: bad arguments");
    if (nbytes == 0) return PCVAE_OK;
    unsigned char* b = static_cast<unsigned char*>(p);
    // bytes in front of the first 16-byte boundary and behind the last whole 16-byte word (at most 15 each) go one per thread
    const size_t head = std::min<size_t>(nbytes, (16 - (reinterpret_cast<uintptr_t>(b) & 15)) & 15);
    if (head) hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(256), 0, as_stream(stream), nullptr, (size_t)0, b, head);
    const size_t n16 = (nbytes - head) / 16, ntail = (nbytes - head) % 16;
    if (n16 || ntail) {
        const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>(cdiv((int64_t)n16, 256 * 4), 2048));
        hipLaunchKernelGGL(zero_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), reinterpret_cast<uint4*>(b + head), n16,
                           b + head + 16 * n16, ntail);
    }
    return check_launch("zero");
}

// =============================================================================================
// in-loop evaluation helpers (response model)
// =============================================================================================
__global__ void __launch_bounds__(256) normalize_rows_kernel(float* __restrict__ x, int64_t ldx, int64_t rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per row
    if (r >= rows) return;
    float ss = 0.f;
    for (int c = lane; c < cols; c += 64) { const float v = x[r * ldx + c]; ss += v * v; }
    ss = wave_sum(ss);
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    for (int c = lane; c < cols; c += 64) x[r * ldx + c] *= inv;
}

extern "C" int pcvae_normalize_rows(float* x, int64_t ldx, int64_t rows, int cols, pcvae_stream_t stream) {
    PCVAE_REQUIRE(cols > 0 && ldx >= cols && rows >= 0 && (x || rows == 0), "normalize_rows: bad arguments");
    if (rows == 0) return PCVAE_OK;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, as_stream(stream), x, ldx, rows, cols);
    return check_launch("normalize_rows");
}

__global__ void __launch_bounds__(1024) click_stats_kernel(const float* __restrict__ logits, int64_t B, int S,
                                                           float* __restrict__ nc, float* __restrict__ out3) {
    __shared__ float smin[16], smax[16];
    float mn = INFINITY, mx = -INFINITY, sum = 0.f;
    for (int64_t b = threadIdx.x; b < B; b += blockDim.x) {
        float acc = 0.f;
        for (int s = 0; s < S; ++s) acc += 1.f / (1.f + expf(-logits[b * S + s]));
        if (nc) nc[b] = acc;
        mn = fminf(mn, acc); mx = fmaxf(mx, acc); sum += acc;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = mn; smax[threadIdx.x >> 6] = mx; }
    const float tot = block_sum_1024(sum);  // contains a __syncthreads()
    if (threadIdx.x == 0) {
        float a = INFINITY, b = -INFINITY;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { a = fminf(a, smin[w]); b = fmaxf(b, smax[w]); }
        out3[0] = a; out3[1] = tot / (float)B; out3[2] = b;
    }
}

extern "C" int pcvae_click_stats(const float* logits, int64_t B, int S, float* nc, float* out3, pcvae_stream_t stream) {
    PCVAE_REQUIRE(logits && out3 && B > 0 && S > 0, "click_stats: bad arguments");
    hipLaunchKernelGGL(click_stats_kernel, dim3(1), dim3(1024), 0, as_stream(stream), logits, B, S, nc, out3);
    return check_launch("click_stats");
}

__global__ void philox_randint_kernel(int64_t* __restrict__ out, int64_t n, uint64_t hi, uint64_t seed, uint64_t offset) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t e = offset + (uint64_t)i, ctr = e >> 1;
        const Philox4 p = philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0x55534552u /*"USER"*/, 0u, (uint32_t)seed,
                                        (uint32_t)(seed >> 32));
        const uint64_t u = (e & 1) ? (((uint64_t)p.z << 32) | p.w) : (((uint64_t)p.x << 32) | p.y);
        out[i] = (int64_t)(u % hi);  // 64 random bits: modulo bias < hi / 2^64
    }
}

extern "C" int pcvae_philox_randint(int64_t* out, int64_t n, int64_t hi, uint64_t seed, uint64_t offset,
                                    pcvae_stream_t stream) {
    PCVAE_REQUIRE(n >= 0 && hi > 0 && (out || n == 0), "philox_randint: bad arguments");
    if (n == 0) return PCVAE_OK;
    const int64_t blocks = std::min<int64_t>(cdiv(n, 256), 1024);
    hipLaunchKernelGGL(philox_randint_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), out, n, (uint64_t)hi,
                       seed, offset);
    return check_launch("philox_randint");
}

// =============================================================================================
// fp32 -> bf16 hi/lo split (round-to-nearest-even, NaN preserved)
// =============================================================================================
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);  // keep a NaN a (quiet) NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

__global__ void split_bf16_kernel(const float* __restrict__ src, int64_t n, uint16_t* __restrict__ hi,
                                  uint16_t* __restrict__ lo) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = src[i];
        const uint16_t h = f32_to_bf16_rne(x);
        hi[i] = h;
        if (lo) lo[i] = f32_to_bf16_rne(x - bf16_bits_to_f32(h));
    }
}

extern "C" int pcvae_split_bf16(const float* src, int64_t n, uint16_t* hi, uint16_t* lo, pcvae_stream_t stream) {
    PCVAE_REQUIRE(src && hi && n >= 0, "split_bf16: bad arguments");
    if (n == 0) return PCVAE_OK;
    const int64_t blocks = std::min<int64_t>(cdiv(n, 256), 4096);
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), src, n, hi, lo);
    return check_launch("split_bf16");
}

// [N, D] fp32 -> the table image of the bf16x3 catalog kernel.  D <= 128: [N, 2 D] bf16, row n = hi(E_n) | lo(E_n) (an item's hi
// and lo halves are one contiguous 4 D-byte row: one LDS-DMA stream, k-steps 0 .. D/32-1 multiply the hi half, the rest the lo
// half).  D = 256: TWO such images of 128 dims back to back, [2][N][256]: image i = hi | lo of dims 128 i .. 128 i + 127 (each
// image has the 512-byte rows of the D = 128 kernel; a slot of the D = 256 kernel walks both).
__global__ void split_bf16x2_kernel(const float* __restrict__ src, int64_t N, int D, uint16_t* __restrict__ out) {
    const int64_t n = N * (int64_t)D;
    const int W = D > 128 ? 128 : D;   // dims per image
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / D;
        const int col = (int)(i - row * D), img = col / W, c = col - img * W;
        const float x = src[i];
        const uint16_t h = f32_to_bf16_rne(x);
        uint16_t* o = out + ((int64_t)img * N + row) * 2 * W;
        o[c] = h;
        o[W + c] = f32_to_bf16_rne(x - bf16_bits_to_f32(h));
    }
}

extern "C" int pcvae_split_bf16x2(const float* src, int64_t N, int D, uint16_t* out, pcvae_stream_t stream) {
    PCVAE_REQUIRE(src && out && N >= 0 && D > 0 && (D <= 128 || D % 128 == 0), "split_bf16x2: bad arguments (D <= 128 or a multiple of 128)");
    if (N == 0) return PCVAE_OK;
    const int64_t blocks = std::min<int64_t>(cdiv(N * D, 256), 4096);
    hipLaunchKernelGGL(split_bf16x2_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), src, N, D, out);
    return check_launch("split_bf16x2");
}

// [N, D] fp32 (D <= 128) -> the table image of the bf16x6 catalog kernel: [N, 3 D] bf16, row n = c0 | c1 | c2 of E_n, three RNE bf16
// components whose sum is the fp32 value exactly (each difference is exact in fp32: the residual of an 8-bit rounding of a 24-bit
// significand has at most 16, then at most 8 significant bits)
__global__ void split_bf16x3_kernel(const float* __restrict__ src, int64_t N, int D, uint16_t* __restrict__ out) {
    const int64_t n = N * (int64_t)D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / D;
        const int c = (int)(i - row * D);
        float x = src[i];
        uint16_t* o = out + row * 3 * D;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const uint16_t h = f32_to_bf16_rne(x);
            o[j * D + c] = h;
            x -= bf16_bits_to_f32(h);
        }
    }
}

extern "C" int pcvae_split_bf16x3(const float* src, int64_t N, int D, uint16_t* out, pcvae_stream_t stream) {
    PCVAE_REQUIRE(src && out && N >= 0 && D > 0 && D <= 128, "split_bf16x3: bad arguments (D <= 128)");
    if (N == 0) return PCVAE_OK;
    const int64_t blocks = std::min<int64_t>(cdiv(N * D, 256), 4096);
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), src, N, D, out);
    return check_launch("split_bf16x3");
}

// =============================================================================================
// downsample on a DENSE logits tensor (train_generative.py:36-42) for callers that hold one (small catalogs; the fused losses
// never form it): out[r, n] = pred[r, n] if n == slate[r] or Bernoulli(keep_prob) else 0.  The Bernoulli stream is the one the
// dense masked CE kernels draw (Philox4x32-10 keyed by (seed, row_offset + r, n >> 2, "MASK"), word n & 3 < keep_prob * 2^32),
// restated on the host by tests/philox_ref.keep_mask.
// =============================================================================================
__global__ void downsample_dense_kernel(const float* __restrict__ pred, int64_t ldp, const int64_t* __restrict__ slate, int64_t R,
                                        int64_t N, uint32_t keep_thresh, uint64_t seed, uint64_t row_offset,
                                        float* __restrict__ out, int64_t ldo) {
    const int64_t nq = (N + 3) / 4;
    const int64_t total = R * nq;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / nq, q = i - r * nq;
        const uint64_t grow = row_offset + (uint64_t)r, nb = (uint64_t)q * 4;
        const Philox4 ph = philox4x32_10((uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)(nb >> 2),
                                         (uint32_t)(nb >> 34) ^ 0x4D41534Bu /*"MASK"*/, (uint32_t)seed, (uint32_t)(seed >> 32));
        const uint32_t w[4] = {ph.x, ph.y, ph.z, ph.w};
        const int64_t t = slate[r];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = (int64_t)nb + j;
            if (n < N) out[r * ldo + n] = (w[j] < keep_thresh || n == t) ? pred[r * ldp + n] : 0.f;
        }
    }
}

extern "C" int pcvae_downsample_dense(const float* pred, int64_t ldp, const int64_t* slate, int64_t R, int64_t N, float keep_prob,
                                      uint64_t seed, uint64_t row_offset, float* out, int64_t ldo, pcvae_stream_t stream) {
    PCVAE_REQUIRE(pred && slate && out && R >= 0 && N > 0 && ldp >= N && ldo >= N, "downsample_dense: bad arguments");
    PCVAE_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, "downsample_dense: keep_prob must be in (0, 1] (n_neg > N raises in the reference too)");
    if (R == 0) return PCVAE_OK;
    const double th = (double)keep_prob * 4294967296.0;
    const uint32_t thresh = th >= 4294967295.0 ? 0xffffffffu : (uint32_t)th;
    const int64_t blocks = std::min<int64_t>(cdiv(R * ((N + 3) / 4), 256), 4096);
    hipLaunchKernelGGL(downsample_dense_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), pred, ldp, slate, R, N, thresh,
                       seed, row_offset, out, ldo);
    return check_launch("downsample_dense");
}

// =============================================================================================
// a13: the simulators' click model as in-loop evaluator - URM / URM_P / URM_P_MR.core_forward
// (env/response_model.py:129-150, 286-295, 315-323), forward only, one wave per slate:
//   d_s   = E[slate_s] / max(||E[slate_s]||, 1e-12)                      per-item L2 normalisation
//   p_s   = sigmoid(<d_s, u> + itemBias[slate_s] + userBias[user])       u = RAW user row (the reference computes the
//                                                                        normalised one and overwrites it, :141-142)
//   URM_P    : p_s += sum_d u[d] * posDep[d * S + s] + posBias[s]        posDependentBias [S, D] REINTERPRETED as [D, S]
//                                                                        (`.view(featureSize, slateSize)`, :292 - not a transpose)
//   URM_P_MR : p_s += mr * <d_s, sigmoid(mean_s' d_s')>
// =============================================================================================
__global__ void __launch_bounds__(256) urm_forward_kernel(const float* __restrict__ E, const float* __restrict__ item_bias,
                                                          const float* __restrict__ U, const float* __restrict__ user_bias,
                                                          const int64_t* __restrict__ slates, const int64_t* __restrict__ users,
                                                          const float* __restrict__ pos_bias, const float* __restrict__ pos_dep,
                                                          float mr_factor, int use_mr, int64_t B, int S, int D,
                                                          float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int64_t uid = users[b];
    const float ub = user_bias[uid];
    auto wave_sum = [](float v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
    // pass 1: raw score per slot, mean of the normalised rows (lane owns columns lane, lane + 64, ...: D <= 256)
    float mean[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) {
        const int64_t n = slates[b * S + s];
        float e[4], ss = 0.f, dot = 0.f, pd = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int d = lane + 64 * q;
            e[q] = d < D ? E[n * D + d] : 0.f;
            ss = fmaf(e[q], e[q], ss);
        }
        const float inv = 1.f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int d = lane + 64 * q;
            if (d < D) {
                const float dn = e[q] * inv, u = U[uid * D + d];
                dot = fmaf(dn, u, dot);
                mean[q] += dn;
                if (pos_dep) pd = fmaf(u, pos_dep[(int64_t)d * S + s], pd);
            }
        }
        dot = wave_sum(dot);
        if (pos_dep) pd = wave_sum(pd);
        float sc = 1.f / (1.f + __expf(-(dot + item_bias[n] + ub)));
        if (pos_dep) sc += pd + pos_bias[s];
        if (lane == 0) out[b * S + s] = sc;
    }
    if (!use_mr) return;
    // pass 2: relation term against the slate attention sigmoid(mean of the normalised rows)
    float att[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) att[q] = 1.f / (1.f + __expf(-mean[q] / (float)S));
    for (int s = 0; s < S; ++s) {
        const int64_t n = slates[b * S + s];
        float e[4], ss = 0.f, rel = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int d = lane + 64 * q;
            e[q] = d < D ? E[n * D + d] : 0.f;
            ss = fmaf(e[q], e[q], ss);
        }
        const float inv = 1.f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (lane + 64 * q < D) rel = fmaf(e[q] * inv, att[q], rel);
        rel = wave_sum(rel);
        if (lane == 0) out[b * S + s] += mr_factor * rel;
    }
}

extern "C" int pcvae_urm_forward(const float* E, const float* item_bias, int64_t n_items, const float* U, const float* user_bias,
                                 int64_t n_users, const int64_t* slates, const int64_t* users, const float* pos_bias,
                                 const float* pos_dep, float mr_factor, int use_mr, int64_t B, int S, int D, float* out,
                                 pcvae_stream_t stream) {
    PCVAE_REQUIRE(E && item_bias && U && user_bias && slates && users && out, "urm_forward: null pointer");
    PCVAE_REQUIRE(B >= 0 && S > 0 && D > 0 && D <= 256 && n_items > 0 && n_users > 0, "urm_forward: bad shape B=%lld S=%d D=%d",
                  (long long)B, S, D);
    PCVAE_REQUIRE((pos_bias == nullptr) == (pos_dep == nullptr), "urm_forward: pos_bias and pos_dep come together");
    if (B == 0) return PCVAE_OK;
    hipLaunchKernelGGL(urm_forward_kernel, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, as_stream(stream), E, item_bias, U, user_bias,
                       slates, users, pos_bias, pos_dep, mr_factor, use_mr, B, S, D, out);
    return check_launch("urm_forward");
}

// =============================================================================================
// (f)3: candidate sets on the device (data_loader.py:46-58).  Per slate slot: Cn uniform item ids; if the slot's true item is
// among them, the target is the FIRST column holding it, else column 0 is overwritten with it and the target is 0.
// One wave per slot; element (row, c) of the raw draw is 64 Philox bits mod n_items (counter (row, c >> 1, "CAND"), row = GLOBAL
// slot index: independent of sharding); `raw` non-null replays a recorded draw instead (parity against the reference's rule).
// =============================================================================================
__global__ void __launch_bounds__(256) candidate_draw_kernel(const int64_t* __restrict__ feature, int64_t R, int64_t n_items, int Cn,
                                                             uint64_t seed, uint64_t row_offset, const int64_t* __restrict__ raw,
                                                             int64_t* __restrict__ cand, int64_t* __restrict__ tgt) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int64_t f = feature[r];
    const uint64_t grow = row_offset + (uint64_t)r;
    int first = Cn;   // lowest column holding the true item among this lane's columns
    for (int c0 = 2 * lane; c0 < Cn; c0 += 128) {   // a lane draws columns c0, c0 + 1 from one Philox call
        int64_t v[2];
        if (raw) {
            v[0] = raw[r * Cn + c0];
            v[1] = c0 + 1 < Cn ? raw[r * Cn + c0 + 1] : -1;
        } else {
            const Philox4 ph = philox4x32_10((uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)(c0 >> 1), 0x43414E44u /*"CAND"*/,
                                             (uint32_t)seed, (uint32_t)(seed >> 32));
            v[0] = (int64_t)((((uint64_t)ph.x << 32) | ph.y) % (uint64_t)n_items);
            v[1] = (int64_t)((((uint64_t)ph.z << 32) | ph.w) % (uint64_t)n_items);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (c0 + k < Cn) {
                cand[r * Cn + c0 + k] = v[k];
                if (v[k] == f && c0 + k < first) first = c0 + k;
            }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
    if (lane == 0) {
        if (first == Cn) { cand[r * Cn] = f; first = 0; }   // (lane 0 wrote column 0 itself: same-thread ordering)
        tgt[r] = first;
    }
}

extern "C" int pcvae_candidate_draw(const int64_t* feature, int64_t R, int64_t n_items, int Cn, uint64_t seed, uint64_t row_offset,
                                    const int64_t* raw, int64_t* cand, int64_t* tgt, pcvae_stream_t stream) {
    PCVAE_REQUIRE(feature && cand && tgt, "candidate_draw: null pointer");
    PCVAE_REQUIRE(R >= 0 && n_items > 0 && Cn > 0, "candidate_draw: bad shape R=%lld n_items=%lld Cn=%d", (long long)R,
                  (long long)n_items, Cn);
    if (R == 0) return PCVAE_OK;
    hipLaunchKernelGGL(candidate_draw_kernel, dim3((unsigned)cdiv(R, 4)), dim3(256), 0, as_stream(stream), feature, R, n_items, Cn,
                       seed, row_offset, raw, cand, tgt);
    return check_launch("candidate_draw");
}

// =============================================================================================
// K9 (materialised form, behind forward()'s dense-p contract; the training loss uses the fused pcvae_candidate_ce): candidate-set
// scores p[r, c] = <E[cand[r, c]], rx_r> and their backward drx_r = sum_c dp[r, c] E[cand[r, c]].  Same access pattern as the fused
// kernel: a wave owns a slate row and holds rx_r (bwd: the accumulator) in registers, a lane group of LPI lanes reads one table row
// with 16-byte loads (CS_UNR rows in flight per lane group); widths that are not a multiple of 4 take the scalar kernels.
// =============================================================================================
constexpr int CS_UNR = 4;

template <int LPI, bool BWD>
__global__ void __launch_bounds__(256) candidate_rows_kernel(const float* __restrict__ x, int64_t R, const float* __restrict__ E, int D,
                                                             const int64_t* __restrict__ cand, int Cn, float* __restrict__ out) {
    // fwd: x = rx [R, D], out = p [R, Cn].   bwd: x = dp [R, Cn], out = drx [R, D].   A lane covers chunks j, j + LPI, .. of 4 floats.
    constexpr int IPS = 64 / LPI;
    constexpr int MAXC = 4;                       // chunks per lane: D <= 16 * LPI
    const int lane = threadIdx.x & 63, j = lane % LPI, grp = lane / LPI;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int nch = D >> 2;
    float4 acc[MAXC];
#pragma unroll
    for (int q = 0; q < MAXC; ++q) {
        const int ch = j + q * LPI;
        acc[q] = (!BWD && ch < nch) ? *reinterpret_cast<const float4*>(x + r * D + 4 * ch) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int64_t* crow = cand + r * (int64_t)Cn;
    for (int c0 = 0; c0 < Cn; c0 += IPS * CS_UNR) {
        float4 e[CS_UNR][MAXC];
        float w[CS_UNR];
        bool ok[CS_UNR];
#pragma unroll
        for (int u = 0; u < CS_UNR; ++u) {
            const int c = c0 + u * IPS + grp;
            ok[u] = c < Cn;
            const float* row = E + (ok[u] ? crow[c] : 0) * (int64_t)D;
            w[u] = (BWD && ok[u]) ? x[r * (int64_t)Cn + c] : 0.f;
#pragma unroll
            for (int q = 0; q < MAXC; ++q) {
                const int ch = j + q * LPI;
                e[u][q] = (ok[u] && ch < nch) ? *reinterpret_cast<const float4*>(row + 4 * ch) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < CS_UNR; ++u) {
            if (BWD) {
#pragma unroll
                for (int q = 0; q < MAXC; ++q) {
                    acc[q].x = fmaf(w[u], e[u][q].x, acc[q].x); acc[q].y = fmaf(w[u], e[u][q].y, acc[q].y);
                    acc[q].z = fmaf(w[u], e[u][q].z, acc[q].z); acc[q].w = fmaf(w[u], e[u][q].w, acc[q].w);
                }
            } else {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < MAXC; ++q) {
                    s = fmaf(e[u][q].x, acc[q].x, s); s = fmaf(e[u][q].y, acc[q].y, s);
                    s = fmaf(e[u][q].z, acc[q].z, s); s = fmaf(e[u][q].w, acc[q].w, s);
                }
#pragma unroll
                for (int o = 1; o < LPI; o <<= 1) s += __shfl_xor(s, o, 64);
                if (ok[u] && j == 0) out[r * (int64_t)Cn + c0 + u * IPS + grp] = s;
            }
        }
    }
    if (BWD) {
#pragma unroll
        for (int q = 0; q < MAXC; ++q) {
#pragma unroll
            for (int o = LPI; o < 64; o <<= 1) {   // sum the lane groups' partial rows
                acc[q].x += __shfl_xor(acc[q].x, o, 64); acc[q].y += __shfl_xor(acc[q].y, o, 64);
                acc[q].z += __shfl_xor(acc[q].z, o, 64); acc[q].w += __shfl_xor(acc[q].w, o, 64);
            }
            const int ch = j + q * LPI;
            if (grp == 0 && ch < nch) *reinterpret_cast<float4*>(out + r * D + 4 * ch) = acc[q];
        }
    }
}

__global__ void __launch_bounds__(256) candidate_scores_scalar_kernel(const float* __restrict__ rx, int64_t R,
                                                                      const float* __restrict__ E, int D,
                                                                      const int64_t* __restrict__ cand, int Cn,
                                                                      float* __restrict__ p) {
    const int64_t total = R * (int64_t)Cn;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / Cn;
        const float* e = E + cand[i] * D;
        const float* x = rx + r * D;
        float acc = 0.f;
        for (int d = 0; d < D; ++d) acc = fmaf(e[d], x[d], acc);
        p[i] = acc;
    }
}

__global__ void __launch_bounds__(256) candidate_scores_bwd_scalar_kernel(const float* __restrict__ dp, int64_t R,
                                                                          const float* __restrict__ E, int D,
                                                                          const int64_t* __restrict__ cand, int Cn,
                                                                          float* __restrict__ drx) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r = wave; r < R; r += n_waves) {
        for (int d0 = 0; d0 < D; d0 += 64) {
            const int d = d0 + lane;
            float acc = 0.f;
            if (d < D)
                for (int c = 0; c < Cn; ++c) acc = fmaf(dp[r * Cn + c], E[cand[r * Cn + c] * D + d], acc);
            if (d < D) drx[r * D + d] = acc;
        }
    }
}

// lanes per table row: the smallest power of two >= D / 16 (a lane then holds at most 4 chunks of 4 floats), at least 2
static int cand_lpi(int D) {
    int lpi = 2;
    while (lpi * 16 < D) lpi <<= 1;
    return lpi;
}

template <bool BWD>
static int launch_candidate_rows(const float* x, int64_t R, const float* E, int D, const int64_t* cand, int Cn, float* out,
                                 hipStream_t st) {
    const dim3 grid((unsigned)cdiv(R, 4)), block(256);
    switch (cand_lpi(D)) {
        case 2: hipLaunchKernelGGL((candidate_rows_kernel<2, BWD>), grid, block, 0, st, x, R, E, D, cand, Cn, out); break;
        case 4: hipLaunchKernelGGL((candidate_rows_kernel<4, BWD>), grid, block, 0, st, x, R, E, D, cand, Cn, out); break;
        case 8: hipLaunchKernelGGL((candidate_rows_kernel<8, BWD>), grid, block, 0, st, x, R, E, D, cand, Cn, out); break;
        case 16: hipLaunchKernelGGL((candidate_rows_kernel<16, BWD>), grid, block, 0, st, x, R, E, D, cand, Cn, out); break;
        case 32: hipLaunchKernelGGL((candidate_rows_kernel<32, BWD>), grid, block, 0, st, x, R, E, D, cand, Cn, out); break;
        default: hipLaunchKernelGGL((candidate_rows_kernel<64, BWD>), grid, block, 0, st, x, R, E, D, cand, Cn, out); break;
    }
    return check_launch(BWD ? "candidate_scores_bwd" : "candidate_scores");
}

static bool cand_vec_ok(const float* a, const float* E, const float* out, int D, bool bwd) {
    return D % 4 == 0 && D <= 1024 && ((uintptr_t)E % 16 == 0) && ((uintptr_t)(bwd ? out : a) % 16 == 0);
}

extern "C" int pcvae_candidate_scores(const float* rx, int64_t R, const float* E, int64_t N, int D,
                                      const int64_t* cand, int Cn, float* p, pcvae_stream_t stream) {
    PCVAE_REQUIRE(rx && E && cand && p && D > 0 && Cn > 0 && N > 0, "candidate_scores: bad arguments");
    if (R == 0) return PCVAE_OK;
    PCVAE_REQUIRE(cdiv(R, 4) <= 2147483647LL, "candidate_scores: R too large");
    if (cand_vec_ok(rx, E, p, D, false)) return launch_candidate_rows<false>(rx, R, E, D, cand, Cn, p, as_stream(stream));
    const int64_t blocks = std::min<int64_t>(cdiv(R * Cn, 256), 256 * 16);
    hipLaunchKernelGGL(candidate_scores_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), rx, R, E, D,
                       cand, Cn, p);
    return check_launch("candidate_scores");
}

extern "C" int pcvae_candidate_scores_bwd(const float* dp, int64_t R, const float* E, int64_t N, int D,
                                          const int64_t* cand, int Cn, float* drx, pcvae_stream_t stream) {
    PCVAE_REQUIRE(dp && E && cand && drx && D > 0 && Cn > 0 && N > 0, "candidate_scores_bwd: bad arguments");
    if (R == 0) return PCVAE_OK;
    PCVAE_REQUIRE(cdiv(R, 4) <= 2147483647LL, "candidate_scores_bwd: R too large");
    if (cand_vec_ok(dp, E, drx, D, true)) return launch_candidate_rows<true>(dp, R, E, D, cand, Cn, drx, as_stream(stream));
    const int64_t blocks = std::min<int64_t>(cdiv(R, 4), 256 * 8);
    hipLaunchKernelGGL(candidate_scores_bwd_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), dp, R, E,
                       D, cand, Cn, drx);
    return check_launch("candidate_scores_bwd");
}

// =============================================================================================
// dense softmax-CE over a small class axis: one wave per row, two passes over the row (max, then sum + gradient)
// =============================================================================================
__global__ void __launch_bounds__(256) dense_ce_kernel(const float* __restrict__ p, int64_t ldp, int64_t R, int C,
                                                       const int64_t* __restrict__ target, float* __restrict__ nll,
                                                       float* __restrict__ dp, int64_t lddp) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* row = p + r * ldp;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, row[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float sum = 0.f;
    for (int c = lane; c < C; c += 64) sum += expf(row[c] - m);
    sum = wave_sum(sum);
    const int64_t t = target[r];
    const bool ok = t >= 0 && t < C;
    const float lse = m + logf(sum);
    if (lane == 0) nll[r] = ok ? lse - row[t] : NAN;
    if (dp) {
        const float inv = 1.f / sum;
        for (int c = lane; c < C; c += 64) dp[r * lddp + c] = expf(row[c] - m) * inv - (c == t ? 1.f : 0.f);
    }
}

extern "C" int pcvae_dense_ce(const float* p, int64_t ldp, int64_t R, int C, const int64_t* target, float* nll, float* dp,
                              int64_t lddp, pcvae_stream_t stream) {
    PCVAE_REQUIRE(p && target && nll && C > 0 && ldp >= C && (!dp || lddp >= C) && R >= 0, "dense_ce: bad arguments");
    if (R == 0) return PCVAE_OK;
    hipLaunchKernelGGL(dense_ce_kernel, dim3((unsigned)cdiv(R, 4)), dim3(256), 0, as_stream(stream), p, ldp, R, C, target,
                       nll, dp, lddp);
    return check_launch("dense_ce");
}

// =============================================================================================
// K8: Adam over a flat buffer (torch.optim.Adam arithmetic: lerp for m, addcmul for v, addcdiv)
// =============================================================================================
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float one_minus_b1, float b2, float one_minus_b2,
                            float eps, float step_size, float sqrt_bc2, float grad_scale, float weight_decay) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i] * grad_scale;
        if (weight_decay != 0.f) gi = fmaf(weight_decay, p[i], gi);  // torch.optim.Adam(weight_decay): L2 term in the gradient
        const float mi = m[i] + (gi - m[i]) * one_minus_b1;
        const float vi = v[i] * b2 + one_minus_b2 * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / sqrt_bc2 + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

extern "C" int pcvae_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                               float eps, int step, float grad_scale, pcvae_stream_t stream) {
    return pcvae_adam_step_l2(p, g, m, v, n, lr, b1, b2, eps, step, grad_scale, 0.f, stream);
}

extern "C" int pcvae_adam_step_l2(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                                  float eps, int step, float grad_scale, float weight_decay, pcvae_stream_t stream) {
    PCVAE_REQUIRE(p && g && m && v && n >= 0 && step >= 1 && weight_decay >= 0.f, "adam_step: bad arguments");
    if (n == 0) return PCVAE_OK;
    const double bc1 = 1.0 - pow((double)b1, step);
    const double bc2 = 1.0 - pow((double)b2, step);
    const float step_size = (float)((double)lr / bc1);
    const float sqrt_bc2 = (float)sqrt(bc2);
    const int64_t blocks = std::min<int64_t>(cdiv(n, 256), 4096);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p, g, m, v, n,
                       (float)(1.0 - (double)b1), b2, (float)(1.0 - (double)b2), eps, step_size, sqrt_bc2,
                       grad_scale, weight_decay);
    return check_launch("adam_step");
}

// =============================================================================================
// Training the click model (reference pretrain_env.py:25-139): the pieces the forward-only evaluation path did not need.
// =============================================================================================
// embedding backward: dtable[idx[i], :] += g_row(i), g laid out like the output of pcvae_gather_rows.  Rows that occur several
// times in a batch are summed in arrival order (the reference's index_add on the GPU is unordered too) by a compare-and-swap loop
// (common.h: atomic_add_f32 - the plain fp32 atomicAdd loses updates between XCDs).
__global__ void __launch_bounds__(256) scatter_add_rows_kernel(const float* __restrict__ g, int64_t g_ld, int group, int D,
                                                               const int64_t* __restrict__ idx, int64_t n_idx,
                                                               float* __restrict__ dtable) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    for (int64_t i = wave; i < n_idx; i += n_waves) {
        const float* src = g + (i / group) * g_ld + (i % group) * (int64_t)D;
        float* dst = dtable + idx[i] * (int64_t)D;
        for (int d = lane; d < D; d += 64) atomic_add_f32(dst + d, src[d]);
    }
}
extern "C" int pcvae_scatter_add_rows(const float* g, int64_t g_ld, int group, int D, const int64_t* idx, int64_t n_idx,
                                      float* dtable, int64_t n_rows, pcvae_stream_t stream) {
    if (n_idx == 0) return PCVAE_OK;
    PCVAE_REQUIRE(g && idx && dtable && D > 0 && group > 0 && n_rows > 0 && g_ld >= (int64_t)group * D,
                  "scatter_add_rows: bad arguments");
    const int64_t blocks = std::min<int64_t>(cdiv(n_idx, 4), 256 * 8);
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), g, g_ld, group, D,
                       idx, n_idx, dtable);
    return check_launch("scatter_add_rows");
}

// F.normalize with the row norms kept for the backward pass, and its backward: dx = (g - y <y, g>) / max(||x||, 1e-12)
__global__ void __launch_bounds__(256) normalize_rows_norm_kernel(float* __restrict__ x, int64_t ldx, int64_t rows, int cols,
                                                                  float* __restrict__ norm) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float ss = 0.f;
    for (int c = lane; c < cols; c += 64) { const float v = x[r * ldx + c]; ss += v * v; }
    ss = wave_sum(ss);
    const float n = fmaxf(sqrtf(ss), 1e-12f);
    if (lane == 0) norm[r] = n;
    const float inv = 1.f / n;
    for (int c = lane; c < cols; c += 64) x[r * ldx + c] *= inv;
}
__global__ void __launch_bounds__(256) normalize_rows_bwd_kernel(const float* __restrict__ y, int64_t ldy,
                                                                 const float* __restrict__ norm, const float* __restrict__ g,
                                                                 int64_t ldg, float* __restrict__ dx, int64_t lddx,
                                                                 int64_t rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float dot = 0.f;
    for (int c = lane; c < cols; c += 64) dot = fmaf(y[r * ldy + c], g[r * ldg + c], dot);
    dot = wave_sum(dot);
    const float inv = 1.f / norm[r];
    for (int c = lane; c < cols; c += 64) dx[r * lddx + c] = (g[r * ldg + c] - y[r * ldy + c] * dot) * inv;
}
extern "C" int pcvae_normalize_rows_norm(float* x, int64_t ldx, int64_t rows, int cols, float* norm, pcvae_stream_t stream) {
    PCVAE_REQUIRE(cols > 0 && ldx >= cols && rows >= 0 && ((x && norm) || rows == 0), "normalize_rows_norm: bad arguments");
    if (rows == 0) return PCVAE_OK;
    hipLaunchKernelGGL(normalize_rows_norm_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, as_stream(stream), x, ldx,
                       rows, cols, norm);
    return check_launch("normalize_rows_norm");
}
extern "C" int pcvae_normalize_rows_bwd(const float* y, int64_t ldy, const float* norm, const float* g, int64_t ldg,
                                        float* dx, int64_t lddx, int64_t rows, int cols, pcvae_stream_t stream) {
    PCVAE_REQUIRE(cols > 0 && ldy >= cols && ldg >= cols && lddx >= cols && rows >= 0 && ((y && norm && g && dx) || rows == 0),
                  "normalize_rows_bwd: bad arguments");
    if (rows == 0) return PCVAE_OK;
    hipLaunchKernelGGL(normalize_rows_bwd_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, as_stream(stream), y, ldy,
                       norm, g, ldg, dx, lddx, rows, cols);
    return check_launch("normalize_rows_bwd");
}

// nn.BCELoss()(sigmoid(x), t) per element with its gradient (pretrain_env.py:57-58,84): logs clamped at -100 as torch does,
// dL/dx = (s - t) / max(s (1 - s), 1e-12) * s (1 - s) * grad_scale
__global__ void bce_sigmoid_kernel(const float* __restrict__ x, const float* __restrict__ t, int64_t n,
                                   float* __restrict__ loss, float* __restrict__ dx, float grad_scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float s = 1.f / (1.f + expf(-x[i]));
        const float l1 = fmaxf(logf(s), -100.f), l0 = fmaxf(logf(1.f - s), -100.f);
        loss[i] = -(t[i] * l1 + (1.f - t[i]) * l0);
        if (dx) {
            const float q = s * (1.f - s);
            dx[i] = (s - t[i]) / fmaxf(q, 1e-12f) * q * grad_scale;
        }
    }
}
extern "C" int pcvae_bce_sigmoid(const float* x, const float* t, int64_t n, float* loss, float* dx, float grad_scale,
                                 pcvae_stream_t stream) {
    if (n == 0) return PCVAE_OK;
    PCVAE_REQUIRE(x && t && loss && n > 0, "bce_sigmoid: bad arguments");
    const int64_t blocks = std::min<int64_t>(cdiv(n, 256), 2048);
    hipLaunchKernelGGL(bce_sigmoid_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, t, n, loss, dx,
                       grad_scale);
    return check_launch("bce_sigmoid");
}

// F.relu backward keyed on the activated output: g *= (y > 0)
__global__ void relu_bwd_kernel(float* __restrict__ g, int64_t ldg, const float* __restrict__ y, int64_t ldy, int64_t rows,
                                int cols) {
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i % cols);
        if (!(y[r * ldy + c] > 0.f)) g[r * ldg + c] = 0.f;
    }
}
extern "C" int pcvae_relu_bwd(float* g, int64_t ldg, const float* y, int64_t ldy, int64_t rows, int cols,
                              pcvae_stream_t stream) {
    PCVAE_REQUIRE(g && y && cols > 0 && ldg >= cols && ldy >= cols, "relu_bwd: bad arguments");
    if (rows == 0) return PCVAE_OK;
    const int64_t blocks = std::min<int64_t>(cdiv(rows * cols, 256), 2048);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), g, ldg, y, ldy, rows, cols);
    return check_launch("relu_bwd");
}

// =============================================================================================
// Offline metrics of generated slates (reference analysis.py:5-30)
// =============================================================================================
// item coverage: mark every generated id in an N-bit map (atomicOr), then count the set bits
__global__ void coverage_mark_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t N, unsigned int* __restrict__ bits) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = ids[i];
        if (v >= 0 && v < N) atomicOr(bits + (v >> 5), 1u << (v & 31));
    }
}
__global__ void __launch_bounds__(1024) coverage_count_kernel(const unsigned int* __restrict__ bits, int64_t words,
                                                              int64_t* __restrict__ count) {
    __shared__ unsigned long long part[16];
    unsigned long long c = 0;
    for (int64_t i = threadIdx.x; i < words; i += 1024) c += __popc(bits[i]);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < 16; ++w) t += part[w];
        *count = (int64_t)t;
    }
}
extern "C" int pcvae_coverage_count(const int64_t* ids, int64_t n, int64_t N, unsigned int* bits, int64_t* count,
                                    pcvae_stream_t stream) {
    PCVAE_REQUIRE(N > 0 && n >= 0 && bits && count && (ids || n == 0), "coverage_count: bad arguments");
    const int64_t words = cdiv(N, 32);
    if (int rc = pcvae_zero(bits, (size_t)words * 4, stream)) return rc;   // the fill kernel, never a memset node (pcvae_zero)
    if (n > 0) {
        const int64_t blocks = std::min<int64_t>(cdiv(n, 256), 2048);
        hipLaunchKernelGGL(coverage_mark_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), ids, n, N, bits);
    }
    hipLaunchKernelGGL(coverage_count_kernel, dim3(1), dim3(1024), 0, as_stream(stream), bits, words, count);
    return check_launch("coverage_count");
}

// intra-list similarity: with e_i the L2-normalised embedding of slot i, sum_{i,j} <e_i, e_j> = ||sum_i e_i||^2, so
// ILS = (||sum_i e_i||^2 - S) / (S (S - 1)); one wave per slate, no [S, S] matrix
__global__ void __launch_bounds__(256) ils_kernel(const float* __restrict__ E, int D, const int64_t* __restrict__ slates,
                                                  int64_t B, int S, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float tot = 0.f;   // ||sum_i e_i||^2 accumulated over the d slices this lane owns
    for (int d0 = 0; d0 < D; d0 += 64) {
        const int d = d0 + lane;
        float acc = 0.f;
        for (int i = 0; i < S; ++i) {
            const float* row = E + slates[b * S + i] * (int64_t)D;
            float ss = 0.f;
            for (int k = lane; k < D; k += 64) ss = fmaf(row[k], row[k], ss);
            ss = wave_sum(ss);
            if (d < D) acc += row[d] / fmaxf(sqrtf(ss), 1e-12f);
        }
        tot = fmaf(acc, acc, tot);
    }
    tot = wave_sum(tot);
    if (lane == 0) out[b] = (tot - (float)S) / (float)(S * (S - 1));
}
extern "C" int pcvae_ils(const float* E, int64_t N, int D, const int64_t* slates, int64_t B, int S, float* out,
                         pcvae_stream_t stream) {
    PCVAE_REQUIRE(E && N > 0 && D > 0 && S > 1 && B >= 0 && ((slates && out) || B == 0), "ils: bad arguments");
    if (B == 0) return PCVAE_OK;
    hipLaunchKernelGGL(ils_kernel, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, as_stream(stream), E, D, slates, B, S, out);
    return check_launch("ils");
}
