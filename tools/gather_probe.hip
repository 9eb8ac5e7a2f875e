// gather_probe: A/B of row-gather kernel shapes on the config-4 gather (98 304 rows of 512 B from a 1M-row table),
// cold caches (a 512 MB fill between launches), HIP events around each launch.  Build + run:
//   hipcc --offload-arch=gfx950 -O3 -o build/gather_probe tools/gather_probe.hip && build/gather_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- V0: the product kernel's shape: LPR lanes per row, UNROLL row groups in flight, one or more sweeps per wave
template <int UNROLL, bool NT_LD, bool NT_ST>
__global__ void __launch_bounds__(256) k_batch(const f32x4* __restrict__ table, int chunks, int lpr,
                                               const int64_t* __restrict__ idx, int64_t n_idx, f32x4* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int rows_per_wave = 64 / lpr;
    const int sub = lane / lpr, chunk0 = lane % lpr;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t n_groups = (n_idx + rows_per_wave - 1) / rows_per_wave;
    for (int64_t g0 = wave * UNROLL; g0 < n_groups; g0 += n_waves * UNROLL) {
        int64_t src[UNROLL], dst[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t i = (g0 + u) * rows_per_wave + sub;
            const bool ok = g0 + u < n_groups && i < n_idx;
            src[u] = ok ? idx[i] : -1;
            dst[u] = i * chunks;
        }
        for (int c = chunk0; c < chunks; c += lpr) {
            f32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                if (src[u] >= 0) v[u] = NT_LD ? __builtin_nontemporal_load(&table[src[u] * chunks + c]) : table[src[u] * chunks + c];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                if (src[u] >= 0) { if (NT_ST) __builtin_nontemporal_store(v[u], &out[dst[u] + c]); else out[dst[u] + c] = v[u]; }
        }
    }
}

// ---- V1: software-pipelined: a wave owns a CONTIGUOUS run of row groups; batch b+1's rows are requested before
// batch b's stores are issued, so reads and writes overlap inside a wave and the wave makes several sweeps.
// Row = 32 lanes x 16 B (D = 128 only: chunks == 32, lpr == 32, 2 rows per wave-instruction)
template <int UNROLL, bool NT_LD, bool NT_ST>
__global__ void __launch_bounds__(256) k_pipe(const f32x4* __restrict__ table, const int64_t* __restrict__ idx,
                                              int64_t n_idx, f32x4* __restrict__ out, int groups_per_wave) {
    const int lane = threadIdx.x & 63;
    const int sub = lane >> 5, c = lane & 31;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_groups = (n_idx + 1) / 2;
    const int64_t gb = wave * groups_per_wave;
    const int64_t ge = gb + groups_per_wave < n_groups ? gb + groups_per_wave : n_groups;
    if (gb >= ge) return;
    f32x4 v[UNROLL];
    int64_t srcn[UNROLL];
    // prologue: indices of batch 0, rows of batch 0, indices of batch 1
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const int64_t i = (gb + u) * 2 + sub;
        srcn[u] = (gb + u < ge && i < n_idx) ? idx[i] : -1;
    }
    for (int64_t g0 = gb; g0 < ge; g0 += UNROLL) {
        int64_t src[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) src[u] = srcn[u];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            if (src[u] >= 0) v[u] = NT_LD ? __builtin_nontemporal_load(&table[src[u] * 32 + c]) : table[src[u] * 32 + c];
        // next batch's indices ride under this batch's row loads
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t i = (g0 + UNROLL + u) * 2 + sub;
            srcn[u] = (g0 + UNROLL + u < ge && i < n_idx) ? idx[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            if (src[u] >= 0) {
                const int64_t i = (g0 + u) * 2 + sub;
                if (NT_ST) __builtin_nontemporal_store(v[u], &out[i * 32 + c]); else out[i * 32 + c] = v[u];
            }
    }
}

// ---- V2: two register banks: rows of batch b+1 are in flight while batch b is stored
template <int UNROLL>
__global__ void __launch_bounds__(256) k_pipe2(const f32x4* __restrict__ table, const int64_t* __restrict__ idx,
                                               int64_t n_idx, f32x4* __restrict__ out, int groups_per_wave) {
    const int lane = threadIdx.x & 63;
    const int sub = lane >> 5, c = lane & 31;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_groups = (n_idx + 1) / 2;
    const int64_t gb = wave * groups_per_wave;
    const int64_t ge = gb + groups_per_wave < n_groups ? gb + groups_per_wave : n_groups;
    if (gb >= ge) return;
    f32x4 va[UNROLL], vb[UNROLL];
    int64_t sa[UNROLL], sb[UNROLL];
#define LOAD_IDX(S, G0)                                                       \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                      \
        const int64_t i = ((G0) + u) * 2 + sub;                               \
        S[u] = ((G0) + u < ge && i < n_idx) ? idx[i] : -1;                    \
    }
#define LOAD_ROWS(V, S)                                                       \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u)                        \
        if (S[u] >= 0) V[u] = __builtin_nontemporal_load(&table[S[u] * 32 + c]);
#define STORE_ROWS(V, S, G0)                                                  \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u)                        \
        if (S[u] >= 0) __builtin_nontemporal_store(V[u], &out[(((G0) + u) * 2 + sub) * 32 + c]);
    LOAD_IDX(sa, gb)
    LOAD_IDX(sb, gb + UNROLL)
    LOAD_ROWS(va, sa)
    for (int64_t g0 = gb; g0 < ge; g0 += 2 * UNROLL) {
        LOAD_ROWS(vb, sb)
        STORE_ROWS(va, sa, g0)
        LOAD_IDX(sa, g0 + 2 * UNROLL)
        LOAD_ROWS(va, sa)
        STORE_ROWS(vb, sb, g0 + UNROLL)
        LOAD_IDX(sb, g0 + 3 * UNROLL)
    }
}

// ---- V3 (round 4): the wave's 32 row indices by ONE coalesced 256-byte load (lane l < 32 loads idx[first + l]), distributed by
// cross-lane reads; then the product kernel's 16 row loads + 16 stores.  D = 128 only (32 lanes per row, 2 rows per instruction).
template <bool NT_LD, bool NT_ST>
__global__ void __launch_bounds__(256) k_coal(const f32x4* __restrict__ table, const int64_t* __restrict__ idx, int64_t n_idx,
                                              f32x4* __restrict__ out) {
    const int lane = threadIdx.x & 63, sub = lane >> 5, c = lane & 31;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t first = wave * 32;
    if (first >= n_idx) return;
    const int64_t mine = first + (lane & 31) < n_idx ? idx[first + (lane & 31)] : -1;
    int lo = (int)(mine & 0xffffffff), hi = (int)(mine >> 32);
    f32x4 v[16];
    int64_t src[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int from = 2 * u + sub;
        const int l2 = __shfl(lo, from, 64), h2 = __shfl(hi, from, 64);
        src[u] = ((int64_t)h2 << 32) | (uint32_t)l2;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
        if (src[u] >= 0) v[u] = NT_LD ? __builtin_nontemporal_load(&table[src[u] * 32 + c]) : table[src[u] * 32 + c];
#pragma unroll
    for (int u = 0; u < 16; ++u)
        if (src[u] >= 0) {
            f32x4* o = &out[(first + 2 * u + sub) * 32 + c];
            if (NT_ST) __builtin_nontemporal_store(v[u], o); else *o = v[u];
        }
}

// ---- V4 (round 4): as V3 with the row loads and stores of the two halves of the batch interleaved (8 loads, 8 loads, 8 stores, 8 stores)
template <int U2>
__global__ void __launch_bounds__(256) k_coal2(const f32x4* __restrict__ table, const int64_t* __restrict__ idx, int64_t n_idx,
                                               f32x4* __restrict__ out) {
    const int lane = threadIdx.x & 63, sub = lane >> 5, c = lane & 31;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t first = wave * (2 * U2);
    if (first >= n_idx) return;
    const int64_t mine = (lane & 31) < 2 * U2 && first + (lane & 31) < n_idx ? idx[first + (lane & 31)] : -1;
    int lo = (int)(mine & 0xffffffff), hi = (int)(mine >> 32);
    f32x4 v[U2];
    int64_t src[U2];
#pragma unroll
    for (int u = 0; u < U2; ++u) {
        const int from = 2 * u + sub;
        src[u] = ((int64_t)__shfl(hi, from, 64) << 32) | (uint32_t)__shfl(lo, from, 64);
    }
#pragma unroll
    for (int u = 0; u < U2; ++u)
        if (src[u] >= 0) v[u] = __builtin_nontemporal_load(&table[src[u] * 32 + c]);
#pragma unroll
    for (int u = 0; u < U2; ++u)
        if (src[u] >= 0) __builtin_nontemporal_store(v[u], &out[(first + 2 * u + sub) * 32 + c]);
}

// ---- V5 (round 4): V3 with the index load non-temporal too, any block size; V6: 64 rows per wave (one 64-lane index load), the
// second half's row loads issued before the first half's stores
template <bool NT_IDX>
__global__ void k_coal_b(const f32x4* __restrict__ table, const int64_t* __restrict__ idx, int64_t n_idx, f32x4* __restrict__ out) {
    const int lane = threadIdx.x & 63, sub = lane >> 5, c = lane & 31;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t first = wave * 32;
    if (first >= n_idx) return;
    int64_t mine = -1;
    if (first + (lane & 31) < n_idx) mine = NT_IDX ? __builtin_nontemporal_load(&idx[first + (lane & 31)]) : idx[first + (lane & 31)];
    int lo = (int)(mine & 0xffffffff), hi = (int)(mine >> 32);
    f32x4 v[16];
    int64_t src[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) src[u] = ((int64_t)__shfl(hi, 2 * u + sub, 64) << 32) | (uint32_t)__shfl(lo, 2 * u + sub, 64);
#pragma unroll
    for (int u = 0; u < 16; ++u)
        if (src[u] >= 0) v[u] = __builtin_nontemporal_load(&table[src[u] * 32 + c]);
#pragma unroll
    for (int u = 0; u < 16; ++u)
        if (src[u] >= 0) __builtin_nontemporal_store(v[u], &out[(first + 2 * u + sub) * 32 + c]);
}
__global__ void __launch_bounds__(256) k_coal64(const f32x4* __restrict__ table, const int64_t* __restrict__ idx, int64_t n_idx,
                                                f32x4* __restrict__ out) {
    const int lane = threadIdx.x & 63, sub = lane >> 5, c = lane & 31;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t first = wave * 64;
    if (first >= n_idx) return;
    const int64_t mine = first + lane < n_idx ? idx[first + lane] : -1;
    int lo = (int)(mine & 0xffffffff), hi = (int)(mine >> 32);
    f32x4 v[32];
    int64_t src[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) src[u] = ((int64_t)__shfl(hi, 2 * u + sub, 64) << 32) | (uint32_t)__shfl(lo, 2 * u + sub, 64);
#pragma unroll
    for (int u = 0; u < 32; ++u)
        if (src[u] >= 0) v[u] = __builtin_nontemporal_load(&table[src[u] * 32 + c]);
#pragma unroll
    for (int u = 0; u < 32; ++u)
        if (src[u] >= 0) __builtin_nontemporal_store(v[u], &out[(first + 2 * u + sub) * 32 + c]);
}

__global__ void k_empty() {}

int main(int argc, char** argv) {
    const int64_t N = 1000000, D = 128, B = 8192, S = 10;
    const int64_t mult = argc > 1 ? atoi(argv[1]) : 1;   // rows = mult x the config-4 gather
    const int64_t n_idx = B * (S + 2) * mult;
    const int chunks = D / 4;
    std::vector<float> h_table((size_t)N * D);
    for (size_t i = 0; i < h_table.size(); ++i) h_table[i] = (float)(i % 1000003) * 1e-3f;
    std::vector<int64_t> h_idx(n_idx);
    uint64_t s = 88172645463325252ull;
    for (auto& v : h_idx) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (int64_t)(s % (uint64_t)N); }
    float *table, *out, *flush;
    int64_t* idx;
    CK(hipMalloc(&table, h_table.size() * 4));
    CK(hipMalloc(&out, (size_t)n_idx * D * 4));
    CK(hipMalloc(&flush, 512ull << 20));
    CK(hipMalloc(&idx, n_idx * 8));
    CK(hipMemcpy(table, h_table.data(), h_table.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(idx, h_idx.data(), n_idx * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double nbytes = (double)n_idx * (2 * D * 4 + 8);
    std::vector<float> h_out((size_t)n_idx * D);

    auto run = [&](const char* name, auto launch) {
        double tot = 0, best = 1e30;
        const int n = 12;
        for (int it = 0; it < n + 3; ++it) {
            CK(hipMemsetAsync(flush, it, 512ull << 20, 0));
            CK(hipEventRecord(e0, 0));
            launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 3) { tot += ms; best = std::min(best, (double)ms); }
        }
        CK(hipMemcpy(h_out.data(), out, h_out.size() * 4, hipMemcpyDeviceToHost));
        bool ok = true;
        for (int64_t i = 0; i < n_idx && ok; i += 97)
            for (int d = 0; d < D; d += 31) ok = ok && h_out[i * D + d] == h_table[h_idx[i] * D + d];
        CK(hipMemsetAsync(out, 0, (size_t)n_idx * D * 4, 0));
        printf("%-44s avg %7.2f us  min %7.2f us  %6.0f GB/s avg  %6.0f GB/s best  frac %.3f  %s\n", name, tot / n * 1e3,
               best * 1e3, nbytes / (tot / n * 1e-3) / 1e9, nbytes / (best * 1e-3) / 1e9, nbytes / (tot / n * 1e-3) / 8e12,
               ok ? "ok" : "WRONG");
        fflush(stdout);
    };

    const int64_t n_groups = n_idx / 2;
    run("empty kernel (event-to-event floor)", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); });
    const f32x4* T = reinterpret_cast<const f32x4*>(table);
    f32x4* O = reinterpret_cast<f32x4*>(out);
#define BATCH(U, NL, NS, BLOCKS)                                                                                   \
    run("batch U=" #U " ntld=" #NL " ntst=" #NS " blocks=" #BLOCKS, [&] {                                          \
        int blocks = BLOCKS > 0 ? BLOCKS : (int)((n_groups / U + 3) / 4);                                          \
        hipLaunchKernelGGL((k_batch<U, NL, NS>), dim3(blocks), dim3(256), 0, 0, T, chunks, 32, idx, n_idx, O);     \
    })
    BATCH(16, true, true, 0);
    BATCH(16, false, true, 0);
    BATCH(16, true, false, 0);
    BATCH(16, false, false, 0);
    BATCH(8, true, true, 0);
    BATCH(4, true, true, 0);
    BATCH(8, true, true, 768);
    BATCH(4, true, true, 1024);
    BATCH(4, true, true, 2048);
    BATCH(2, true, true, 2048);
#define PIPE(U, NL, NS, WAVES_PER_CU)                                                                              \
    run("pipe U=" #U " ntld=" #NL " ntst=" #NS " waves/CU=" #WAVES_PER_CU, [&] {                                   \
        const int64_t waves = 256 * WAVES_PER_CU;                                                                  \
        int gpw = (int)((n_groups + waves - 1) / waves);                                                           \
        gpw = (gpw + U - 1) / U * U;                                                                               \
        const int64_t used = (n_groups + gpw - 1) / gpw;                                                           \
        hipLaunchKernelGGL((k_pipe<U, NL, NS>), dim3((unsigned)((used + 3) / 4)), dim3(256), 0, 0, T, idx, n_idx, O, gpw); \
    })
    PIPE(4, true, true, 8);
    PIPE(4, true, true, 16);
    PIPE(4, true, true, 32);
    PIPE(8, true, true, 8);
    PIPE(8, true, true, 16);
    PIPE(2, true, true, 32);
    PIPE(4, false, true, 16);
#define PIPE2(U, WAVES_PER_CU)                                                                                     \
    run("pipe2 U=" #U " waves/CU=" #WAVES_PER_CU, [&] {                                                            \
        const int64_t waves = 256 * WAVES_PER_CU;                                                                  \
        int gpw = (int)((n_groups + waves - 1) / waves);                                                           \
        gpw = (gpw + 2 * U - 1) / (2 * U) * (2 * U);                                                               \
        const int64_t used = (n_groups + gpw - 1) / gpw;                                                           \
        hipLaunchKernelGGL((k_pipe2<U>), dim3((unsigned)((used + 3) / 4)), dim3(256), 0, 0, T, idx, n_idx, O, gpw); \
    })
    PIPE2(4, 8);
    PIPE2(4, 16);
    PIPE2(8, 8);
    PIPE2(2, 16);
    PIPE2(2, 32);
    // reference points: plain device copy of the same byte count, and a read-only / write-only split
    run("coal (one idx load per wave) nt nt", [&] { hipLaunchKernelGGL((k_coal<true, true>), dim3((unsigned)((n_idx / 32 + 3) / 4)), dim3(256), 0, 0, T, idx, n_idx, O); });
    run("coal (one idx load per wave) plain ld, nt st", [&] { hipLaunchKernelGGL((k_coal<false, true>), dim3((unsigned)((n_idx / 32 + 3) / 4)), dim3(256), 0, 0, T, idx, n_idx, O); });
    run("coal2 U2=8 (16 rows per wave)", [&] { hipLaunchKernelGGL((k_coal2<8>), dim3((unsigned)((n_idx / 16 + 3) / 4)), dim3(256), 0, 0, T, idx, n_idx, O); });
    run("coal2 U2=4 (8 rows per wave)", [&] { hipLaunchKernelGGL((k_coal2<4>), dim3((unsigned)((n_idx / 8 + 3) / 4)), dim3(256), 0, 0, T, idx, n_idx, O); });
    for (int rep = 0; rep < 2; ++rep) {
    run("coal (one idx load per wave) nt nt, 256 thr", [&] { hipLaunchKernelGGL((k_coal<true, true>), dim3((unsigned)((n_idx / 32 + 3) / 4)), dim3(256), 0, 0, T, idx, n_idx, O); });
    run("coal_b plain idx, 128 thr", [&] { hipLaunchKernelGGL((k_coal_b<false>), dim3((unsigned)((n_idx / 32 + 1) / 2)), dim3(128), 0, 0, T, idx, n_idx, O); });
    run("coal_b plain idx, 64 thr", [&] { hipLaunchKernelGGL((k_coal_b<false>), dim3((unsigned)(n_idx / 32)), dim3(64), 0, 0, T, idx, n_idx, O); });
    run("coal_b plain idx, 512 thr", [&] { hipLaunchKernelGGL((k_coal_b<false>), dim3((unsigned)((n_idx / 32 + 7) / 8)), dim3(512), 0, 0, T, idx, n_idx, O); });
    run("coal_b plain idx, 1024 thr", [&] { hipLaunchKernelGGL((k_coal_b<false>), dim3((unsigned)((n_idx / 32 + 15) / 16)), dim3(1024), 0, 0, T, idx, n_idx, O); });
    run("coal_b nt idx, 256 thr", [&] { hipLaunchKernelGGL((k_coal_b<true>), dim3((unsigned)((n_idx / 32 + 3) / 4)), dim3(256), 0, 0, T, idx, n_idx, O); });
    run("coal64 (64 rows per wave), 256 thr", [&] { hipLaunchKernelGGL(k_coal64, dim3((unsigned)((n_idx / 64 + 3) / 4)), dim3(256), 0, 0, T, idx, n_idx, O); });
    run("coal64 (64 rows per wave), 128 thr", [&] { hipLaunchKernelGGL(k_coal64, dim3((unsigned)((n_idx / 64 + 1) / 2)), dim3(128), 0, 0, T, idx, n_idx, O); });
    }
    BATCH(16, true, true, 0);
    run("hipMemcpyDtoD 50 MB (same bytes r+w)", [&] { CK(hipMemcpyAsync(out, table, (size_t)n_idx * D * 4, hipMemcpyDeviceToDevice, 0)); });
    return 0;
}
