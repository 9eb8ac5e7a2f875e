// gather_window_probe: what would a REGION-BUCKETED candidate kernel gain?  Every wave of the chip gathers random 512-byte rows of a
// 1M-row fp32 table and dot-products them (the candidate kernel's inner loop), either uniformly over the whole table (what
// candidate_ce_kernel does: every request crosses the fabric, 7.5 TB/s) or inside a window of W rows that all waves sweep over the
// table IN STEP (region k = rows [k W, (k + 1) W): a bucketed kernel would process each slate row's candidates region by region,
// so that an XCD's 4 MB L2 holds the region every one of its waves is working in).  Same number of rows gathered in every mode.
//   hipcc --offload-arch=gfx950 -O3 -o build/gather_window_probe tools/gather_window_probe.hip && build/gather_window_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {   // cheap counter hash (the probe needs spread, not quality)
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// a wave = one "slate row": `per_region` gathers in each of n_regions regions of W rows; 16 lanes per row, 4 rows per step, 4 steps
// in flight (the product kernel's shape at D = 128)
__global__ void __launch_bounds__(256) k_window(const float* __restrict__ E, int64_t N, int W, int n_regions, int per_region,
                                                float* __restrict__ out) {
    const int lane = threadIdx.x & 63, j = lane & 15, grp = lane >> 4;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    float acc = 0.f;
    for (int k = 0; k < n_regions; ++k) {
        const int64_t base = (int64_t)k * W;
        for (int i0 = 0; i0 < per_region; i0 += 16) {
            float4 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 4 * u + grp;
                const uint32_t h = mix((uint32_t)wave * 2654435761u + (uint32_t)(k * 4099 + i));
                int64_t n = base + (int64_t)(h % (uint32_t)W);
                if (n >= N) n = N - 1;
                const float* e = E + n * 128;
                a[u] = *reinterpret_cast<const float4*>(e + 4 * j);
                b[u] = *reinterpret_cast<const float4*>(e + 64 + 4 * j);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                acc += a[u].x + a[u].y + a[u].z + a[u].w + b[u].x + b[u].y + b[u].z + b[u].w;
        }
    }
    if (acc == 12345.678f) out[wave] = acc;   // never true: keeps the loads alive
}

// the same with a PERSISTENT grid (as many workgroups as are resident at once, all starting together): every wave makes `sweeps`
// passes over the regions, so the chip's waves stay in step without any synchronisation as long as they run equally fast
__global__ void __launch_bounds__(256) k_window_persistent(const float* __restrict__ E, int64_t N, int W, int n_regions, int per_region,
                                                           int sweeps, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, j = lane & 15, grp = lane >> 4;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    float acc = 0.f;
    for (int sw = 0; sw < sweeps; ++sw)
        for (int k = 0; k < n_regions; ++k) {
            const int64_t base = (int64_t)k * W;
            for (int i0 = 0; i0 < per_region; i0 += 16) {
                float4 a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + 4 * u + grp;
                    const uint32_t h = mix((uint32_t)wave * 2654435761u + (uint32_t)((sw * n_regions + k) * 4099 + i));
                    int64_t n = base + (int64_t)(h % (uint32_t)W);
                    if (n >= N) n = N - 1;
                    const float* e = E + n * 128;
                    a[u] = *reinterpret_cast<const float4*>(e + 4 * j);
                    b[u] = *reinterpret_cast<const float4*>(e + 64 + 4 * j);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc += a[u].x + a[u].y + a[u].z + a[u].w + b[u].x + b[u].y + b[u].z + b[u].w;
            }
        }
    if (acc == 12345.678f) out[wave] = acc;
}

int main() {
    const int64_t N = 1000000, R = 81920;
    const int total = 1000;   // rows gathered per wave, as at Cn = 1000
    float *E, *out;
    CK(hipMalloc(&E, N * 128 * sizeof(float)));
    CK(hipMalloc(&out, R * sizeof(float)));
    CK(hipMemset(E, 0, N * 128 * sizeof(float)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%-38s %10s %12s\n", "mode", "ms", "TB/s requested");
    struct M { const char* name; int W, regions, per; } modes[] = {
        {"uniform over the table (1 region)", 1000000, 1, 1008},
        {"regions of 250000 rows (128 MB), 4", 250000, 4, 256},
        {"regions of 62500 rows (32 MB), 16", 62500, 16, 64},
        {"regions of 15625 rows (8 MB), 64", 15625, 64, 16},
        {"regions of 7813 rows (4 MB), 128", 7813, 128, 16},      // (per-region count rounds up to one step: 2x the rows - see TB/s)
        {"regions of 3907 rows (2 MB), 256", 3907, 256, 16},
        {"regions of 1954 rows (1 MB), 512", 1954, 512, 16},
    };
    for (auto& m : modes) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_window, dim3((unsigned)(R / 4)), dim3(256), 0, 0, E, N, m.W, m.regions, m.per, out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        const double rows = (double)R * m.regions * ((m.per + 15) / 16 * 16);
        printf("%-38s %10.3f %12.2f   (%.0f rows per wave; %d would be the job)\n", m.name, best, rows * 512 / (best * 1e-3) / 1e12,
               rows / R, total);
    }
    printf("\npersistent grid (1280 workgroups = 5 per CU, all resident), 16000 rows per wave = the same 81.92 M rows:\n");
    struct P { const char* name; int W, regions, per, sweeps; } pm[] = {
        {"uniform over the table", 1000000, 1, 16000, 1},
        {"regions of 62500 rows (32 MB) x 16", 62500, 16, 16, 62},
        {"regions of 15625 rows (8 MB) x 64", 15625, 64, 16, 16},
        {"regions of 7813 rows (4 MB) x 128", 7813, 128, 16, 8},
        {"regions of 3907 rows (2 MB) x 256", 3907, 256, 16, 4},
        {"regions of 1954 rows (1 MB) x 512", 1954, 512, 16, 2},
        {"regions of 3907 rows (2 MB), 64 per", 3907, 256, 64, 1},
        {"regions of 1954 rows (1 MB), 32 per", 1954, 512, 32, 1},
    };
    for (auto& m : pm) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_window_persistent, dim3(1280), dim3(256), 0, 0, E, N, m.W, m.regions, m.per, m.sweeps, out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        const double rows = 5120.0 * m.regions * m.per * m.sweeps;
        printf("%-38s %10.3f %12.2f   (%.2f M rows)\n", m.name, best, rows * 512 / (best * 1e-3) / 1e12, rows / 1e6);
    }
    return 0;
}
