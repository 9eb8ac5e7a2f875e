// GPU clock under light load: a chain of dependent v_fma_f32 (4 cycles each at one wave per SIMD) timed with HIP events, for 1 .. 1024
// workgroups, launched back to back (20 launches per measurement, as a captured step would).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) chain(float* out, int n) {
    float x = threadIdx.x * 1e-9f, a = 1.0000001f, b = 1e-9f;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x = __builtin_fmaf(x, a, b);
    }
    if (x == 123.456f) out[0] = x;
}
__global__ void __launch_bounds__(256) mfma_chain(float* out, int n) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float a = threadIdx.x * 1e-9f, b = 1e-9f;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        }
    }
    if (c0[0] + c1[0] + c2[0] + c3[0] == 123.456f) out[0] = c0[0];
}
int main() {
    float* d;
    hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which)
        for (int wgs : {1, 64, 256, 1024}) {
            for (int n : {64, 1024}) {
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    for (int l = 0; l < 20; ++l) {
                        if (which == 0) hipLaunchKernelGGL(chain, dim3(wgs), dim3(256), 0, 0, d, n);
                        else hipLaunchKernelGGL(mfma_chain, dim3(wgs), dim3(256), 0, 0, d, n);
                    }
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    const double us = ms * 1e3 / 20;
                    // chain: 16 n dependent fmas of 4 cycles; mfma: 16 n MFMAs of 32 cycles (8 passes), 4 accumulators
                    const double cyc = which == 0 ? 16.0 * n * 4 : 16.0 * n * 32;
                    if (rep) printf("%s wgs %4d n %5d: %8.2f us per launch -> %.2f GHz if issue-bound (launch overhead included)\n",
                                    which ? "mfma16x16x4" : "fma chain  ", wgs, n, us, cyc / us * 1e-3);
                }
            }
        }
    return 0;
}
