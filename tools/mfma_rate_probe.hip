// v_mfma_f32_16x16x4_f32 issue rate with (0) constant tiny operands, (1) random operands in the same two registers, (2) random operands
// in 16 distinct B registers + 4 distinct A registers (the stack kernel's group).  One workgroup of 4 waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) k(const float* in, float* out, int n, int n2) {
    __shared__ __attribute__((aligned(16))) float sm[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) sm[i] = in[i];
    __syncthreads();
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float a[4], b[16];
    for (int i = 0; i < 4; ++i) a[i] = MODE == 0 ? 1e-9f : in[threadIdx.x * 4 + i];
    for (int i = 0; i < 16; ++i) b[i] = MODE == 0 ? 1e-9f : in[1024 + threadIdx.x * 16 + i];
    for (int i = 0; i < n; ++i) {
        float am[4];
        for (int j = 0; j < 4; ++j) am[j] = (MODE >= 3 && i >= n2) ? 0.f : a[j];   // MODE 3: a select per operand per group
        if (MODE == 4) {                                                           // MODE 4: + an LDS read per group
            const f32x4 v = *reinterpret_cast<const f32x4*>(sm + ((threadIdx.x * 4 + 16 * i) & 1023));
            for (int j = 0; j < 4; ++j) am[j] = v[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float aa = MODE >= 2 ? am[j] : a[0];
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aa, MODE >= 2 ? b[4 * j] : b[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aa, MODE >= 2 ? b[4 * j + 1] : b[0], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(aa, MODE >= 2 ? b[4 * j + 2] : b[0], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(aa, MODE >= 2 ? b[4 * j + 3] : b[0], c3, 0, 0, 0);
        }
    }
    out[threadIdx.x] = c0[0] + c1[0] + c2[0] + c3[0];
}
int main() {
    float *din, *dout;
    std::vector<float> h(1024 + 4096);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    hipMalloc(&din, h.size() * 4);
    hipMalloc(&dout, 4096);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 0; mode < 5; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            const int n = 1024;
            hipEventRecord(e0);
            for (int l = 0; l < 20; ++l) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, din, dout, n, n);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, din, dout, n, n);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), 0, 0, din, dout, n, n);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(256), 0, 0, din, dout, n, n);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(256), 0, 0, din, dout, n, n);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("mode %d: %.2f us per launch, %.1f ns = %.1f cycles at 2.4 GHz per MFMA\n", mode, ms * 1e3 / 20,
                            ms * 1e6 / 20 / (16.0 * n), ms * 1e6 / 20 / (16.0 * n) * 2.4);
        }
    return 0;
}
