// Bare bf16 MFMA loops on random register operands: 32x32x16 vs 16x16x32 at equal FLOPs per wave
// (does the chip hold a higher clock on one shape?  MI355X_MICROARCH.md 'DVFS give-back' item 7)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ void __launch_bounds__(512, 1) probe(const float* __restrict__ in, float* __restrict__ out, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) {
            a[i][j] = (__bf16)in[(threadIdx.x * 64 + i * 8 + j) & 65535];
            b[i][j] = (__bf16)in[(threadIdx.x * 64 + 32 + i * 8 + j + blockIdx.x) & 65535];
        }
    float r = 0.f;
    if (SHAPE == 32) {
        f32x16 c[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[i], c[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) r += c[i][j];
    } else {
        f32x4 c[8];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) c[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i + 1) & 3], c[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) r += c[i][j];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
    float *in, *out;
    std::vector<float> h(65536);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMalloc(&in, 65536 * 4); hipMalloc(&out, 256 * 8 * 512 * 4);
    hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200000;
    for (int rep = 0; rep < 3; ++rep)
        for (int shape : {32, 16}) {
            hipEventRecord(e0);
            if (shape == 32) hipLaunchKernelGGL(probe<32>, dim3(256), dim3(512), 0, 0, in, out, iters);
            else hipLaunchKernelGGL(probe<16>, dim3(256), dim3(512), 0, 0, in, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // flops: 32-shape: 4 mfma x 32768 per iter per wave; 16-shape: 8 x 16384 per iter per wave
            double fl = (double)iters * 4 * 32768.0 * 8 * 256;
            printf("shape %dx: %.2f ms  %.0f TFLOP/s\n", shape, ms, fl / (ms * 1e-3) / 1e12);
        }
    return 0;
}
