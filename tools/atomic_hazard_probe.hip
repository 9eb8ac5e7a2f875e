// atomic_hazard_probe: does a float atomicAdd (agent scope) from workgroups on different XCDs see the zeros a PREVIOUS kernel (or
// hipMemset) wrote to the same buffer?  Kernel add<<<nblk>>> adds 1 to every element of a 64 x 64 tile from each of nblk blocks
// (block b runs on XCD b & 7); expected nblk everywhere.
//   hipcc --offload-arch=gfx950 -O3 -o build/atomic_hazard_probe tools/atomic_hazard_probe.hip && build/atomic_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void zero_plain(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = 0.f; }
__global__ void zero_sc1(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) __hip_atomic_store(p + i, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void add_one(float* p, int n, int spin) {
    for (volatile int s = 0; s < spin; ++s) {}
    for (int i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(p + i, 1.0f);
}
__global__ void add_one_sys(float* p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) __hip_atomic_fetch_add(p + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
int main() {
    const int n = 4096, nblk = 4, iters = 200;
    float* d; CK(hipMalloc(&d, n * 4));
    std::vector<float> h(n);
    const char* names[] = {"hipMemset + sync", "zero kernel (plain stores)", "zero kernel (sc1 stores)", "hipMemset + sync + 5 ms sleep",
                           "zero kernel (plain) + system-scope atomics", "zero kernel (plain), adds delayed by a spin"};
    for (int mode = 0; mode < 6; ++mode) {
        int bad_iters = 0; long bad_elems = 0;
        for (int it = 0; it < iters; ++it) {
            for (int i = 0; i < n; ++i) h[i] = 7.f;
            CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));   // stale content: 7
            if (mode == 0 || mode == 3) { CK(hipMemset(d, 0, n * 4)); CK(hipDeviceSynchronize()); }
            else if (mode == 2) hipLaunchKernelGGL(zero_sc1, dim3(n / 256), dim3(256), 0, 0, d, n);
            else hipLaunchKernelGGL(zero_plain, dim3(n / 256), dim3(256), 0, 0, d, n);
            if (mode == 3) { struct timespec ts{0, 5000000L}; nanosleep(&ts, nullptr); }
            if (mode == 4) hipLaunchKernelGGL(add_one_sys, dim3(nblk), dim3(256), 0, 0, d, n);
            else hipLaunchKernelGGL(add_one, dim3(nblk), dim3(256), 0, 0, d, n, mode == 5 ? 200000 : 0);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
            long b = 0;
            for (int i = 0; i < n; ++i) b += h[i] != (float)nblk;
            bad_elems += b; bad_iters += b != 0;
        }
        printf("%-48s: %d of %d iterations wrong, %ld wrong elements in total\n", names[mode], bad_iters, iters, bad_elems);
    }
    for (int spin : {50, 200, 500, 1000, 2000, 4000, 8000, 16000, 50000}) {   // atomics a few microseconds into the kernel
        int bad_iters = 0; long bad_elems = 0;
        for (int it = 0; it < iters; ++it) {
            for (int i = 0; i < n; ++i) h[i] = 7.f;
            CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
            CK(hipMemset(d, 0, n * 4)); CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(add_one, dim3(nblk), dim3(256), 0, 0, d, n, spin);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
            long b = 0;
            for (int i = 0; i < n; ++i) b += h[i] != (float)nblk;
            bad_elems += b; bad_iters += b != 0;
        }
        printf("hipMemset + sync, adds after a spin of %6d      : %d of %d iterations wrong, %ld wrong elements\n", spin, bad_iters, iters, bad_elems);
    }
    return 0;
}
