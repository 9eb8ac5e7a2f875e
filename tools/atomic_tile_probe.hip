// atomic_tile_probe: the weight-gradient epilogue alone - nz batch splits x (ny x nx) tiles of 64 x 64, every workgroup adds 1.0
// to each element of its tile with the MFMA C/D lane map; expected nz everywhere.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <int MODE>
__global__ void __launch_bounds__(256) add_tile(float* C, long ldc, int nx, int ny, int nz, long M, long N, int spin, int use_lds) {
    extern __shared__ char smem[];
    const int l = blockIdx.x;
    if (l >= nx * ny * nz) return;
    const int bx = l % nx, by = (l / nx) % ny;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1, li = lane & 31, h = lane >> 5;
    if (use_lds) { reinterpret_cast<float*>(smem)[threadIdx.x] = (float)l; __syncthreads(); }
    for (volatile int s = 0; s < spin; ++s) {}
    const long n = (long)bx * 64 + wn * 32 + li;
    if (n >= N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long m = (long)by * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < M) {
            float* a = &C[m * ldc + n];
            if (MODE == 0) atomicAdd(a, 1.0f);                                                                    // HIP default
            else if (MODE == 1) __hip_atomic_fetch_add(a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // sc1
            else if (MODE == 2) atomicAdd(reinterpret_cast<int*>(a), 1);                                          // integer, agent
            else if (MODE == 3) __hip_atomic_fetch_add(reinterpret_cast<int*>(a), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else {   // compare-and-swap loop on the bits
                unsigned* u = reinterpret_cast<unsigned*>(a);
                unsigned old = __hip_atomic_load(u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), assumed;
                do { assumed = old; old = atomicCAS(u, assumed, __float_as_uint(__uint_as_float(assumed) + 1.0f)); } while (old != assumed);
            }
        }
    }
}
int main(int argc, char** argv) {
    const long M = argc > 1 ? atol(argv[1]) : 256, N = argc > 2 ? atol(argv[2]) : 1408;
    const int nz = argc > 3 ? atoi(argv[3]) : 4, spin = argc > 4 ? atoi(argv[4]) : 300, lds = argc > 5 ? atoi(argv[5]) : 32768;
    const int nx = (N + 63) / 64, ny = (M + 63) / 64;
    float* d; CK(hipMalloc(&d, M * N * 4));
    std::vector<float> h(M * N);
    const char* names[] = {"float atomicAdd (HIP default)", "float add, system scope", "int atomicAdd (agent)", "int add, system scope", "float CAS loop"};
    for (int mode = 0; mode < 5; ++mode) {
        int bad_iters = 0; long bad = 0;
        for (int it = 0; it < 20; ++it) {
            CK(hipMemset(d, 0, M * N * 4)); CK(hipDeviceSynchronize());
            const dim3 g(nx * ny * nz), b(256);
            if (mode == 0) hipLaunchKernelGGL(add_tile<0>, g, b, lds, 0, d, N, nx, ny, nz, M, N, spin, lds > 0);
            if (mode == 1) hipLaunchKernelGGL(add_tile<1>, g, b, lds, 0, d, N, nx, ny, nz, M, N, spin, lds > 0);
            if (mode == 2) hipLaunchKernelGGL(add_tile<2>, g, b, lds, 0, d, N, nx, ny, nz, M, N, spin, lds > 0);
            if (mode == 3) hipLaunchKernelGGL(add_tile<3>, g, b, lds, 0, d, N, nx, ny, nz, M, N, spin, lds > 0);
            if (mode == 4) hipLaunchKernelGGL(add_tile<4>, g, b, lds, 0, d, N, nx, ny, nz, M, N, spin, lds > 0);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), d, M * N * 4, hipMemcpyDeviceToHost));
            long bb = 0;
            for (auto v : h) {
                const float want = (float)nz;
                int vi; __builtin_memcpy(&vi, &v, 4);
                bb += (mode == 2 || mode == 3) ? (vi != nz) : (v != want);
            }
            bad += bb; bad_iters += bb != 0;
        }
        printf("%-32s M=%ld N=%ld nz=%d spin=%d: %d of 20 iterations wrong, %ld wrong elements\n", names[mode], M, N, nz, spin, bad_iters, bad);
    }
    return 0;
}
