// Shared device/host helpers for libpcvae_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/pcvae.h"

namespace pcvae {

// ---- host-side error plumbing (thread-local text behind pcvae_last_error()) ------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define PCVAE_REQUIRE(cond, ...)                       \
    do {                                               \
        if (!(cond)) {                                 \
            ::pcvae::set_error(__VA_ARGS__);           \
            return PCVAE_EINVAL;                       \
        }                                              \
    } while (0)

// opt a kernel in to `bytes` of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize) once per (kernel, device), under a lock,
// return code checked: the kernels over 64 KB cannot launch without it, and a process may drive several devices / threads
int lds_optin(const void* kernel, int bytes);

static inline hipStream_t as_stream(pcvae_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---- per-kernel timing (pcvae_kernel_timer): HIP events ATTACHED to a dispatch (hipExtLaunchKernelGGL's start / stop events carry
// the kernel's own begin / end timestamps - what rocprofv3's kernel trace shows), as opposed to a hipEventRecord pair around the
// launch, which also measures its own two marker packets (~2.4 us on this chip: 12 % of the 20 us gather).  Off by default.
bool timer_on();
void timer_events(int tag, hipEvent_t* start, hipEvent_t* stop);   // a fresh pair, remembered under `tag`
#define PCVAE_LAUNCH_TIMED(tag, kernel, grid, block, shmem, stream, ...)                                        \
    do {                                                                                                        \
        if (::pcvae::timer_on()) {                                                                              \
            hipEvent_t e0_, e1_;                                                                                \
            ::pcvae::timer_events(tag, &e0_, &e1_);                                                             \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, e0_, e1_, 0, __VA_ARGS__);               \
        } else {                                                                                                \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                \
        }                                                                                                       \
    } while (0)
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr float kLeakySlope = 0.01f;  // nn.LeakyReLU() default (reference models/cvae.py:43)
constexpr int kWave = 64;

// ---- XCD-aware block remap ---------------------------------------------------------------------
// Blocks b and b+8 share an XCD (round-robin dispatch, MI355X_MICROARCH.md "Workgroup dispatch").
// Map the hardware block id to a logical id such that every XCD walks a CONTIGUOUS range of logical
// ids: neighbours in logical order (which share catalog tiles) then hit the same private L2.
// Bijective for any grid size; affects speed only, never results.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7;
    const int q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// ---- Philox4x32-10 (counter-based RNG; Salmon et al. 2011) -----------------------------------
struct Philox4 {
    uint32_t x, y, z, w;
};
__host__ __device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}
__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                           uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = mulhi32(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = mulhi32(M1, c2), lo1 = M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return Philox4{c0, c1, c2, c3};
}

// x % n for n >= 1 without a 64-bit division: magic = floor((2^64 - 1) / n) from the host; the quotient estimate is short of
// floor(x / n) by at most 2.  The uniform item draws (candidate sets, rejection sampler) reduce 64 Philox bits with it.
__device__ __forceinline__ uint64_t mod_magic(uint64_t x, uint64_t n, uint64_t magic) {
    uint64_t r = x - __umul64hi(x, magic) * n;
    while (r >= n) r -= n;
    return r;
}

__device__ __forceinline__ float leaky(float x) { return x > 0.f ? x : kLeakySlope * x; }

// fp32 atomic add that is correct when workgroups on DIFFERENT XCDs add into one cache line.  HIP's atomicAdd(float*) compiles to an
// agent-scope global_atomic_add_f32 without sc1, which the issuing XCD's L2 executes; concurrent adders on other XCDs lose updates
// line by line (tools/atomic_tile_probe.hip: 11 workgroups adding 1.0 to the same 64 x 64 tiles - 17 of 20 launches wrong; integer
// adds and a compare-and-swap loop - 0 of 20).  Hence a CAS loop on the bits; used off the hot path only (the weight gradients
// do not use atomics on their output at all: gemm_f32.hip).
__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
    unsigned* u = reinterpret_cast<unsigned*>(p);
    unsigned old = __hip_atomic_load(u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), assumed;
    do {
        assumed = old;
        old = atomicCAS(u, assumed, __float_as_uint(__uint_as_float(assumed) + v));
    } while (old != assumed);
}

}  // namespace pcvae
