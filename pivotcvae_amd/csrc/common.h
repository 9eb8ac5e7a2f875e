// Shared device/host helpers for libpcvae_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <type_traits>
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

// ---- one table row for a lane group of D / 8 lanes (the gather kernels: sparse K5, K9) ------------------------------------------
// fp32 table: lane j holds columns [4 j, 4 j + 4) and [D/2 + 4 j, D/2 + 4 j + 4) - two 16-byte loads, each a contiguous half row over
// the group.  bf16 table (the stated arithmetic of configs 3 / 5; rows of 2 D bytes): lane j holds columns [8 j, 8 j + 8) - ONE
// 16-byte load; the values are widened exactly (a bf16 is the top half of an fp32), products and sums stay fp32.
#ifndef PCVAE_GATHER_UNR_F32
#define PCVAE_GATHER_UNR_F32 4
#endif
#ifndef PCVAE_GATHER_UNR_BF16
#define PCVAE_GATHER_UNR_BF16 4   // (4, 6, 8, 12 measured: the same time - these kernels sit at the fabric's byte rate, not at a latency bound)
#endif
template <int D, bool BF16>
struct GatherRow {
    struct RawF32 { float4 a, b; };
    struct RawBF16 { uint4 w; };
    // what a lane keeps in flight per row: 32 bytes of fp32 or 16 bytes of bf16 - so twice as many bf16 rows for the same registers
    using Raw = typename std::conditional<BF16, RawBF16, RawF32>::type;
    static constexpr int UNR = BF16 ? PCVAE_GATHER_UNR_BF16 : PCVAE_GATHER_UNR_F32;   // lane-group steps in flight per lane
    static __device__ __forceinline__ int col_a(int j) { return BF16 ? 8 * j : 4 * j; }
    static __device__ __forceinline__ int col_b(int j) { return BF16 ? 8 * j + 4 : D / 2 + 4 * j; }
    static __device__ __forceinline__ Raw zero() {
        Raw r;
        if constexpr (BF16) r.w = make_uint4(0u, 0u, 0u, 0u);
        else { r.a = make_float4(0.f, 0.f, 0.f, 0.f); r.b = r.a; }
        return r;
    }
    static __device__ __forceinline__ Raw load_raw(const void* E, int64_t n, int j) {
        Raw r;
        if constexpr (BF16) r.w = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(E) + n * D + 8 * j);
        else {
            const float* e = reinterpret_cast<const float*>(E) + n * D;
            r.a = *reinterpret_cast<const float4*>(e + 4 * j);
            r.b = *reinterpret_cast<const float4*>(e + D / 2 + 4 * j);
        }
        return r;
    }
    static __device__ __forceinline__ void widen(const Raw& r, float4& a, float4& b) {
        if constexpr (BF16) {
            a = make_float4(__uint_as_float(r.w.x << 16), __uint_as_float(r.w.x & 0xffff0000u), __uint_as_float(r.w.y << 16),
                            __uint_as_float(r.w.y & 0xffff0000u));
            b = make_float4(__uint_as_float(r.w.z << 16), __uint_as_float(r.w.z & 0xffff0000u), __uint_as_float(r.w.w << 16),
                            __uint_as_float(r.w.w & 0xffff0000u));
        } else { a = r.a; b = r.b; }
    }
    static __device__ __forceinline__ void load(const void* E, int64_t n, int j, float4& a, float4& b) {
        widen(load_raw(E, n, j), a, b);
    }
};

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
