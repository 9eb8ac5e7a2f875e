// valu_rate_probe.hip - what the instructions of the split-bf16 operand split cost the vector pipe on gfx950.
//
// The MLP GEMMs (csrc/gemm_f32.hip, bf16x3 / bf16x6) split every fp32 operand into bf16 components in registers, per pair of values and
// level: v_cvt_pk_bf16_f32, v_lshlrev_b32, v_and_b32, v_pk_add_f32.  The counters (profiles/r06_mlp_gemm_pmc_summary.txt) count one issue
// quad-cycle per vector instruction - 48 % of the SIMD-cycles in bf16x6 -, while the loop's throughput saturates at two workgroups per CU
// (profiles/r06_gemm_occupancy_probe.txt): something shared is full.  This probe times bare loops of ONE instruction kind (8 independent
// chains, 64 instructions per trip, 2048 trips; one or two waves per SIMD, every CU busy) with the shader clock and reports cycles per
// instruction and SIMD: an instruction that takes two passes through the 16-lane pipe shows 8 cycles instead of 4.
//
//     hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe_bin tools/valu_rate_probe.hip && ./valu_rate_probe_bin > profiles/r06_valu_rate_probe.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int TRIPS = 2048, PER_TRIP = 64;

template <int KIND>
__device__ __forceinline__ void one(unsigned (&w)[8], f32x2 (&p)[8], f32x4 (&acc)[8], const bf16x8& fa, const bf16x8& fb, const int i) {
    if constexpr (KIND == 0) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(w[i]));
    else if constexpr (KIND == 1) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(w[i]));
    else if constexpr (KIND == 2) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 7]));
    else if constexpr (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
    else if constexpr (KIND == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[i]) : "v"(p[i][0]), "v"(p[i][1]));
    else if constexpr (KIND == 5) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(w[i]) : "v"(p[i][0]), "v"(p[i][1]), "v"(w[(i + 3) & 7]));
    else if constexpr (KIND == 6) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(w[i]) : "v"(w[(i + 1) & 7]));
    else if constexpr (KIND == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
    else if constexpr (KIND == 8) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(fa), "v"(fb));
    else if constexpr (KIND >= 9 && KIND <= 12) {
        // one MFMA in four instructions, the other three plain (KIND 9: v_and_b32, 10: v_cvt_pk_bf16_f32, 11: v_pk_add_f32) or, KIND 12,
        // one in eight with seven v_and_b32: do the vector instructions run in the shadow of the MFMA (time = the MFMAs alone) or not (sum)?
        constexpr int period = KIND == 12 ? 8 : 4;
        if (i % period == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(fa), "v"(fb));
        else if constexpr (KIND == 9 || KIND == 12) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(w[i]));
        else if constexpr (KIND == 10) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[i]) : "v"(p[i][0]), "v"(p[i][1]));
        else asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
    }
}

template <int KIND>
__global__ void __launch_bounds__(512) rate_kernel(long long* cycles, float* sink) {
    unsigned w[8];
    f32x2 p[8];
    f32x4 acc[8];
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) {
        w[i] = 0x3f800000u + threadIdx.x * 8 + i;
        p[i] = f32x2{1.f + threadIdx.x, 2.f + i};
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(1.f + i); fb[i] = (__bf16)(0.5f * i); }
    __syncthreads();
    const long long t0 = clock64();
#pragma unroll 1
    for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
        for (int j = 0; j < PER_TRIP; ++j) one<KIND>(w, p, acc, fa, fb, j & 7);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += __uint_as_float(w[i]) + p[i][0] + p[i][1] + acc[i][0] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int waves_per_simd, long long* d_cycles, float* d_sink) {
    const int threads = 256 * waves_per_simd, nwaves = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((rate_kernel<KIND>), dim3(256), dim3(threads), 0, 0, d_cycles, d_sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((rate_kernel<KIND>), dim3(256), dim3(threads), 0, 0, d_cycles, d_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> c(nwaves);
    CHECK(hipMemcpy(c.data(), d_cycles, nwaves * sizeof(long long), hipMemcpyDeviceToHost));
    double sum = 0;
    for (long long v : c) sum += (double)v;
    // clock64() ticks at 100 MHz on this part (s_memtime): convert through the event time instead - instructions per SIMD / wall time
    const double per_simd = (double)TRIPS * PER_TRIP * waves_per_simd;
    printf("%-22s %d wave%s per SIMD: %8.1f us per launch = %6.3f ns per instruction and SIMD (= %5.2f cycles at 2.4 GHz)   [clock64 ticks per wave: %.0f]\n",
           name, waves_per_simd, waves_per_simd > 1 ? "s" : " ", ms * 1e3, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, sum / nwaves);
}

int main() {
    long long* d_cycles;
    float* d_sink;
    CHECK(hipMalloc(&d_cycles, 256 * 8 * sizeof(long long)));
    CHECK(hipMalloc(&d_sink, 16));
    printf("# bare loops of one instruction kind, %d trips x %d instructions (8 independent chains), 256 workgroups (one per CU)\n", TRIPS, PER_TRIP);
    printf("# mixes: per INSTRUCTION of the mix (a 1 + 3 mix that hides its vector instructions behind the MFMA shows a quarter of the MFMA's time)\n");
    for (int wps = 1; wps <= 2; ++wps) {
        run<0>("v_and_b32", wps, d_cycles, d_sink);
        run<1>("v_lshlrev_b32", wps, d_cycles, d_sink);
        run<2>("v_sub_f32", wps, d_cycles, d_sink);
        run<6>("v_fma_f32", wps, d_cycles, d_sink);
        run<3>("v_pk_add_f32", wps, d_cycles, d_sink);
        run<7>("v_pk_fma_f32", wps, d_cycles, d_sink);
        run<4>("v_cvt_pk_bf16_f32", wps, d_cycles, d_sink);
        run<5>("v_perm_b32", wps, d_cycles, d_sink);
        run<8>("v_mfma_16x16x32_bf16", wps, d_cycles, d_sink);
        run<9>("1 mfma + 3 v_and", wps, d_cycles, d_sink);
        run<10>("1 mfma + 3 v_cvt_pk", wps, d_cycles, d_sink);
        run<11>("1 mfma + 3 v_pk_add", wps, d_cycles, d_sink);
        run<12>("1 mfma + 7 v_and", wps, d_cycles, d_sink);
    }
    return 0;
}
