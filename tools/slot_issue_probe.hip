// slot_issue_probe.hip - what would v_mfma_f32_32x32x16_bf16 buy the D = 64 cross-entropy slot?
//
// The D = 64 bf16 kernel (catalog_ce_bf16_pipe_kernel<64, 4>) is bound by vector ISSUE, not by the matrix pipe: per 32-item subtile
// and 64-row wave it issues 36 v_mfma_f32_16x16x32_bf16 (an MFMA holds the issue port for 8 of its 16 cycles), 32 v_exp_f32 (8 each),
// 16 v_cvt_pk_bf16_f32, 4 ds_read_b128 and 8 ds_read_b64_tr_b16 - ~690 issue cycles against 576 matrix-pipe cycles
// (profiles/r04_bf16_d64_timing_probes.txt).  The same contraction in 32x32x16 tiles is 16 MFMAs of 32 cycles (the same 512 pipe
// cycles) + 4 row-sum MFMAs (128 cycles instead of 64), each holding the port for 8 of its 32 cycles: 160 issue cycles instead of 288.
//
// This probe issues both instruction mixes as bare loops (one wave per SIMD, every CU busy, operands in registers, the LDS reads
// from a conflict-free dummy image, results garbage) and reports shader cycles per slot:
//     mix A  36 x 16x16x32 + fillers        mix B  20 x 32x32x16 + the same fillers        A0 / B0  the MFMAs alone
// The fillers sit one or two per MFMA gap, evenly spread, as in the kernel's schedule.  No dependencies between fillers and MFMAs
// (registers disjoint): what is measured is issue, which is what bounds the real slot.
//
//     hipcc --offload-arch=gfx950 -O3 -o slot_issue_probe tools/slot_issue_probe.hip && ./slot_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// filler number F of a slot (0 .. 59): 32 exponentials, 16 conversions, 4 A-fragment reads, 8 transposed reads, interleaved
template <int F>
__device__ __forceinline__ void filler(float (&x)[8], unsigned (&w)[4], bf16x8& frag, s16x4& tr, const unsigned lds) {
    constexpr int k = F % 15;   // 15-periodic pattern: 8 exp, 4 cvt, 1 b128, 2 tr  (x 4 = 32 / 16 / 4 / 8)
    if constexpr (k == 0 || k == 2 || k == 4 || k == 6 || k == 8 || k == 10 || k == 12 || k == 14)
        asm volatile("v_exp_f32 %0, %1" : "=v"(x[(F / 2) % 8]) : "v"(x[(F / 2 + 3) % 8]));
    else if constexpr (k == 1 || k == 5 || k == 9 || k == 13)
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[F % 4]) : "v"(x[F % 8]), "v"(x[(F + 1) % 8]));
    else if constexpr (k == 3)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(frag) : "v"(lds), "n"((F % 4) * 1024));
    else
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(tr) : "v"(lds), "n"((F % 8) * 512));
}
template <int F0, int F1>
__device__ __forceinline__ void fillers(float (&x)[8], unsigned (&w)[4], bf16x8& frag, s16x4& tr, const unsigned lds) {
    if constexpr (F0 < F1) {
        filler<F0>(x, w, frag, tr, lds);
        fillers<F0 + 1, F1>(x, w, frag, tr, lds);
    }
}

// mix A: 36 MFMAs 16x16x32, filler f of 60 behind MFMA floor(f * 36 / 60)
template <int M, bool FILL>
__device__ __forceinline__ void slot_a(f32x4 (&acc)[12], const bf16x8 (&a)[4], const bf16x8 (&b)[4], float (&x)[8], unsigned (&w)[4],
                                       bf16x8& frag, s16x4& tr, const unsigned lds) {
    if constexpr (M < 36) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[M % 12]) : "v"(a[M % 4]), "v"(b[(M / 4) % 4]));
        if constexpr (FILL) fillers<(M * 60 + 35) / 36, ((M + 1) * 60 + 35) / 36>(x, w, frag, tr, lds);
        slot_a<M + 1, FILL>(acc, a, b, x, w, frag, tr, lds);
    }
}
// mix B: 20 MFMAs 32x32x16, filler f of 60 behind MFMA floor(f * 20 / 60): three per gap
template <int M, bool FILL>
__device__ __forceinline__ void slot_b(f32x16 (&acc)[5], const bf16x8 (&a)[4], const bf16x8 (&b)[4], float (&x)[8], unsigned (&w)[4],
                                       bf16x8& frag, s16x4& tr, const unsigned lds) {
    if constexpr (M < 20) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[M % 5]) : "v"(a[M % 4]), "v"(b[(M / 4) % 4]));
        if constexpr (FILL) fillers<M * 3, M * 3 + 3>(x, w, frag, tr, lds);
        slot_b<M + 1, FILL>(acc, a, b, x, w, frag, tr, lds);
    }
}

template <int MIX, bool FILL>
__global__ void __launch_bounds__(256, 1) probe(const float* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc,
                                                const int slots) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // 96 KB: one workgroup per CU, one wave per SIMD
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = in[i];
    __syncthreads();
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) {
            a[i][j] = (__bf16)in[(threadIdx.x * 64 + i * 8 + j) & 65535];
            b[i][j] = (__bf16)in[(threadIdx.x * 64 + 32 + i * 8 + j + blockIdx.x) & 65535];
        }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = in[(lane + i * 64) & 65535] * 0.01f;
    unsigned w[4] = {0, 0, 0, 0};
    bf16x8 frag = a[0];
    s16x4 tr = {0, 0, 0, 0};
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)(lane * 16);
    float r = 0.f;
    long long t0, t1;
    if constexpr (MIX == 0) {
        f32x4 acc[12];
        for (int i = 0; i < 12; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
        for (int s = 0; s < slots; ++s) {
            slot_a<0, FILL>(acc, a, b, x, w, frag, tr, lds);
            if constexpr (FILL) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        }
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        for (int i = 0; i < 12; ++i) for (int j = 0; j < 4; ++j) r += acc[i][j];
    } else {
        f32x16 acc[5];
        for (int i = 0; i < 5; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
        for (int s = 0; s < slots; ++s) {
            slot_b<0, FILL>(acc, a, b, x, w, frag, tr, lds);
            if constexpr (FILL) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        }
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        for (int i = 0; i < 5; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
    }
    for (int i = 0; i < 8; ++i) r += x[i];
    for (int i = 0; i < 4; ++i) r += (float)w[i];
    r += (float)frag[0] + (float)tr[0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MIX, bool FILL>
static void run(const char* name, const float* in, float* out, long long* cyc, int slots) {
    const int grid = 256;
    CHECK(hipFuncSetAttribute((const void*)probe<MIX, FILL>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    probe<MIX, FILL><<<grid, 256, 96 * 1024>>>(in, out, cyc, slots);   // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    probe<MIX, FILL><<<grid, 256, 96 * 1024>>>(in, out, cyc, slots);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(grid * 4);
    CHECK(hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    double sum = 0;
    for (long long v : h) sum += (double)v;
    const double ticks = sum / h.size() / slots;                  // s_memtime ticks (100 MHz on this part) per slot and wave
    const double ns = (double)ms * 1e6 / slots;                   // wall time per slot (all waves run the same loop side by side)
    // 64 rows x 32 items x 64 dims, two contractions: 4 * 64 * 32 * 64 flop per wave-slot, 1024 waves
    const double tf = 4.0 * 64 * 32 * 64 * 1024 / (ns * 1e-9) / 1e12;
    printf("%-44s %8.1f ns per slot  = %6.0f TF chip-wide = %.3f of 2.5 PF   (s_memtime: %.2f ticks per slot)\n", name, ns, tf, tf / 2500.0, ticks);
}

int main() {
    float *in, *out;
    long long* cyc;
    std::vector<float> h(65536);
    srand(7);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    CHECK(hipMalloc(&in, h.size() * 4));
    CHECK(hipMalloc(&out, 256 * 256 * 4));
    CHECK(hipMalloc(&cyc, 256 * 4 * 8));
    CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int slots = 20000;
    run<0, false>("A0: 36 x 16x16x32, MFMAs alone", in, out, cyc, slots);
    run<1, false>("B0: 20 x 32x32x16, MFMAs alone", in, out, cyc, slots);
    run<0, true>("A : 36 x 16x16x32 + 32 exp 16 cvt 12 lds", in, out, cyc, slots);
    run<1, true>("B : 20 x 32x32x16 + 32 exp 16 cvt 12 lds", in, out, cyc, slots);
    run<0, true>("A  again", in, out, cyc, slots);
    run<1, true>("B  again", in, out, cyc, slots);
    return 0;
}
