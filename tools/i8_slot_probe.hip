// i8_slot_probe.hip - would an int8 slicing of the fp32 operands beat bf16x6 on the D = 128 cross-entropy slot?   
//
// The headline kernel (catalog_ce_x3_pipe_kernel<128, 2, 3>) computes fp32-exact products on the bf16 matrix cores: every fp32 operand
// as 3 bf16 components, 6 v_mfma_f32_16x16x32_bf16 per 32-deep product step - 198 MFMAs per slot (one wave, 32 rows x 32 items, both
// contractions + row sums), 87-90 % MFMA-pipe busy.  I8 MFMAs run at twice the bf16 rate with exact i32 accumulation.  A row-scaled slicing of an
// operand into 4 x 7-bit integers (28 bits >= 24) needs the 10 slice pairs (i, j) with i + j <= 3: per 64-deep step 10
// v_mfma_i32_16x16x64_i8 against 12 bf16 MFMAs of the same duration - at most 198 -> 165 MFMAs per slot = 1.2x, IF nothing else grows.
// What grows (per slot and wave, conservatively counted):
//   * the softmax numerators (32 per lane and slot) must be sliced in the kernel: scale, v_cvt_i32_f32, then per slice a shift, a
//     subtract and a byte pack: ~14 vector ops per value against ~5 for the three bf16 components (cvt_pk, sub, cvt_pk, sub, cvt_pk);
//   * the i32 accumulators of the 4 slice-pair shifts (i + j = 0 .. 3) must be converted and combined with their scales:
//     2 ops x 4 shifts x 4 registers per 16 x 16 logits tile, 8 tiles per slot: 256 ops (bf16x6 accumulates in fp32: none);
//   * operands are no longer the fp32 values themselves (block scaling per row: an element 2^-20 of its row's maximum keeps 8 bits).
// This probe issues both instruction mixes as bare loops (one wave per SIMD, every CU busy, operands in registers, LDS reads from a
// conflict-free dummy image, results garbage; no dependencies between fillers and MFMAs) and reports wall time and shader cycles per
// slot: X0 / I0 = the MFMAs alone (the 1.2x), X / I = with each scheme's vector work spread evenly between the MFMAs.
// The part is power-limited under MFMA-dense load (profiles/r04_slot_issue_probe.txt): wall time, not cycles, is the verdict.
//
//     hipcc --offload-arch=gfx950 -O3 -o i8_slot_probe_bin tools/i8_slot_probe.hip && ./i8_slot_probe_bin > profiles/r06_i8_slot_probe.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// filler F of a slot: NEXP exponentials first in the rotation, NLDS 16-byte LDS reads, the rest plain 4-cycle vector ops
// (fma / shift / and / cvt_i32 in turn), all spread evenly: filler f sits behind MFMA floor(f * NMFMA / NFILL)
template <int F, int NFILL, int NEXP, int NLDS>
__device__ __forceinline__ void filler(float (&x)[8], unsigned (&w)[8], bf16x8& frag, const unsigned lds) {
    // Bresenham-style interleave of the three kinds
    constexpr bool is_exp = (F * NEXP) / NFILL != ((F + 1) * NEXP) / NFILL;
    constexpr bool is_lds = !is_exp && (F * NLDS) / NFILL != ((F + 1) * NLDS) / NFILL;
    if constexpr (is_exp)
        asm volatile("v_exp_f32 %0, %1" : "=v"(x[F % 8]) : "v"(x[(F + 3) % 8]));
    else if constexpr (is_lds)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(frag) : "v"(lds), "n"((F % 16) * 1024));
    else if constexpr (F % 4 == 0)
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[F % 8]) : "v"(x[(F + 1) % 8]), "v"(x[(F + 2) % 8]));
    else if constexpr (F % 4 == 1)
        asm volatile("v_lshrrev_b32 %0, 7, %1" : "=v"(w[F % 8]) : "v"(w[(F + 3) % 8]));
    else if constexpr (F % 4 == 2)
        asm volatile("v_and_b32 %0, 0x7f, %1" : "=v"(w[F % 8]) : "v"(w[(F + 5) % 8]));
    else
        asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(w[F % 8]) : "v"(x[(F + 1) % 8]));
}
template <int F0, int F1, int NFILL, int NEXP, int NLDS>
__device__ __forceinline__ void fillers(float (&x)[8], unsigned (&w)[8], bf16x8& frag, const unsigned lds) {
    if constexpr (F0 < F1) {
        filler<F0, NFILL, NEXP, NLDS>(x, w, frag, lds);
        fillers<F0 + 1, F1, NFILL, NEXP, NLDS>(x, w, frag, lds);
    }
}

template <int M, int NM, int NFILL, int NEXP, int NLDS>
__device__ __forceinline__ void slot_bf16(f32x4 (&acc)[12], const bf16x8 (&a)[4], const bf16x8 (&b)[4], float (&x)[8], unsigned (&w)[8],
                                          bf16x8& frag, const unsigned lds) {
    if constexpr (M < NM) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[M % 12]) : "v"(a[M % 4]), "v"(b[(M / 4) % 4]));
        if constexpr (NFILL > 0) fillers<(M * NFILL) / NM, ((M + 1) * NFILL) / NM, NFILL, NEXP, NLDS>(x, w, frag, lds);
        slot_bf16<M + 1, NM, NFILL, NEXP, NLDS>(acc, a, b, x, w, frag, lds);
    }
}
template <int M, int NM, int NFILL, int NEXP, int NLDS>
__device__ __forceinline__ void slot_i8(i32x4 (&acc)[12], const i32x4 (&a)[4], const i32x4 (&b)[4], float (&x)[8], unsigned (&w)[8],
                                        bf16x8& frag, const unsigned lds) {
    if constexpr (M < NM) {
        asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[M % 12]) : "v"(a[M % 4]), "v"(b[(M / 4) % 4]));
        if constexpr (NFILL > 0) fillers<(M * NFILL) / NM, ((M + 1) * NFILL) / NM, NFILL, NEXP, NLDS>(x, w, frag, lds);
        slot_i8<M + 1, NM, NFILL, NEXP, NLDS>(acc, a, b, x, w, frag, lds);
    }
}

// MIX 0: bf16x6 (NM bf16 MFMAs), MIX 1: int8 slices (NM i8 MFMAs)
template <int MIX, int NM, int NFILL, int NEXP, int NLDS>
__global__ void __launch_bounds__(256, 1) probe(const float* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc,
                                                const int slots) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // 96 KB: one workgroup per CU, one wave per SIMD
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32768 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = in[i];
    __syncthreads();
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = in[(lane + i * 64) & 65535] * 0.01f;
    unsigned w[8];
    for (int i = 0; i < 8; ++i) w[i] = (unsigned)(lane * 2654435761u + i);
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)(lane * 16);
    bf16x8 frag;
    for (int j = 0; j < 8; ++j) frag[j] = (__bf16)0.f;
    float r = 0.f;
    long long t0, t1;
    if constexpr (MIX == 0) {
        bf16x8 a[4], b[4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 8; ++j) {
                a[i][j] = (__bf16)in[(threadIdx.x * 64 + i * 8 + j) & 65535];
                b[i][j] = (__bf16)in[(threadIdx.x * 64 + 32 + i * 8 + j + blockIdx.x) & 65535];
            }
        f32x4 acc[12];
        for (int i = 0; i < 12; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
        for (int s = 0; s < slots; ++s) {
            slot_bf16<0, NM, NFILL, NEXP, NLDS>(acc, a, b, x, w, frag, lds);
            if constexpr (NLDS > 0) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        }
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        for (int i = 0; i < 12; ++i) for (int j = 0; j < 4; ++j) r += acc[i][j];
    } else {
        i32x4 a[4], b[4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {   // random bytes in [-64, 63]: 7-bit slices
                unsigned va = 0, vb = 0;
                for (int k = 0; k < 4; ++k) {
                    va |= (unsigned)(((int)(in[(threadIdx.x * 97 + i * 16 + j * 4 + k) & 65535] * 63.f)) & 0xff) << (8 * k);
                    vb |= (unsigned)(((int)(in[(threadIdx.x * 89 + 7 + i * 16 + j * 4 + k + blockIdx.x) & 65535] * 63.f)) & 0xff) << (8 * k);
                }
                a[i][j] = (int)va;
                b[i][j] = (int)vb;
            }
        i32x4 acc[12];
        for (int i = 0; i < 12; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
        for (int s = 0; s < slots; ++s) {
            slot_i8<0, NM, NFILL, NEXP, NLDS>(acc, a, b, x, w, frag, lds);
            if constexpr (NLDS > 0) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        }
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        for (int i = 0; i < 12; ++i) for (int j = 0; j < 4; ++j) r += (float)acc[i][j];
    }
    for (int i = 0; i < 8; ++i) r += x[i] + (float)w[i];
    r += (float)frag[0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static double g_ref_ns = 0.0;

template <int MIX, int NM, int NFILL, int NEXP, int NLDS>
static double run(const char* name, const float* in, float* out, long long* cyc, int slots) {
    const int grid = 256;
    CHECK(hipFuncSetAttribute((const void*)probe<MIX, NM, NFILL, NEXP, NLDS>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    probe<MIX, NM, NFILL, NEXP, NLDS><<<grid, 256, 96 * 1024>>>(in, out, cyc, slots);   // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    probe<MIX, NM, NFILL, NEXP, NLDS><<<grid, 256, 96 * 1024>>>(in, out, cyc, slots);
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
    // the slot's ALGORITHMIC work: 32 rows x 32 items x 128 dims, two contractions = 4 * 32 * 32 * 128 flop per wave-slot, 1024 waves
    // (198 MFMAs x 16384 flop = 6 x that + the row-sum MFMAs)
    const double tf = 4.0 * 32 * 32 * 128 * 1024 / (ns * 1e-9) / 1e12;
    if (g_ref_ns == 0.0) g_ref_ns = ns;
    printf("%-58s %8.1f ns per slot = %5.0f TF algorithmic = %.3f of 2.5 PF  (%7.1f s_memtime ticks per slot)  x%.3f vs first line\n",
           name, ns, tf, tf / 2500.0, ticks, g_ref_ns / ns);
    return ns;
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
    const int slots = 4000;
    printf("# tools/i8_slot_probe.hip on one MI355X: 256 workgroups x 4 waves (one wave per SIMD), %d slots per wave; a slot = the D = 128 cross-entropy\n"
           "# subtile of one wave (32 rows x 32 items x 128 dims, logits + gradient contraction) in fp32-exact split arithmetic\n", slots);
    // vector work per slot: bf16x6 - 32 exp, ~5 ops x 32 numerators + ~40 of bookkeeping = 200 plain ops, 36 LDS reads (three component images);
    //                       int8   - 32 exp, ~14 ops x 32 numerators + 256 accumulator conversions + ~40 = 744 plain ops, 48 LDS reads (four slice images)
    const double x0 = run<0, 198, 0, 0, 0>("X0: bf16x6, 198 x v_mfma_f32_16x16x32_bf16 alone", in, out, cyc, slots);
    const double i0 = run<1, 165, 0, 0, 0>("I0: int8 4x7-bit, 165 x v_mfma_i32_16x16x64_i8 alone", in, out, cyc, slots);
    const double xf = run<0, 198, 268, 32, 36>("X : bf16x6 + 32 exp, 200 vector ops, 36 ds_read_b128", in, out, cyc, slots);
    const double i_f = run<1, 165, 824, 32, 48>("I : int8 + 32 exp, 744 vector ops, 48 ds_read_b128", in, out, cyc, slots);
    const double i_h = run<1, 165, 452, 32, 48>("I/2: int8 with HALF of that vector work (372 ops)", in, out, cyc, slots);
    run<0, 198, 268, 32, 36>("X  again", in, out, cyc, slots);
    run<1, 165, 824, 32, 48>("I  again", in, out, cyc, slots);
    printf("# MFMAs alone: int8 is x%.3f of bf16x6 (the instruction-count bound is 198 / 165 = x1.200)\n", x0 / i0);
    printf("# with each scheme's vector work: int8 is x%.3f of bf16x6; with only HALF of the int8 scheme's vector work x%.3f\n", xf / i_f, xf / i_h);
    printf("# decision rule (VERDICT r5 item 9): build the kernel only at >= x1.15 with the error table of the f32 kernel\n");
    return 0;
}
