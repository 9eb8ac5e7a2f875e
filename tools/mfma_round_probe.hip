// How does v_mfma_f32_16x16x32_bf16 round?  (hipcc --offload-arch=gfx950 -O2 tools/mfma_round_probe.hip -o /tmp/mfma_round_probe)
// Every row of A and every column of B hold the same 32-vector, so every element of the 16 x 16 result is the same dot product
// sum_k a[k] b[k] + c and the operand layout does not matter.  bf16 values are exact powers of two (or 3 x a power of two), so
// every product is exact and the expected results under each rounding hypothesis are known in closed form.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Case { float a[32], b[32], c; };

__global__ void probe(const Case* cases, int n, float* out) {
    const int lane = threadIdx.x, g = lane >> 4;
    for (int t = 0; t < n; ++t) {
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)cases[t].a[8 * g + i]; b[i] = (__bf16)cases[t].b[8 * g + i]; }
        f32x4 acc = {cases[t].c, cases[t].c, cases[t].c, cases[t].c};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        if (lane == 0) out[t] = acc[0];
    }
}

static Case mk(float c) { Case k; memset(&k, 0, sizeof(k)); k.c = c; return k; }

int main() {
    Case cs[16];
    const char* what[16];
    int n = 0;
    // A: c = 1, one product of 0.75 ulp(1) = 3 * 2^-25
    cs[n] = mk(1.f); cs[n].a[0] = 3.f * ldexpf(1.f, -13); cs[n].b[0] = ldexpf(1.f, -12);
    what[n++] = "1 + 0.75 ulp            RNE: 1+2^-23   RTZ: 1";
    // B: c = 1, one product of -0.25 lower-ulp = -2^-26
    cs[n] = mk(1.f); cs[n].a[0] = -ldexpf(1.f, -13); cs[n].b[0] = ldexpf(1.f, -13);
    what[n++] = "1 - 0.25 ulp_below      RNE: 1         RTZ: 1-2^-24";
    // C: c = 1, 32 products of 2^-26 each (sum = 4 ulp)
    cs[n] = mk(1.f); for (int k = 0; k < 32; ++k) { cs[n].a[k] = ldexpf(1.f, -13); cs[n].b[k] = ldexpf(1.f, -13); }
    what[n++] = "1 + 32 x 2^-26          exact sum first: 1+2^-21   one by one: 1";
    // D: c = 0, one product 1 and 31 products of 2^-28 (31 x 2^-28 = 0.97 ulp(1))
    cs[n] = mk(0.f); for (int k = 0; k < 32; ++k) { cs[n].a[k] = ldexpf(1.f, -14); cs[n].b[k] = ldexpf(1.f, -14); } cs[n].a[5] = 1.f; cs[n].b[5] = 1.f;
    what[n++] = "1 + 31 x 2^-28 (c = 0)   exact sum, RNE: 1+2^-23   small terms cut off: 1";
    // E: the same with c = 1 and the big product 0: the addend c aligned against tiny products
    cs[n] = mk(1.f); for (int k = 0; k < 32; ++k) { cs[n].a[k] = ldexpf(1.f, -14); cs[n].b[k] = ldexpf(1.f, -14); } cs[n].a[5] = 0.f;
    what[n++] = "c = 1 + 31 x 2^-28       exact sum, RNE: 1+2^-23   cut off: 1";
    // F: cancellation inside the dot product: +x, -x pairs of large products and one small one; c = 0
    cs[n] = mk(0.f); for (int k = 0; k < 30; k += 2) { cs[n].a[k] = 1.f; cs[n].b[k] = 1.f; cs[n].a[k + 1] = -1.f; cs[n].b[k + 1] = 1.f; }
    cs[n].a[30] = ldexpf(1.f, -15); cs[n].b[30] = ldexpf(1.f, -15);
    what[n++] = "15 x (+1 -1) + 2^-30     exact: 2^-30 = 9.31e-10   cut off at 2^-24..-27 below the max: 0";
    // G: 0.5 ulp tie: c = 1, product 2^-24 -> RNE ties to even: 1; round-half-up: 1+2^-23
    cs[n] = mk(1.f); cs[n].a[0] = ldexpf(1.f, -12); cs[n].b[0] = ldexpf(1.f, -12);
    what[n++] = "1 + 0.5 ulp (tie)       RNE: 1   half-up: 1+2^-23";
    // H: c = 1 + 2^-23 (odd), product 2^-24: tie -> even = 1 + 2^-22
    cs[n] = mk(1.f + ldexpf(1.f, -23)); cs[n].a[0] = ldexpf(1.f, -12); cs[n].b[0] = ldexpf(1.f, -12);
    what[n++] = "(1+2^-23) + 0.5 ulp     RNE: 1+2^-22   RTZ: 1+2^-23";
    // I: two products of 0.3 ulp each (3*2^-27*... use 5*2^-27 ~ 0.3125 ulp each: sum 0.625 ulp)
    cs[n] = mk(1.f); cs[n].a[0] = 5.f * ldexpf(1.f, -14); cs[n].b[0] = ldexpf(1.f, -13); cs[n].a[9] = 5.f * ldexpf(1.f, -14); cs[n].b[9] = ldexpf(1.f, -13);
    what[n++] = "1 + 2 x 0.3125 ulp      exact sum first, RNE: 1+2^-23   one by one RNE: 1";
    // J: denormal-size product with c = 0
    cs[n] = mk(0.f); cs[n].a[0] = ldexpf(1.f, -70); cs[n].b[0] = ldexpf(1.f, -70);
    what[n++] = "2^-140 (c = 0)           kept: 7.17e-43 (denormal)   flushed: 0";

    Case* d; float* o;
    hipMalloc(&d, sizeof(cs)); hipMalloc(&o, sizeof(float) * 16);
    hipMemcpy(d, cs, sizeof(cs), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, n, o);
    float h[16];
    hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    for (int t = 0; t < n; ++t) {
        uint32_t bits; memcpy(&bits, &h[t], 4);
        printf("%c: %-75s -> %.10g (0x%08x)\n", 'A' + t, what[t], (double)h[t], bits);
    }
    return 0;
}
