// K10: the sampled pivot rules (spt / spi / sgt; models/pivotcvae.py:341-352, 362-374, 407-415):
//     pivot ~ Categorical(sigmoid(<x_r, E_n>))   over the WHOLE catalog, one draw per slate.
// The reference scores every item ([B, N] = mm + sigmoid) and hands the matrix to torch.multinomial; rounds 1-4 did the same
// contraction inside a Gumbel-max kernel (2 B N D flops on the f32 matrix pipe + N Philox / log / exp per row: 48 ms per config-4
// step).  None of that work is needed to draw ONE index from these weights: sigmoid is bounded by 1, so
//
//     repeat:  n ~ Uniform{0 .. N-1};  u ~ Uniform(0, 1);  until u < sigmoid(<x_r, E_n>)          (rejection sampling)
//
// returns n with probability (1 / N) sigmoid(s_n) / sum_m (1 / N) sigmoid(s_m) = sigmoid(s_n) / sum_m sigmoid(s_m): EXACTLY the
// reference's categorical, for the price of 1 / mean_n sigmoid(s_n) dot products per slate (2 when the scores straddle zero, as
// they do for L2-normalised item rows) instead of N.  The proposals of a row are a fixed sequence k = 0, 1, 2, ... from Philox
// keyed by (seed, GLOBAL row, k) - independent of launch geometry and of how a batch is sharded over ranks - and the sample is
// the proposal with the LOWEST accepted k: a wave owns a row, a lane group of D/8 lanes scores one proposal (two 16-byte loads per
// lane of the fp32 table row, an exact fmaf dot product), CS_UNR * 64 / (D/8) proposals per round.
// A row whose first CS_KMAX = 512 proposals are all rejected (likely only when the mean sigmoid is below ~1e-2: most items scored
// under -4.5; round 5's cap of 4096 let such a row run 256 serial rounds of random row gathers BEFORE paying the full-catalog pass as
// well - ADVICE r5 - now at most 32 rounds at D = 128) is flagged and
// drawn by the exact Gumbel-max kernel over the whole catalog instead (catalog_argmax_f32_kernel<D, true>, an independent exact
// sampler: the mixture is still the reference's distribution); workgroups of that launch leave at once when none of their rows is
// flagged.  tests/philox_ref.py restates the proposal stream on the host.
#include "catalog_plan.h"

using namespace pcvae;

namespace {

constexpr int CS_UNR = 4;         // lane-group steps in flight per lane (2 x 16-byte loads each)
constexpr int CS_KMAX = 512;      // proposals before a row goes to the Gumbel-max kernel (a multiple of every round size)

struct SampleParams {
    const float* x;     // [R, D]
    const float* E;     // [N, D] fp32
    int64_t R, N;
    uint64_t seed, row_offset, magic;
    const uint64_t* row_offset_dev;   // or null: added to row_offset (a captured graph replays at a new stream position)
    int64_t* idx;       // [R]
    uint8_t* unres;     // [R]: 1 = all CS_KMAX proposals rejected (idx[r] is not written)
};

template <int D>
__global__ void __launch_bounds__(256) catalog_sample_reject_kernel(SampleParams p) {
    constexpr int LPI = D / 8;            // lanes per proposal: a lane holds columns [4 j, 4 j + 4) and [D/2 + 4 j, D/2 + 4 j + 4)
    constexpr int IPS = 64 / LPI;         // proposals per step of a wave
    static_assert(CS_KMAX % (IPS * CS_UNR) == 0, "the cap is a whole number of rounds at every width");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= p.R) return;                 // wave-uniform
    const int j = lane % LPI, grp = lane / LPI;
    const float4 xa = *reinterpret_cast<const float4*>(p.x + r * D + 4 * j);
    const float4 xb = *reinterpret_cast<const float4*>(p.x + r * D + D / 2 + 4 * j);
    const uint64_t grow = p.row_offset + (p.row_offset_dev ? *p.row_offset_dev : 0ull) + (uint64_t)r;

    int64_t found = -1;
    for (int k0 = 0; k0 < CS_KMAX && found < 0; k0 += IPS * CS_UNR) {
        float4 ea[CS_UNR], eb[CS_UNR];
        int64_t n[CS_UNR];
        float uu[CS_UNR];
#pragma unroll
        for (int u = 0; u < CS_UNR; ++u) {
            const int k = k0 + u * IPS + grp;
            const Philox4 ph = philox4x32_10((uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)k, 0x524A4354u /*"RJCT"*/,
                                             (uint32_t)p.seed, (uint32_t)(p.seed >> 32));
            n[u] = (int64_t)mod_magic(((uint64_t)ph.x << 32) | ph.y, (uint64_t)p.N, p.magic);
            uu[u] = ((float)ph.z + 0.5f) * 2.3283064365386963e-10f;   // 2^-32: (0, 1], relative resolution 2^-24 down to 1e-10
            const float* e = p.E + n[u] * D;
            ea[u] = *reinterpret_cast<const float4*>(e + 4 * j);
            eb[u] = *reinterpret_cast<const float4*>(e + D / 2 + 4 * j);
        }
        int kacc = 0x7fffffff;
        int64_t nacc = -1;
#pragma unroll
        for (int u = CS_UNR - 1; u >= 0; --u) {     // downwards: the lowest accepted k of the lane group is kept
            float s = ea[u].x * xa.x;
            s = fmaf(ea[u].y, xa.y, s); s = fmaf(ea[u].z, xa.z, s); s = fmaf(ea[u].w, xa.w, s);
            s = fmaf(eb[u].x, xb.x, s); s = fmaf(eb[u].y, xb.y, s); s = fmaf(eb[u].z, xb.z, s); s = fmaf(eb[u].w, xb.w, s);
#pragma unroll
            for (int o = 1; o < LPI; o <<= 1) s += __shfl_xor(s, o, 64);   // every lane of the group holds the score
            const float sig = 1.0f / (1.0f + __expf(-s));
            if (uu[u] < sig) { kacc = k0 + u * IPS + grp; nacc = n[u]; }
        }
        // the wave's lowest accepted k and its item
        int kmin = kacc;
#pragma unroll
        for (int o = LPI; o < 64; o <<= 1) kmin = min(kmin, __shfl_xor(kmin, o, 64));
        if (kmin != 0x7fffffff) {                    // wave-uniform
            int64_t best = kacc == kmin ? nacc : -1;   // exactly one lane group proposed k = kmin; ids are >= 0
#pragma unroll
            for (int o = LPI; o < 64; o <<= 1) {
                const int olo = __shfl_xor((int)(best & 0xffffffff), o, 64), ohi = __shfl_xor((int)(best >> 32), o, 64);
                const int64_t other = ((int64_t)ohi << 32) | (uint32_t)olo;
                best = other > best ? other : best;
            }
            found = best;
        }
    }
    if (lane == 0) {
        p.unres[r] = found < 0 ? 1 : 0;
        if (found >= 0) p.idx[r] = found;
    }
}

template <int D>
int launch_reject(const SampleParams& p, hipStream_t st) {
    hipLaunchKernelGGL((catalog_sample_reject_kernel<D>), dim3((unsigned)cdiv(p.R, 4)), dim3(256), 0, st, p);
    return check_launch("catalog_sample_reject");
}

}  // namespace

namespace pcvae {

int catalog_sample_reject(const float* x, int64_t R, const float* E, int64_t N, int D, uint64_t seed, uint64_t row_offset,
                          const uint64_t* row_offset_dev, int64_t* idx, uint8_t* unres, hipStream_t st) {
    SampleParams p{x, E, R, N, seed, row_offset, ~0ull / (uint64_t)N, row_offset_dev, idx, unres};
    switch (D) {
        case 16: return launch_reject<16>(p, st);
        case 32: return launch_reject<32>(p, st);
        case 64: return launch_reject<64>(p, st);
        case 128: return launch_reject<128>(p, st);
        case 256: return launch_reject<256>(p, st);
    }
    set_error("catalog_sample: unsupported D=%d (16, 32, 64, 128, 256)", D);
    return PCVAE_EINVAL;
}

}  // namespace pcvae
