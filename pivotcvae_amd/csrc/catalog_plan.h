// Work decomposition of the catalog kernels (shared by the f32 and bf16 variants and by
// pcvae_catalog_ws_bytes, so the host sizes the workspace exactly as the kernels index it).
#pragma once
#include "common.h"
#include <algorithm>
#include <cstdlib>

#ifndef X3_CT
#define X3_CT 2   // column tiles of 16 rows per wave of the bf16x3 kernel: 32 rows per wave, 128 per workgroup (3 and 4 spill: HISTORY.md 3.1b)
#endif
#ifndef X3_CT256
#define X3_CT256 1   // D = 256: 16 rows per wave, 64 per workgroup (U alone is 128 accumulator registers at 32 rows: hipcc then spills)
#endif
static inline constexpr int x3_ct(int D) { return D == 256 ? X3_CT256 : X3_CT; }
#ifndef PIPE_CT64
#define PIPE_CT64 4    // column tiles of 16 rows per wave of the D = 64 bf16 cross-entropy kernel.  2 (128-row workgroups of 256 registers per
                       // lane, TWO resident per CU - what gave the D = 64 screening kernels 17 %) was built and measured in round 6, same box,
                       // alternating: config 3's kernel 0.865 vs 0.842 ms - its slot is bound by the exponentials (one quarter-rate v_exp_f32
                       // per MFMA), which two waves share no better than one, and the A fragments are read twice.  4 stays.
#endif
#ifndef PIPE_CT256
#define PIPE_CT256 2   // column tiles of 16 rows per wave of the D = 256 bf16 cross-entropy kernel (64 PIPE_CT256 rows per workgroup).
                       // 3 (48 rows per wave, a third less LDS traffic per MFMA) builds with a copy-free steady-state loop but 592 bytes
                       // of scratch per lane in the fenced paths; measured round 4: N = 1M 27.5 ms (no gain), config 5 five times
                       // SLOWER, one parity failure at R = 300 - not pursued
#endif

#ifndef PCVAE_PLAN_ROUND_TILES_DEFAULT
#define PCVAE_PLAN_ROUND_TILES_DEFAULT 54   // per-round overhead of the bf16-pipe kernels in tile-times (catalog_plan's cost model).
                                            // Measured, round 5 (tools/nsplit_sweep.sh, profiles/r05_nsplit_sweep_config3_bf16.txt):
                                            // config 3 with 8 ranges (5 rounds of 392 tiles) 0.934 ms, with 3 ranges (2 rounds of 1044)
                                            // 0.920 ms: 5 (392 + X) / (2 (1044 + X)) = 1.015 -> X = 54.  Changes the choice at config 3
                                            // (8 -> 3 ranges) and for its small shards (38 -> 12 ranges at B = 512); configs 4, 5 keep theirs
#endif

#ifndef PCVAE_RANGE_MB_DEFAULT
#define PCVAE_RANGE_MB_DEFAULT 0   // longest catalog range in MB of table stream (0: no limit); see catalog_plan
#endif

namespace pcvae {

struct CatalogPlan {
    int nrb;              // row blocks of 128 rows
    int ntiles;           // 32-item catalog tiles
    int nsplit;           // catalog ranges per row block (separate workgroups, merged afterwards)
    int tiles_per_split;
};

// Deterministic in (R, N, D, precision) only - never in the device or the launch - so results are reproducible.
//   f32 kernels : 128-row workgroups (4 waves), 2 resident per CU, any tile count per range
//   bf16 kernels: 256-row workgroups (8 waves), 1 resident per CU, ranges are whole 128-item LDS chunks
//   (wg_per_cu: resident workgroups per CU of the kernel the plan is for, where that is not the precision's default - the pipelined
//   screening kernels at D <= 128 run two)
static inline CatalogPlan catalog_plan(int64_t R, int64_t N, int D, int prec, int wg_per_cu = 1) {
    const bool f32 = prec == PCVAE_PREC_F32;
    // split-bf16 kernels (bf16x3: table rows of 4 D bytes, the geometry of the D = 256 bf16 kernel; bf16x6: 6 D bytes): one plan
    const bool x3 = prec == PCVAE_PREC_BF16X3 || prec == PCVAE_PREC_BF16X6;
    const int rows_wg = f32 ? 128 : 256;
    const int quant = (f32 || x3) ? 1 : 4;
    CatalogPlan p;
    p.nrb = (int)cdiv(R, rows_wg);
    p.ntiles = (int)cdiv(N, 32);
    // Choose the number of catalog ranges so that the grid is (nearly) a whole number of rounds of resident
    // workgroups: the cost model is rounds x tiles-per-range (e.g. 40 row blocks: 26 ranges = 1040 workgroups =
    // 4.06 rounds -> 5 rounds of 1202 tiles; 32 ranges = 1280 workgroups = exactly 5 rounds of 980 tiles, 19 % less).
    // Ranges stay long enough that the per-range prologue (rx fragments) and epilogue (partials) are amortised.
    const int64_t slots = 256 * (f32 ? 2 : 1) * wg_per_cu;
    // the bf16 fast kernel for D = 256 runs 128-row workgroups (4 waves, one per SIMD): twice the workgroups per range
    const int64_t nblk = x3 ? cdiv(R, 64 * x3_ct(D)) : ((!f32 && D == 256) ? cdiv(R, 64 * PIPE_CT256) : p.nrb);
    const int64_t cap = std::max<int64_t>(1, std::min<int64_t>(64, p.ntiles / (f32 ? 16 : 64)));
    // fp32-grade results (exact f32, bf16x3): a range's row sums and U are ONE fp32 accumulation chain, and past ~1M items the
    // running sum's ulp swallows the small terms of a peaked row (N = 10M, |logit| up to 6: lse 6e-5 low, round 3).  Ranges of at
    // most 16 384 tiles (512K items); the merge adds the <= 64 partials.  (bf16: its own 2^-9 per term dwarfs that - plan unchanged.)
    int64_t ns_min = (f32 || x3) ? std::min<int64_t>(cap, cdiv(p.ntiles, 16384)) : 1;
    // Range-length limit (experiments only; default off).  The 32 workgroups an XCD runs side by side stream the SAME catalog range
    // and share it in the XCD's 4 MB L2 while they stay within ~2 MB of each other in that stream; at config 5 (5 - 10 GB images)
    // they do not (PMC, round 4: 573 GB per launch for the bf16 kernel = 2.8x the shared ideal, 17.3 TB for bf16x3).  Hypothesis:
    // they drift apart over a long pass, so shorter ranges would re-align them.  Refuted (experiments/tools/range_sweep.sh,
    // profiles/r04_config5_range_sweep.txt): ranges of <= 1024 / 256 / 96 MB move the reads by 17 % and the time by +0.2 .. +0.4 %;
    // the memory-side bytes do not set these kernels' time.  The knob stays for measurements.
    const int64_t row_bytes = (int64_t)D * (f32 ? 4 : prec == PCVAE_PREC_BF16X6 ? 6 : x3 ? 4 : 2);
    const char* env_mb = getenv("PCVAE_RANGE_MB");   // (experiments: experiments/tools/range_sweep.sh; read per call like PCVAE_PIPE_MIN_TILES)
    const int64_t range_mb = env_mb ? atoll(env_mb) : PCVAE_RANGE_MB_DEFAULT;
    if (range_mb > 0)
        ns_min = std::max<int64_t>(ns_min, std::min<int64_t>(cap, cdiv((int64_t)p.ntiles * 32 * row_bytes, range_mb << 20)));
    // Every round of resident workgroups pays a fill (rx fragments, the first ring chunks) and a drain (partials stored, the tail
    // wave of the slowest CU) on top of its tiles: PLAN_ROUND_TILES is that overhead in tile-times.  0 for the exact f32 kernels
    // (tile time 16x longer: the overhead is noise there).  PCVAE_PLAN_ROUND_TILES / PCVAE_PLAN_NSPLIT (environment, read per call):
    // experiments only (tools/nsplit_sweep.sh: the sweep behind the default).
    const char* env_rt = getenv("PCVAE_PLAN_ROUND_TILES");
    const int64_t round_tiles = env_rt ? atoll(env_rt) : (f32 ? 0 : PCVAE_PLAN_ROUND_TILES_DEFAULT);
    const char* env_ns = getenv("PCVAE_PLAN_NSPLIT");
    const int64_t ns_forced = env_ns ? std::max<int64_t>(1, std::min<int64_t>(cap, atoll(env_ns))) : 0;
    int64_t best_cost = -1;
    for (int64_t ns = ns_forced ? ns_forced : ns_min; ns <= (ns_forced ? ns_forced : cap); ++ns) {
        const int64_t tps = cdiv(cdiv(p.ntiles, ns), quant) * quant;
        const int64_t ns_eff = cdiv(p.ntiles, tps);
        const int64_t rounds = cdiv(nblk * ns_eff, slots);
        const int64_t cost = rounds * (tps + round_tiles);
        if (best_cost < 0 || cost < best_cost) {  // strict '<': ties keep the fewer, longer ranges
            best_cost = cost;
            p.tiles_per_split = (int)tps;
            p.nsplit = (int)ns_eff;
        }
    }
    // software-pipelined bf16 kernel for D = 256: its steady-state trip is 6 slots after one fill slot, what is left over
    // runs fenced; a range of 12k+8 tiles leaves 1 such slot instead of up to 5.  (Applied after the choice above, so that
    // the few extra tiles do not bias it; D = 128 hands the rest of a range to the other kernel, D = 64 runs it at speed.)
    if (!f32 && D == 256 && p.tiles_per_split >= 64) {
        int64_t tps = p.tiles_per_split;
        tps += ((8 - tps % 12) + 12) % 12;
        p.tiles_per_split = (int)tps;
        p.nsplit = (int)cdiv(p.ntiles, tps);
    }
    // bf16x3: one fill slot + steady-state trips of 6 slots (4 at D = 256: two ring chunks per slot); 6k+1 (4k+1) tiles leave no
    // fenced slot at all
    if (x3 && p.tiles_per_split >= 64) {
        const int64_t trip = D == 256 ? 4 : 6;
        int64_t tps = p.tiles_per_split;
        tps += ((1 - tps % trip) + trip) % trip;
        p.tiles_per_split = (int)tps;
        p.nsplit = (int)cdiv(p.ntiles, tps);
    }
    return p;
}

// bf16 training call: software-pipelined kernel (one wave per SIMD) or the two-waves-per-SIMD kernel?  D = 256 always
// pipelined; D = 128 / 64 on long catalog ranges only (the pipeline's fill / drain / last slots run fenced, at about half
// speed).  PCVAE_PIPE_MIN_TILES (environment, read per call) overrides the threshold: the tests use it.
static inline bool catalog_bf16_pipelined(int D, int tiles_per_split) {
    if (D == 256) return true;
    const char* env_min = getenv("PCVAE_PIPE_MIN_TILES");
    // D = 64 runs what is left after its last full trip at steady speed, so it pays off from shorter ranges on (config 3: 391
    // tiles per range, 0.939 against 0.967 ms per launch - round 3)
    return tiles_per_split >= (env_min ? atoi(env_min) : (D == 64 ? 256 : 512));
}

int catalog_ce_f32(const float* rx, int64_t R, const float* E, int64_t N, int D, const int64_t* target,
                   float keep_prob, uint64_t seed, uint64_t row_offset, const uint8_t* keep_mask, float* nll,
                   float* lse, float* dx, float dx_scale, void* ws, hipStream_t st);
int catalog_ce_bf16(const float* rx, int64_t R, const uint16_t* E, int64_t N, int D, float e_max_norm,
                    const int64_t* target, float keep_prob, uint64_t seed, uint64_t row_offset,
                    const uint8_t* keep_mask, float* nll, float* lse, float* dx, float dx_scale, void* ws, hipStream_t st);
int catalog_ce_x3(const float* rx, int64_t R, const uint16_t* Ex, const float* Ef, int64_t N, int D, int ncomp, float e_max_norm,
                  const int64_t* target, float* nll, float* lse, float* dx, float dx_scale, void* ws, hipStream_t st);
// the exact f32 kernel restricted to the 256-row blocks whose flag is 1 (the fallback of the max-free bf16x3 kernel)
int catalog_ce_f32_flagged(const float* rx, int64_t R, const float* E, int64_t N, int D, const int64_t* target, float* nll,
                           float* lse, float* dx, float dx_scale, void* ws, const uint8_t* flags, hipStream_t st);
size_t catalog_x3_ws_bytes(int64_t R, int64_t N, int D);
int catalog_argmax_screened(const float* x, int64_t R, const uint16_t* Eb, const float* Ef, int64_t N, int D, float e_max_norm,
                            int64_t* idx, float* best, void* ws, hipStream_t st);
// sample = true: Gumbel-max draw from Categorical(sigmoid(scores)); with `unres` [R] only for the rows flagged there
int catalog_argmax_f32(const float* x, int64_t R, const float* E, int64_t N, int D, bool sample, uint64_t seed,
                       uint64_t row_offset, int64_t* idx, float* best, void* ws, hipStream_t st, const uint8_t* unres = nullptr,
                       const uint64_t* row_offset_dev = nullptr);
// rejection sampler of Categorical(sigmoid(scores)) (catalog_sample.hip): idx[r] for every row it resolves, unres[r] = 1 otherwise
int catalog_sample_reject(const float* x, int64_t R, const float* E, int64_t N, int D, uint64_t seed, uint64_t row_offset,
                          const uint64_t* row_offset_dev, int64_t* idx, uint8_t* unres, hipStream_t st);
// where the sampler's per-row flags live in the catalog workspace: behind the argmax partials of the f32 plan
static inline size_t catalog_sample_unres_offset(int64_t R, int64_t N, int D) {
    const CatalogPlan pl = catalog_plan(R, N, D, PCVAE_PREC_F32);
    const size_t rows = (size_t)pl.nsplit * (size_t)R;
    return (((rows + 1) & ~(size_t)1) * sizeof(float) + rows * sizeof(int64_t) + 15) & ~(size_t)15;
}

}  // namespace pcvae
