// Work decomposition of the catalog kernels (shared by the f32 and bf16 variants and by
// pcvae_catalog_ws_bytes, so the host sizes the workspace exactly as the kernels index it).
#pragma once
#include "common.h"

namespace pcvae {

struct CatalogPlan {
    int nrb;              // row blocks of 128 rows
    int ntiles;           // 32-item catalog tiles
    int nsplit;           // catalog ranges per row block (separate workgroups, merged afterwards)
    int tiles_per_split;
};

// Deterministic in (R, N, D) only - never in the device or the launch - so results are reproducible.
static inline CatalogPlan catalog_plan(int64_t R, int64_t N, int D) {
    (void)D;
    CatalogPlan p;
    p.nrb = (int)cdiv(R, 128);
    p.ntiles = (int)cdiv(N, 32);
    // aim for >= 2048 workgroups (256 CUs x 2 resident x 4 rounds) but keep >= 16 tiles per range so
    // the per-range prologue (rx load) and epilogue (partial write) stay amortised
    int64_t want = cdiv(2048, p.nrb);
    int64_t cap = std::max<int64_t>(1, p.ntiles / 16);
    int64_t ns = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(want, cap), 64));
    p.tiles_per_split = (int)cdiv(p.ntiles, ns);
    p.nsplit = (int)cdiv(p.ntiles, p.tiles_per_split);
    return p;
}

int catalog_ce_f32(const float* rx, int64_t R, const float* E, int64_t N, int D, const int64_t* target,
                   float keep_prob, uint64_t seed, uint64_t row_offset, const uint8_t* keep_mask, float* nll,
                   float* lse, float* dx, void* ws, hipStream_t st);
int catalog_argmax_f32(const float* x, int64_t R, const float* E, int64_t N, int D, bool sample, uint64_t seed,
                       uint64_t row_offset, int64_t* idx, float* best, void* ws, hipStream_t st);

}  // namespace pcvae
