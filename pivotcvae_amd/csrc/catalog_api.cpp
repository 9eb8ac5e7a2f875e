// extern "C" entry points of the catalog kernels: argument validation + precision dispatch.
#include "catalog_plan.h"
#include <algorithm>

using namespace pcvae;

static bool supported_d(int D) { return D == 16 || D == 32 || D == 64 || D == 128 || D == 256; }

extern "C" size_t pcvae_catalog_ws_bytes(int64_t R, int64_t N, int D, int want_dx) {
    if (R <= 0 || N <= 0 || D <= 0) return 0;
    size_t need = 0;
    for (int prec : {PCVAE_PREC_F32, PCVAE_PREC_BF16}) {  // one answer valid for every precision mode
        const CatalogPlan pl = catalog_plan(R, N, D, prec);
        const size_t rows = (size_t)pl.nsplit * (size_t)R;
        const size_t ce = rows * 2 * sizeof(float) + (want_dx ? rows * (size_t)D * sizeof(float) : 0);
        const size_t am = (rows + 2) * sizeof(float) + rows * sizeof(int64_t) + 32 + (size_t)R;   // (+ the sampler's per-row flags)
        need = std::max(need, std::max(ce, am) + (size_t)pl.nrb + 64);
    }
    need = std::max(need, (size_t)R * (8 + 4 + 4 + 64 * 4) + 64);  // screened argmax: keys, thresholds, <= 64 partial maxima
    if (D == 128 || D == 256) need = std::max(need, catalog_x3_ws_bytes(R, N, D));
    return need + 256;
}

// bf16x3: its own partials (always with U), the row-block flags, then the f32 kernel's partials for flagged row blocks
size_t pcvae::catalog_x3_ws_bytes(int64_t R, int64_t N, int D) {
    // bf16x3 and bf16x6 lay their scratch out from their OWN plans; the two agree by default but not under PCVAE_RANGE_MB (a
    // table row is 4 D against 6 D bytes of stream): one answer that covers both
    const CatalogPlan pf = catalog_plan(R, N, D, PCVAE_PREC_F32);
    const size_t rf = (size_t)pf.nsplit * (size_t)R;
    size_t need = 0;
    for (int prec : {PCVAE_PREC_BF16X3, PCVAE_PREC_BF16X6}) {
        if (prec == PCVAE_PREC_BF16X6 && D != 128) continue;
        const CatalogPlan px = catalog_plan(R, N, D, prec);
        const size_t rx = (size_t)px.nsplit * (size_t)R;
        need = std::max(need, rx * (2 + (size_t)D) * sizeof(float) + (((size_t)px.nrb + 255) / 256) * 256 +
                                  rf * (2 + (size_t)D) * sizeof(float) + 64);
    }
    return need;
}

extern "C" int pcvae_catalog_ce_variant(int64_t R, int64_t N, int D, int prec) {
    if (R <= 0 || N <= 0 || !supported_d(D)) return -1;
    if (prec == PCVAE_PREC_F32) return 0;
    if (prec == PCVAE_PREC_BF16X3) return (D == 128 || D == 256) ? 3 : -1;
    if (prec == PCVAE_PREC_BF16X6) return D == 128 ? 4 : -1;
    if (prec != PCVAE_PREC_BF16 || (D != 64 && D != 128 && D != 256)) return -1;
    return catalog_bf16_pipelined(D, catalog_plan(R, N, D, PCVAE_PREC_BF16).tiles_per_split) ? 2 : 1;
}

extern "C" int pcvae_catalog_ce(const float* rx, int64_t R, const void* E, const void* E_lo, int64_t N, int D,
                                int prec, float e_max_norm, const int64_t* target, float keep_prob, uint64_t seed,
                                uint64_t row_offset, const uint8_t* keep_mask, float* nll, float* lse, float* dx,
                                void* ws, size_t ws_bytes, pcvae_stream_t stream) {
    return pcvae_catalog_ce_scaled(rx, R, E, E_lo, N, D, prec, e_max_norm, target, keep_prob, seed, row_offset, keep_mask, nll, lse,
                                   dx, 1.0f, ws, ws_bytes, stream);
}

extern "C" int pcvae_catalog_ce_scaled(const float* rx, int64_t R, const void* E, const void* E_lo, int64_t N, int D,
                                       int prec, float e_max_norm, const int64_t* target, float keep_prob, uint64_t seed,
                                       uint64_t row_offset, const uint8_t* keep_mask, float* nll, float* lse, float* dx,
                                       float dx_scale, void* ws, size_t ws_bytes, pcvae_stream_t stream) {
    PCVAE_REQUIRE(rx && E && target && nll && ws, "catalog_ce: null pointer");
    PCVAE_REQUIRE(R > 0 && N > 0, "catalog_ce: empty problem R=%lld N=%lld", (long long)R, (long long)N);
    PCVAE_REQUIRE(supported_d(D), "catalog_ce: unsupported D=%d (16, 32, 64, 128, 256)", D);
    PCVAE_REQUIRE(keep_prob > 0.f, "catalog_ce: keep_prob must be > 0 (n_neg > N raises in the reference too)");
    PCVAE_REQUIRE(((uintptr_t)rx % 16 == 0) && ((uintptr_t)E % 16 == 0) && ((uintptr_t)ws % 16 == 0) &&
                      (!dx || (uintptr_t)dx % 16 == 0),
                  "catalog_ce: rx/E/ws/dx must be 16-byte aligned");
    if (ws_bytes < pcvae_catalog_ws_bytes(R, N, D, dx != nullptr)) {
        set_error("catalog_ce: workspace %zu < %zu bytes", ws_bytes, pcvae_catalog_ws_bytes(R, N, D, dx != nullptr));
        return PCVAE_EWORKSPACE;
    }
    if (prec == PCVAE_PREC_F32) {
        (void)E_lo;
        (void)e_max_norm;
        return catalog_ce_f32(rx, R, reinterpret_cast<const float*>(E), N, D, target, keep_prob, seed, row_offset,
                              keep_mask, nll, lse, dx, dx_scale, ws, as_stream(stream));
    }
    if (prec == PCVAE_PREC_BF16) {
        (void)E_lo;
        return catalog_ce_bf16(rx, R, reinterpret_cast<const uint16_t*>(E), N, D, e_max_norm, target, keep_prob, seed,
                               row_offset, keep_mask, nll, lse, dx, dx_scale, ws, as_stream(stream));
    }
    if (prec == PCVAE_PREC_BF16X3 || prec == PCVAE_PREC_BF16X6) {
        // E = the split-bf16 image (pcvae_split_bf16x2: [N, 2 D] c0 | c1; pcvae_split_bf16x3: [N, 3 D] c0 | c1 | c2), E_lo = the
        // fp32 table itself (exact target logit / target row, and the exact f32 kernel for masked calls and for row blocks whose
        // norms rule out the max-free kernel)
        const bool x6 = prec == PCVAE_PREC_BF16X6;
        PCVAE_REQUIRE((D == 128 || (D == 256 && !x6)) && E_lo && ((uintptr_t)E_lo % 16 == 0) && e_max_norm > 0.f,
                      "catalog_ce(bf16x3 / bf16x6): needs D = 128 (bf16x3: or 256), the fp32 table in E_lo and e_max_norm > 0");
        if (keep_mask || keep_prob < 1.0f)
            return catalog_ce_f32(rx, R, reinterpret_cast<const float*>(E_lo), N, D, target, keep_prob, seed, row_offset,
                                  keep_mask, nll, lse, dx, dx_scale, ws, as_stream(stream));
        return catalog_ce_x3(rx, R, reinterpret_cast<const uint16_t*>(E), reinterpret_cast<const float*>(E_lo), N, D, x6 ? 3 : 2,
                             e_max_norm, target, nll, lse, dx, dx_scale, ws, as_stream(stream));
    }
    set_error("catalog_ce: precision mode %d not available in this build", prec);
    return PCVAE_EINVAL;
}

static int argmax_common(const float* x, int64_t R, const void* E, const void* E_lo, int64_t N, int D, int prec,
                         float e_max_norm, bool sample, uint64_t seed, uint64_t row_offset, int64_t* idx, float* best, void* ws,
                         size_t ws_bytes, pcvae_stream_t stream) {
    PCVAE_REQUIRE(x && E && idx && ws, "catalog_argmax: null pointer");
    PCVAE_REQUIRE(R > 0 && N > 0, "catalog_argmax: empty problem R=%lld N=%lld", (long long)R, (long long)N);
    PCVAE_REQUIRE(supported_d(D), "catalog_argmax: unsupported D=%d (16, 32, 64, 128, 256)", D);
    PCVAE_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)E % 16 == 0) && ((uintptr_t)ws % 16 == 0),
                  "catalog_argmax: x/E/ws must be 16-byte aligned");
    if (ws_bytes < pcvae_catalog_ws_bytes(R, N, D, 0)) {
        set_error("catalog_argmax: workspace %zu < %zu bytes", ws_bytes, pcvae_catalog_ws_bytes(R, N, D, 0));
        return PCVAE_EWORKSPACE;
    }
    if (prec == PCVAE_PREC_F32) {
        (void)E_lo;
        return catalog_argmax_f32(x, R, reinterpret_cast<const float*>(E), N, D, sample, seed, row_offset, idx, best,
                                  ws, as_stream(stream));
    }
    if (prec == PCVAE_PREC_SCREENED) {
        PCVAE_REQUIRE((D == 64 || D == 128 || D == 256) && !sample && E_lo && e_max_norm > 0.f && N < 0xffffffffLL &&
                          R < 0xffffffffLL && ((uintptr_t)E_lo % 16 == 0),
                      "catalog_argmax(screened): needs D in {64,128,256}, the fp32 table in E_lo, e_max_norm > 0 and N < 2^32");
        return catalog_argmax_screened(x, R, reinterpret_cast<const uint16_t*>(E), reinterpret_cast<const float*>(E_lo),
                                       N, D, e_max_norm, idx, best, ws, as_stream(stream));
    }
    set_error("catalog_argmax: precision mode %d not available in this build", prec);
    return PCVAE_EINVAL;
}

extern "C" int pcvae_catalog_argmax(const float* x, int64_t R, const void* E, const void* E_lo, int64_t N, int D,
                                    int prec, float e_max_norm, int64_t* idx, float* best, void* ws, size_t ws_bytes,
                                    pcvae_stream_t stream) {
    return argmax_common(x, R, E, E_lo, N, D, prec, e_max_norm, false, 0, 0, idx, best, ws, ws_bytes, stream);
}

extern "C" int pcvae_catalog_sample(const float* x, int64_t R, const void* E, const void* E_lo, int64_t N, int D,
                                    int prec, uint64_t seed, uint64_t row_offset, int64_t* idx, void* ws,
                                    size_t ws_bytes, pcvae_stream_t stream) {
    return pcvae_catalog_sample_at(x, R, E, E_lo, N, D, prec, seed, row_offset, nullptr, idx, ws, ws_bytes, stream);
}

extern "C" int pcvae_catalog_sample_at(const float* x, int64_t R, const void* E, const void* E_lo, int64_t N, int D,
                                       int prec, uint64_t seed, uint64_t row_offset, const uint64_t* row_offset_dev, int64_t* idx,
                                       void* ws, size_t ws_bytes, pcvae_stream_t stream) {
    PCVAE_REQUIRE(x && E && idx && ws, "catalog_sample: null pointer");
    PCVAE_REQUIRE(R > 0 && N > 0, "catalog_sample: empty problem R=%lld N=%lld", (long long)R, (long long)N);
    PCVAE_REQUIRE(supported_d(D), "catalog_sample: unsupported D=%d (16, 32, 64, 128, 256)", D);
    PCVAE_REQUIRE(prec == PCVAE_PREC_F32, "catalog_sample: scores are exact fp32 (E = the fp32 table); precision mode %d is not one "
                  "of its modes", prec);
    PCVAE_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)E % 16 == 0) && ((uintptr_t)ws % 16 == 0),
                  "catalog_sample: x/E/ws must be 16-byte aligned");
    if (ws_bytes < pcvae_catalog_ws_bytes(R, N, D, 0)) {
        set_error("catalog_sample: workspace %zu < %zu bytes", ws_bytes, pcvae_catalog_ws_bytes(R, N, D, 0));
        return PCVAE_EWORKSPACE;
    }
    (void)E_lo;
    // 1. rejection sampling: a handful of dot products per row; 2. rows that rejected every proposal (flagged, practically never):
    //    the exact Gumbel-max kernel over the whole catalog - its workgroups leave at once when none of their rows is flagged
    uint8_t* unres = reinterpret_cast<uint8_t*>(ws) + catalog_sample_unres_offset(R, N, D);
    const float* Ef = reinterpret_cast<const float*>(E);
    int rc = catalog_sample_reject(x, R, Ef, N, D, seed, row_offset, row_offset_dev, idx, unres, as_stream(stream));
    if (rc != PCVAE_OK) return rc;
    return catalog_argmax_f32(x, R, Ef, N, D, true, seed, row_offset, idx, nullptr, ws, as_stream(stream), unres, row_offset_dev);
}
