/* CPU oracle for the catalog kernels (K5/K6 of SURVEY.md 2.1).  TEST INFRASTRUCTURE ONLY.
 *
 * Restates, in plain C, the arithmetic the HIP f32 catalog kernels perform so that parity can be
 * checked BIT-EXACTLY: every score is the k-ordered single-rounding chain
 *     acc = 0;  for k in 0..D-1: acc = fmaf(x[r][k], E[n][k], acc)
 * which is what v_mfma_f32_32x32x2_f32 produces (cdna_hip_programming.md section 3,
 * "FP32-input MFMA: ... bit-for-bit a k-ordered f32 fmaf chain").
 *
 * Reference behaviour restated (relative to /root/reference):
 *   catalog_argmax   models/cvae.py:97-101 (mm + max(1): first maximal index)
 *                    models/pivotcvae.py:191 (mm(E, pivot_output^T).max(0))
 *   catalog_ce       models/pivotcvae.py:274 (p = rx @ E^T) + train_generative.py:36-42,59
 *                    (downsample: masked-out logits := 0, target always kept; CrossEntropyLoss
 *                    mean over rows) - here per-row values in double so the caller chooses the mean.
 *
 * Pinned through tests/test_catalog_oracle.py: the argmax equals the golden item ids minted from the
 * reference (all golden rows have a top-2 margin far above fp32 rounding), and the CE equals the
 * torch oracle's to 1e-6.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#if defined(__x86_64__)
#define HW_FMA __attribute__((target("fma")))
#else
#define HW_FMA
#endif

static inline float dot_sw(const float* x, const float* e, int D) {
    float acc = 0.0f;
    for (int k = 0; k < D; ++k) acc = fmaf(x[k], e[k], acc);
    return acc;
}
HW_FMA static inline float dot_hw(const float* x, const float* e, int D) {
    float acc = 0.0f;
    for (int k = 0; k < D; ++k) acc = __builtin_fmaf(x[k], e[k], acc);
    return acc;
}

static void scores_sw(const float* x, const float* E, int64_t R, int64_t N, int D, float* out) {
    for (int64_t r = 0; r < R; ++r)
        for (int64_t n = 0; n < N; ++n) out[r * N + n] = dot_sw(x + r * D, E + n * D, D);
}
HW_FMA static void scores_hw(const float* x, const float* E, int64_t R, int64_t N, int D, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; ++r)
        for (int64_t n = 0; n < N; ++n) out[r * N + n] = dot_hw(x + r * D, E + n * D, D);
}

static int has_hw_fma(void) {
#if defined(__x86_64__)
    return __builtin_cpu_supports("fma");
#else
    return 0;
#endif
}

/* out[r][n] = <x_r, E_n> as the k-ordered fmaf chain. */
void catalog_scores_fma(const float* x, const float* E, int64_t R, int64_t N, int D, float* out) {
    if (has_hw_fma()) scores_hw(x, E, R, N, D, out); else scores_sw(x, E, R, N, D, out);
}

/* idx[r] = first n maximising the chain score; best[r] (optional) = that score. */
void catalog_argmax_fma(const float* x, const float* E, int64_t R, int64_t N, int D, int64_t* idx, float* best) {
    const int hw = has_hw_fma();
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; ++r) {
        float bv = -INFINITY;
        int64_t bi = 0;
        for (int64_t n = 0; n < N; ++n) {
            float v = hw ? dot_hw(x + r * D, E + n * D, D) : dot_sw(x + r * D, E + n * D, D);
            if (v > bv) { bv = v; bi = n; }
        }
        idx[r] = bi;
        if (best) best[r] = bv;
    }
}

/* Per-row masked softmax cross-entropy over the whole catalog, and its gradient direction.
 *   keep: NULL (all kept) or uint8 [R][N]; the target column is always kept.
 *   z_n   = keep ? score : 0          (train_generative.py:42: pred * mask)
 *   lse_r = log sum_n exp(z_n)        (double accumulation, max-shifted)
 *   nll_r = lse_r - z_target
 *   dx_r  = sum_n keep_n * (softmax_n - [n == target]) * E_n     (NOT yet divided by the row count)
 * nll/lse are double, dx (optional) is float [R][D]. */
void catalog_ce_fma(const float* x, const float* E, const int64_t* target, const uint8_t* keep, int64_t R,
                    int64_t N, int D, double* nll, double* lse_out, float* dx) {
    const int hw = has_hw_fma();
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; ++r) {
        const int64_t t = target[r];
        double m = -INFINITY;
        for (int64_t n = 0; n < N; ++n) {
            int k = (n == t) || !keep || keep[r * N + n];
            double z = k ? (double)(hw ? dot_hw(x + r * D, E + n * D, D) : dot_sw(x + r * D, E + n * D, D)) : 0.0;
            if (z > m) m = z;
        }
        double sum = 0.0, zt = 0.0;
        for (int64_t n = 0; n < N; ++n) {
            int k = (n == t) || !keep || keep[r * N + n];
            double z = k ? (double)(hw ? dot_hw(x + r * D, E + n * D, D) : dot_sw(x + r * D, E + n * D, D)) : 0.0;
            if (n == t) zt = z;
            sum += exp(z - m);
        }
        const double lse = m + log(sum);
        nll[r] = lse - zt;
        if (lse_out) lse_out[r] = lse;
        if (dx) {
            double acc[512];
            for (int d = 0; d < D; ++d) acc[d] = 0.0;
            for (int64_t n = 0; n < N; ++n) {
                int k = (n == t) || !keep || keep[r * N + n];
                if (!k) continue;
                double z = (double)(hw ? dot_hw(x + r * D, E + n * D, D) : dot_sw(x + r * D, E + n * D, D));
                double g = exp(z - lse) - (n == t ? 1.0 : 0.0);
                for (int d = 0; d < D; ++d) acc[d] += g * (double)E[n * D + d];
            }
            for (int d = 0; d < D; ++d) dx[r * D + d] = (float)acc[d];
        }
    }
}
