/* pcvae.h - C ABI of libpcvae_hip.so: the MI355X (gfx950) kernels behind the PivotCVAE
 * slate-generation hot path.
 *
 * The reference (CharlieMat/PivotCVAE) has no FFI/plugin interface: its hot path is a chain of aten
 * calls inside Python.  Each entry point below replaces one such call site; the
 * reference file:line it stands in for is cited per function (paths relative to /root/reference).
 *
 * Conventions
 *   - plain pointers + sizes; every pointer is DEVICE memory owned by the caller (torch's caching
 *     allocator on the Python host side); nothing is allocated, freed or retained by the library.
 *   - all matrices are row-major fp32 with an explicit leading dimension `ld*` (floats between rows),
 *     so concatenations (torch.cat in the reference) are just column windows of one buffer.
 *   - indices are int64 (torch.LongTensor in the reference).
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on that stream and
 *     graph-capturable (no allocation, no synchronisation, no host read-back).
 *   - return value: 0 on success, negative PCVAE_E* otherwise; pcvae_last_error() gives the text
 *     (thread-local).  Shape/argument violations are rejected on the host BEFORE any launch.
 */
#ifndef PCVAE_H
#define PCVAE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCVAE_ABI_VERSION 3   /* 2 (round 5): pcvae_catalog_ce_sparse_scaled takes the table's precision and a device-word seed;
                                 pcvae_candidate_ce, pcvae_catalog_sample_at, pcvae_set_words, pcvae_gather_rows_variant are new
                                 3 (round 6): pcvae_candidate_ce takes n_items (the dataset's id range of the in-kernel draw);
                                 PCVAE_GEMM_X6 */

#define PCVAE_OK 0
#define PCVAE_EINVAL (-1)   /* bad argument / unsupported shape */
#define PCVAE_ELAUNCH (-2)  /* HIP launch error */
#define PCVAE_EWORKSPACE (-3) /* workspace too small */

#define PCVAE_ACT_NONE 0
#define PCVAE_ACT_LEAKY 1   /* LeakyReLU(0.01): models/cvae.py:43 */
#define PCVAE_ACT_RELU 2    /* F.relu of the response model: env/response_model.py:85 (forward only) */

/* catalog precision modes (arithmetic the [R,D]x[D,N] contraction is computed in) */
#define PCVAE_PREC_F32 0    /* v_mfma_f32_32x32x2_f32: exact k-ordered fmaf chain (bit-exact ids) */
#define PCVAE_PREC_BF16 1   /* v_mfma_f32_32x32x16_bf16 on a bf16 copy of the table, fp32 accumulate */
#define PCVAE_PREC_BF16X3 2 /* fp32-equivalent on the bf16 MFMA pipe: hi/lo bf16 split of both operands, 3 MFMAs per product
                               (D = 128 or 256; E = the bf16 image of pcvae_split_bf16x2, E_lo = the fp32 table.  Narrower
                               tables: the caller zero-pads table and rx to 128 columns - a zero column adds exactly 0) */
#define PCVAE_PREC_BF16X6 4 /* the reference's fp32 arithmetic on the bf16 MFMA pipe: both operands as THREE bf16 components
                               (c0 + c1 + c2 = the fp32 value exactly), 6 MFMAs per product (the dropped pairs are <= 2^-25 relative),
                               fp32 accumulate (D = 128; E = the image of pcvae_split_bf16x3, E_lo = the fp32 table; narrower tables
                               zero-padded to 128 columns by the caller) */
#define PCVAE_PREC_SCREENED 3 /* argmax only: bf16 MFMA screening + exact fp32 rescoring of the few candidates;
                                results identical to PCVAE_PREC_F32 (E = bf16 table, E_lo = fp32 table, D = 64 / 128 / 256) */

typedef void* pcvae_stream_t;

int pcvae_abi_version(void);
const char* pcvae_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * K1  item / user embedding gather          models/pivotcvae.py:253,258 ; :194 ; :192
 *     out[(i / group) * out_ld + (i % group) * D + d] = table[idx[i] * D + d],  i in [0, n_idx)
 *     group = 1, out_ld = D is a plain nn.Embedding lookup; group = S, out_ld = enc_in writes the
 *     [B, S*D] slate block straight into the encoder input buffer.
 * ------------------------------------------------------------------------------------------- */
int pcvae_gather_rows(const float* table, int64_t n_rows, int D, const int64_t* idx, int64_t n_idx,
                      int group, float* out, int64_t out_ld, pcvae_stream_t stream);
/* the kernel the call above launches for a shape (table / out 16-byte aligned), for labelling measurements with the name a
 * kernel trace shows: 0 = gather_rows_scalar_kernel, 1 = gather_rows_vec4_kernel, 2 = gather_rows_coal_kernel (D = 64 / 128 / 256) */
int pcvae_gather_rows_variant(int D, int group, int64_t out_ld);

/* K2  click-count one-hot condition         models/cvae.py:85-92
 *     out[b * out_ld + c] = (c == sum_{j < ncols} r[b, j]),  c in [0, S].  r is [B, ncols]: ncols is the slate size S in
 *     training, but the reference's in-loop evaluation passes a 5-column context whatever S is
 *     (train_generative.py:179) - only the row sum matters.                                       */
int pcvae_condition(const float* r, int64_t B, int ncols, int S, float* out, int64_t out_ld, pcvae_stream_t stream);

/* strided 2-D copy (the torch.cat of models/pivotcvae.py:167,203,213,236 becomes column windows) */
int pcvae_copy2d(const float* src, int64_t src_ld, float* dst, int64_t dst_ld, int64_t rows, int cols,
                 pcvae_stream_t stream);
/* torch.cat(parts, 1) of up to four column blocks in one launch (models/pivotcvae.py:167,205,232: the inputs of the
 * encoder / prior / slate-completion stacks).  Part i is [rows, c_i] with leading dimension ld_i; unused trailing parts: c_i = 0. */
int pcvae_concat(const float* s0, int64_t ld0, int c0, const float* s1, int64_t ld1, int c1, const float* s2, int64_t ld2,
                 int c2, const float* s3, int64_t ld3, int c3, float* dst, int64_t dst_ld, int64_t rows,
                 pcvae_stream_t stream);

/* out[r, c] = x[r, c] * scale_host * (scale_dev ? *scale_dev : 1)   (chain rule through 'mean') */
int pcvae_scale_rows(const float* x, int64_t ldx, float* out, int64_t ldo, int64_t rows, int cols,
                     const float* scale_dev, float scale_host, pcvae_stream_t stream);

/* ---- in-loop evaluation against the response model (train_generative.py:169-195, env/response_model.py:76-87) ----
 * normalize_rows : x[r, :] /= max(||x[r, :]||_2, 1e-12) in place (F.normalize of the WHOLE concatenated slate vector)
 * click_stats    : nc[b] = sum_s sigmoid(logits[b, s]);  out[0..2] = (min_b nc, mean_b nc, max_b nc)
 * philox_randint : out[i] = Philox(seed, offset + i) mod hi  (sample_users: uniform user ids, env/response_model.py:10-13) */
int pcvae_normalize_rows(float* x, int64_t ldx, int64_t rows, int cols, pcvae_stream_t stream);
int pcvae_click_stats(const float* logits, int64_t B, int S, float* nc, float* out3, pcvae_stream_t stream);
int pcvae_philox_randint(int64_t* out, int64_t n, int64_t hi, uint64_t seed, uint64_t offset, pcvae_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K3  MLP layers (addmm + leaky_relu)       models/pivotcvae.py:167-173, 205-210, 215-220, 232-239
 *     fwd : Y[M,N]  = act(X[M,K] @ W[N,K]^T + bias[N])          (bias may be NULL: dense scores,
 *                                                                 models/pivotcvae.py:274)
 *     dX  : dX[M,K] = (dY[M,N] @ W[N,K]) * act'(Xact[M,K])      (Xact = ACTIVATED output that fed
 *                                                                 this layer, NULL -> no act')
 *     dW  : dW[N,K] += dY[M,N]^T @ X[M,K] ;  db[N] += colsum(dY)   (accumulating; db may be NULL).  The single-layer entry point
 *           runs the batch as ONE split (no scratch buffer to combine splits through): pcvae_linear_group is the fast path
 * ------------------------------------------------------------------------------------------- */
int pcvae_linear_fwd(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, float* Y,
                     int64_t ldy, int64_t M, int64_t N, int64_t K, int act, pcvae_stream_t stream);
int pcvae_linear_bwd_input(const float* dY, int64_t lddy, const float* W, int64_t ldw, const float* Xact,
                           int64_t ldxa, float* dX, int64_t lddx, int64_t M, int64_t N, int64_t K,
                           pcvae_stream_t stream);
/* dX = (dX + dY . W) * LeakyReLU'(Xact): the second of two Linear layers fed by the same activated input (the mu / logvar
 * heads, models/pivotcvae.py:214-220,236-239): their input gradients are summed and masked in the GEMM epilogue. */
int pcvae_linear_bwd_input_acc(const float* dY, int64_t lddy, const float* W, int64_t ldw, const float* Xact, int64_t ldxa,
                               float* dX, int64_t lddx, int64_t M, int64_t N, int64_t K, pcvae_stream_t stream);
int pcvae_linear_bwd_weight(const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int64_t lddw,
                            float* db, int64_t M, int64_t N, int64_t K, pcvae_stream_t stream);
/* Several INDEPENDENT layer GEMMs as ONE launch (no problem may read another's output; outputs must not overlap): a layer's
 * weight- and input-gradient, or the same layer of two stacks that do not feed each other (encoder and prior:
 * models/pivotcvae.py:159-174 and 229-240 share only their inputs).  Field meaning per kind = the arguments of the single-layer
 * entry points above:
 *     kind               a, lda     b, ldb    c, ldc     aux, ldaux             aux_out   M, N, K
 *     PCVAE_GEMM_FWD     X, ldx     W, ldw    Y, ldy     bias (or NULL), -      -         as pcvae_linear_fwd (+ act)
 *     PCVAE_GEMM_DX      dY, lddy   W, ldw    dX, lddx   Xact (or NULL), ldxa   -         as pcvae_linear_bwd_input
 *     PCVAE_GEMM_DX_ACC  (the same; dX = (dX + dY.W) * act'(Xact), as pcvae_linear_bwd_input_acc)
 *     PCVAE_GEMM_DW      dY, lddy   X, ldx    dW, lddw   -                      db / NULL as pcvae_linear_bwd_weight
 * 1 <= n <= 6.  Problems with M == 0 are skipped.
 * ws (pcvae_linear_group_ws_bytes(descs, n) bytes, ZERO when first used and private to the stream; a launch leaves its counters
 * zero again): the batch splits of a weight gradient store their partial tiles there and the last split to arrive adds their sum,
 * in split order, to dW / db - one writer per element, bitwise reproducible from run to run.  (There are NO fp32 atomics on
 * gradients: atomicAdd(float*) between workgroups of different XCDs loses updates on this part, tools/atomic_tile_probe.hip.)
 * ws == NULL: every weight gradient of the launch runs as ONE batch split - correct, and slow for large layers. */
enum { PCVAE_GEMM_FWD = 0, PCVAE_GEMM_DX = 1, PCVAE_GEMM_DX_ACC = 2, PCVAE_GEMM_DW = 3 };
/* OR-ed into `kind`: run this problem in bf16x3 arithmetic (operands split into bf16 hi + lo in registers, three bf16 MFMAs per
 * product, fp32 accumulate: 16-bit-mantissa operands, held to the GEMM tests' tolerances, ~2x faster on MFMA-bound layers).  Takes
 * effect when EVERY problem of a launch carries it (such a launch always uses the 64 x 64 tiles, whatever its size, so a layer's
 * arithmetic does not depend on the batch); a launch with mixed flags computes in exact fp32.                                    */
#define PCVAE_GEMM_X3 0x100
/* OR-ed into `kind` (ABI 3): run this problem in bf16x6 arithmetic - every fp32 operand as THREE bf16 components whose sum is the fp32
 * value exactly, six bf16 MFMAs per product (the dropped component pairs are <= 2^-25 relative), fp32 accumulate: the reference's
 * fp32 products on the bf16 matrix cores, as PCVAE_PREC_BF16X6 of the catalog kernels.  Same launch rule as PCVAE_GEMM_X3 (every
 * problem of the launch must carry it; with X3 and X6 mixed the launch runs bf16x3).                                            */
#define PCVAE_GEMM_X6 0x200
typedef struct pcvae_gemm_desc {
    int32_t kind;
    int32_t act;
    const float* a;
    int64_t lda;
    const float* b;
    int64_t ldb;
    float* c;
    int64_t ldc;
    const float* aux;
    int64_t ldaux;
    float* aux_out;
    int64_t M, N, K;
} pcvae_gemm_desc;
size_t pcvae_linear_group_ws_bytes(const pcvae_gemm_desc* descs, int n);
int pcvae_linear_group(const pcvae_gemm_desc* descs, int n, void* ws, size_t ws_bytes, pcvae_stream_t stream);
/* in place: g *= (y > 0 ? 1 : 0.01)  - LeakyReLU backward keyed on the activated output */
int pcvae_leaky_bwd(float* g, int64_t ldg, const float* y, int64_t ldy, int64_t rows, int cols,
                    pcvae_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K4  reparameterisation                    models/cvae.py:79-83
 *     z = mu + exp(0.5 * logvar) * eps.   eps_in != NULL: use it (parity mode, the reference draws
 *     eps from torch's global generator).  eps_in == NULL: eps ~ N(0,1) from Philox4x32-10 keyed by
 *     `seed`, counter = offset + element index (results independent of launch geometry and of how a
 *     batch is sharded over ranks).  eps_out (optional) receives the eps used (needed by backward).
 *     bwd: dmu += dz ; dlogvar += dz * eps * 0.5 * exp(0.5 * logvar)
 * ------------------------------------------------------------------------------------------- */
int pcvae_reparam_fwd(const float* mu, const float* logvar, const float* eps_in, uint64_t seed, uint64_t offset,
                      float* z, int64_t ldz, float* eps_out, int64_t B, int Z, pcvae_stream_t stream);
/* out[i] = N(0,1) sample i of the same Philox stream pcvae_reparam_fwd uses (counter = offset + i): lets a caller
 * draw eps outside a captured hipGraph and feed it in through eps_in (kernel arguments are frozen at capture) */
int pcvae_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, pcvae_stream_t stream);
int pcvae_reparam_bwd(const float* dz, int64_t lddz, const float* eps, const float* logvar, float* dmu,
                      float* dlogvar, int64_t B, int Z, pcvae_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K7  Gaussian-prior KL (sum over batch and latent)     train_generative.py:61
 *     kld = -1/2 sum(1 + lv - plv - (exp(lv) + (mu - pmu)^2) / exp(plv))  -> *kld_out (1 float)
 *     bwd: d{mu,lv,pmu,plv} += scale * dKLD/d{...}   (any of the four outputs may be NULL)
 * ------------------------------------------------------------------------------------------- */
int pcvae_kld_fwd(const float* mu, const float* lv, const float* pmu, const float* plv, int64_t n, float* kld_out,
                  pcvae_stream_t stream);
int pcvae_kld_bwd(const float* mu, const float* lv, const float* pmu, const float* plv, int64_t n,
                  const float* scale_dev, float scale_host, float* dmu, float* dlv, float* dpmu, float* dplv,
                  pcvae_stream_t stream);

/* K4 + K7 backward fused (the posterior's mu / logvar get a gradient from both): WRITES
 *   dmu = dz + s dKLD/dmu, dlv = dz eps exp(lv/2)/2 + s dKLD/dlv, dpmu = s dKLD/dpmu, dplv = s dKLD/dplv,
 * s = dkld_host * (dkld_dev ? *dkld_dev : 1): replaces reparam_bwd + kld_bwd + their zero-fills + autograd's adds
 * (models/cvae.py:79-83 and train_generative.py:61 under loss.backward()). */
int pcvae_latent_bwd(const float* dz, int64_t lddz, const float* eps, const float* mu, const float* lv, const float* pmu,
                     const float* plv, const float* dkld_dev, float dkld_host, float* dmu, float* dlv, float* dpmu,
                     float* dplv, int64_t B, int Z, pcvae_stream_t stream);

/* deterministic reduction: *out = scale * sum(x[0..n))   (CrossEntropyLoss 'mean', train_generative.py:59) */
int pcvae_sum(const float* x, int64_t n, float scale, float* out, pcvae_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K5  fused full-catalog softmax cross-entropy          models/pivotcvae.py:274 +
 *                                                       train_generative.py:36-42 (downsample), :59
 *     For every row r of rx[R,D] against the whole table E[N,D], WITHOUT materialising [R,N]:
 *        s_n   = <rx_r, E_n>
 *        keep_n= (n == target_r) | mask(r, n)                    mask: all ones (keep_prob >= 1),
 *                                                                explicit uint8 [R,N], or Philox
 *                                                                Bernoulli(keep_prob) keyed by
 *                                                                (seed, row_offset + r, n)
 *        z_n   = keep_n ? s_n : 0                                (masked-out logits are 0, not -inf)
 *        lse_r = log sum_n exp(z_n) ;  nll_r = lse_r - z_target
 *        dx_r  = sum_n keep_n * (softmax(z)_n - [n == target_r]) * E_n        (optional, [R,D])
 *     i.e. loss AND its gradient direction in one streaming pass (online softmax over catalog
 *     tiles, flash-style), so the backward pass is dx * (upstream / R).
 *     E is the fp32 table for F32, its bf16 copy for BF16, the c0 | c1 image of pcvae_split_bf16x2 for BF16X3, the c0 | c1 | c2
 *     image of pcvae_split_bf16x3 for BF16X6; E_lo (BF16X3 / BF16X6 only) is the fp32 table again: exact target logit / target
 *     row, and the exact f32 kernel for masked calls and for 256-row blocks whose norms rule out the max-free kernel.
 *     BF16X6 is the reference's arithmetic (fp32 operands, exact products, fp32 accumulate) on the bf16 pipe; BF16X3 and BF16 are
 *     narrower.  `ws` is scratch of at least pcvae_catalog_ws_bytes().
 *     e_max_norm: max_n ||E_n||_2 of the table (the model's table is row-normalised: 1.0).  The bf16
 *     path uses it to prove, per 256-row block, that exp2(logit) cannot leave the fp32 range and then
 *     skips the running-max machinery; pass <= 0 when unknown (always take the running-max kernel).
 * ------------------------------------------------------------------------------------------- */
size_t pcvae_catalog_ws_bytes(int64_t R, int64_t N, int D, int want_dx);
/* which kernel an unmasked training call (dx wanted) of this shape runs - for reports, never needed for correctness:
 * 0 = exact f32 MFMA kernel, 1 = bf16 kernel with two waves per SIMD, 2 = software-pipelined bf16 kernel (one wave per
 * SIMD), 3 = bf16x3 kernel, 4 = bf16x6 kernel, -1 = unsupported shape.  Mirrors the launch logic, including the PCVAE_PIPE_MIN_TILES override. */
int pcvae_catalog_ce_variant(int64_t R, int64_t N, int D, int prec);
int pcvae_catalog_ce(const float* rx, int64_t R, const void* E, const void* E_lo, int64_t N, int D, int prec,
                     float e_max_norm, const int64_t* target, float keep_prob, uint64_t seed, uint64_t row_offset,
                     const uint8_t* keep_mask, float* nll, float* lse, float* dx, void* ws, size_t ws_bytes,
                     pcvae_stream_t stream);

/* K6  fused catalog argmax (greedy decode)              models/cvae.py:97-101 ; models/pivotcvae.py:191
 *     idx[r] = first n maximising <x_r, E_n>  (torch.max tie rule: lowest index)                 */
int pcvae_catalog_argmax(const float* x, int64_t R, const void* E, const void* E_lo, int64_t N, int D, int prec,
                         float e_max_norm, int64_t* idx, float* best, void* ws, size_t ws_bytes, pcvae_stream_t stream);

/* K1 + K2 + torch.cat in one launch                    models/pivotcvae.py:250-258 (condition, docEmbed, userEmbed), :166,201,213,231
 *     (the concatenations), :194 (ground-truth pivot row): writes the item rows, the one-hot click count, the user row and the
 *     pivot row straight into the three stack inputs and slot 0 of rx:
 *       enc_in = [items (S D) | cond (S + 1) | user (D)], prior_in = [cond | user], scm_in = [z: untouched (Z) | cond | pivot | user],
 *       rx[:, :D] = pivot.  U = u = NULL: the no-user model (the user windows do not exist).                         */
int pcvae_assemble_inputs(const float* E, int64_t n_items, const float* U, int64_t n_users, const int64_t* s, const float* r,
                          const int64_t* u, int64_t B, int S, int D, int ncols, int Z, float* enc_in, int64_t ld_enc,
                          float* prior_in, int64_t ld_prior, float* scm_in, int64_t ld_scm, float* rx, int64_t ld_rx,
                          pcvae_stream_t stream);

/* K4 + K7 on PACKED head outputs y_enc = [mu | logvar], y_prior = [pmu | plogvar] ([B, 2 Z], leading dimension ld): what one
 * N = 2 Z GEMM per stack produces when the two heads' weights are adjacent.  Forward: z (window, ldz), eps, KL sum; backward: the
 * packed gradients g_enc = [dmu | dlogvar], g_prior = [dpmu | dplogvar].  models/cvae.py:79-83 + train_generative.py:61.       */
int pcvae_latent_fwd_packed(const float* y_enc, const float* y_prior, int64_t ld, const float* eps_in, uint64_t seed,
                            uint64_t offset, float* z, int64_t ldz, float* eps_out, float* kld_out, int64_t B, int Z,
                            pcvae_stream_t stream);
int pcvae_latent_bwd_packed(const float* dz, int64_t lddz, const float* eps, const float* y_enc, const float* y_prior,
                            int64_t ld, const float* dkld_dev, float dkld_host, float* g_enc, float* g_prior, int64_t ldg,
                            int64_t B, int Z, pcvae_stream_t stream);

/* downsample on a dense logits tensor                    train_generative.py:36-42
 *     out[r, n] = pred[r, n] if n == slate[r] or Bernoulli(keep_prob) else 0 (masked-out logits become 0, not -inf);
 *     the mask stream is the dense masked pcvae_catalog_ce's.  For callers that hold dense logits (small catalogs).   */
int pcvae_downsample_dense(const float* pred, int64_t ldp, const int64_t* slate, int64_t R, int64_t N, float keep_prob,
                           uint64_t seed, uint64_t row_offset, float* out, int64_t ldo, pcvae_stream_t stream);

/* K5, sparse form for n_neg << N                        train_generative.py:36-44,59 (downsample(pred, slates, 1000) + CE)
 *     Same result as pcvae_catalog_ce with keep_prob < 1, but only the KEPT items of a row are touched: masked-out logits are
 *     the constant 0 (exp(0) = 1 each in the denominator, no gradient), so the kernel enumerates the kept set of a row
 *     directly - geometric gap sampling, lane l of the row's wave walks catalog segment l, gaps floor(ln U / ln(1 - keep_prob))
 *     with U from Philox4x32-10 keyed by (seed, row_offset + r, lane, draw): i.i.d. Bernoulli(keep_prob) per item, the
 *     target always kept, independent of sharding - gathers those rows of the fp32 table and runs an online softmax over
 *     them plus the closed-form (N - n_kept) * exp(0) term.  Exact fp32; HBM-bound (~R * keep_prob * N * 4 D bytes).
 *     NOTE: this is a different Philox stream from pcvae_catalog_ce's per-item mask (same distribution).            */
int pcvae_catalog_ce_sparse(const float* rx, int64_t R, const float* E, int64_t N, int D, const int64_t* target,
                            float keep_prob, uint64_t seed, uint64_t row_offset, float* nll, float* lse, float* dx,
                            pcvae_stream_t stream);
/* the same two calls writing dx * dx_scale: the 1 / (rows * world_size) of CrossEntropyLoss's mean folded into the kernel, so that
 * the backward of the mean-reduced loss needs no scaling launch when the upstream gradient is 1 (train_generative.py:59,133)  */
int pcvae_catalog_ce_sparse_scaled(const float* rx, int64_t R, const void* E, int prec, int64_t N, int D, const int64_t* target,
                                   float keep_prob, uint64_t seed, uint64_t row_offset, float* nll, float* lse, float* dx,
                                   float dx_scale, const uint64_t* seed_dev, pcvae_stream_t stream);
int pcvae_catalog_ce_scaled(const float* rx, int64_t R, const void* E, const void* E_lo, int64_t N, int D, int prec,
                            float e_max_norm, const int64_t* target, float keep_prob, uint64_t seed, uint64_t row_offset,
                            const uint8_t* keep_mask, float* nll, float* lse, float* dx, float dx_scale, void* ws,
                            size_t ws_bytes, pcvae_stream_t stream);

/* K10 sampled pivot                                     models/pivotcvae.py:349-351, 371-373
 *     idx[r] ~ Categorical(sigmoid(<x_r, E_n>)) over the whole catalog.  Drawn by rejection sampling - propose n uniform in [0, N),
 *     accept with probability sigmoid(s_n) <= 1: exactly that categorical, for ~1 / mean sigmoid dot products per row instead of the
 *     [R, N] score matrix the reference hands to torch.multinomial.  Proposal k of row r = Philox(seed, row_offset + r, k, "RJCT"):
 *     n_k = (x << 32 | y) mod N, u_k = (z + 0.5) 2^-32; idx[r] = n_k of the lowest k with u_k < sigmoid(s) (s = exact fp32 fmaf dot
 *     product of the fp32 table row: E = [N, D] fp32, prec = PCVAE_PREC_F32, E_lo unused).  A row that rejects 512 proposals is
 *     drawn by Gumbel-max over the whole catalog instead (argmax_n log sigmoid(s_n) - log(-log u_n), u_n = Philox(seed, row, n)).
 *     The reference's torch.multinomial stream cannot be matched: parity is distributional (tests) or by feeding the recorded
 *     draw back in on the host side.                                                                                          */
int pcvae_catalog_sample(const float* x, int64_t R, const void* E, const void* E_lo, int64_t N, int D, int prec,
                         uint64_t seed, uint64_t row_offset, int64_t* idx, void* ws, size_t ws_bytes,
                         pcvae_stream_t stream);
/* the same at stream position row_offset + *row_offset_dev (row_offset_dev NULL: as above) */
int pcvae_catalog_sample_at(const float* x, int64_t R, const void* E, const void* E_lo, int64_t N, int D, int prec,
                            uint64_t seed, uint64_t row_offset, const uint64_t* row_offset_dev, int64_t* idx, void* ws,
                            size_t ws_bytes, pcvae_stream_t stream);

/* fp32 table -> bf16 hi (round-to-nearest-even) and optional bf16 lo (residual) copies */
int pcvae_split_bf16(const float* src, int64_t n, uint16_t* hi, uint16_t* lo, pcvae_stream_t stream);

/* fp32 table [N, D] -> the bf16x3 catalog kernel's table image, hi = RNE bf16(E), lo = RNE bf16(E - hi) (hi + lo carries 16
 * mantissa bits of E).  D <= 128: [N, 2 D] bf16, row n = hi(E_n) | lo(E_n).  D = 256: two such images of 128 dims back to back,
 * [2][N][256]: image i = hi | lo of dims 128 i .. 128 i + 127 (N * 2 D elements either way)                                  */
int pcvae_split_bf16x2(const float* src, int64_t N, int D, uint16_t* out, pcvae_stream_t stream);
/* the bf16x6 table image: [N, 3 D] bf16 (D <= 128), row n = c0(E_n) | c1(E_n) | c2(E_n) with c0 = RNE bf16(E), c1 = RNE bf16(E - c0),
 * c2 = RNE bf16(E - c0 - c1): c0 + c1 + c2 == E exactly for every normal fp32 value.  (models/pivotcvae.py:274: the table operand) */
int pcvae_split_bf16x3(const float* src, int64_t N, int D, uint16_t* out, pcvae_stream_t stream);

/* a13  simulator click models as in-loop evaluators     env/response_model.py:129-150 (URM), 286-295 (URM_P), 315-323 (URM_P_MR)
 *     out[b, s] = sigmoid(<E[slate]/||E[slate]||, U[user]> + item_bias[slate] + user_bias[user])
 *                 (+ sum_d U[user, d] * pos_dep[d * S + s] + pos_bias[s]      when pos_bias / pos_dep are given: URM_P)
 *                 (+ mr_factor * <E[slate]/||.||, sigmoid(mean_s E[slate_s]/||.||)>   when use_mr: URM_P_MR)
 *     U is the RAW user table (the reference overwrites the normalised lookup, :141-142); pos_dep is the [S, D] buffer read
 *     as [D, S] (`.view(featureSize, slateSize)`, :292).  D <= 256.                                              */
int pcvae_urm_forward(const float* E, const float* item_bias, int64_t n_items, const float* U, const float* user_bias,
                      int64_t n_users, const int64_t* slates, const int64_t* users, const float* pos_bias,
                      const float* pos_dep, float mr_factor, int use_mr, int64_t B, int S, int D, float* out,
                      pcvae_stream_t stream);

/* (f)3 candidate sets for the sampled-softmax path      data_loader.py:46-58
 *     cand[r, :] = Cn uniform ids in [0, n_items) (Philox keyed by (seed, row_offset + r, column); or the recorded draw `raw`);
 *     if feature[r] is among them tgt[r] = the first such column, else cand[r, 0] = feature[r] and tgt[r] = 0.   */
int pcvae_candidate_draw(const int64_t* feature, int64_t R, int64_t n_items, int Cn, uint64_t seed, uint64_t row_offset,
                         const int64_t* raw, int64_t* cand, int64_t* tgt, pcvae_stream_t stream);

/* K9  candidate-set scores                              models/pivotcvae.py:265-271
 *     p[r, c] = <E[cand[r, c]], rx_r> ;  bwd: drx_r = sum_c dp[r, c] * E[cand[r, c]]            */
int pcvae_candidate_scores(const float* rx, int64_t R, const float* E, int64_t N, int D, const int64_t* cand,
                           int Cn, float* p, pcvae_stream_t stream);
int pcvae_candidate_scores_bwd(const float* dp, int64_t R, const float* E, int64_t N, int D, const int64_t* cand,
                               int Cn, float* drx, pcvae_stream_t stream);

/* dense softmax cross-entropy over a small class axis (candidate sets, Cn <= a few thousand)
 *                                                       train_generative.py:56 (lossFun(pred, sampleTargets))
 *     nll[r] = logsumexp_c p[r, c] - p[r, target[r]] ;  dp[r, c] = softmax(p[r, :])_c - [c == target[r]]  (optional) */
int pcvae_dense_ce(const float* p, int64_t ldp, int64_t R, int C, const int64_t* target, float* nll, float* dp,
                   int64_t lddp, pcvae_stream_t stream);

/* K9 fused: candidate-set softmax cross-entropy, loss AND gradient in one launch, nothing of size [R, Cn] in memory
 *                                                       data_loader.py:46-58 ; models/pivotcvae.py:265-271 ;
 *                                                       train_generative.py:52-57 (the DEFAULT mode: no --mask_train)
 *     ids of row r: given (cand [R, Cn] + cand_target [R]: what a batch of the reference's dataset carries, or a recorded draw)
 *                   or, with cand == NULL, drawn in-kernel exactly as pcvae_candidate_draw(feature, .., seed, row_offset) draws them
 *                   (first-hit / overwrite rule; the stream is keyed by the GLOBAL row row_offset + r);
 *     s_c = <rx_r, E[id_c]> (fp32 fmaf chains) ;  nll[r] = logsumexp_c s_c - s_t ;  lse[r] optional ;
 *     dx[r, :] = dx_scale * (sum_c softmax_c E[id_c] - E[id_t])  (optional) ;  tgt_out[r] = the target column used (optional).
 *     A candidate id or target outside its range makes that row's outputs NaN (the reference raises an index error).
 *     D in {16, 32, 64, 128, 256} (other widths: zero-padded by the caller), N < 2^31 - 1.
 *     prec = PCVAE_PREC_F32: E = the fp32 table (the reference's arithmetic).  prec = PCVAE_PREC_BF16 (here and in
 *     pcvae_catalog_ce_sparse_scaled): E = the bf16 copy of pcvae_split_bf16 - half the gathered bytes; rows are widened exactly,
 *     products and sums stay fp32 (the stated arithmetic of configs 3 / 5; tolerances of the bf16 catalog kernels).           */
int pcvae_candidate_ce(const float* rx, int64_t R, const void* E, int prec, int64_t N, int D, int Cn, const int64_t* feature,
                       uint64_t seed, uint64_t row_offset, const int64_t* cand, const int64_t* cand_target, float* nll,
                       float* lse, float* dx, float dx_scale, int64_t* tgt_out, const uint64_t* seed_dev, int64_t n_items,
                       pcvae_stream_t stream);
/*     n_items (ABI 3): the in-kernel draw is uniform over [0, n_items), n_items <= N - the DATASET's id range, data_loader.py:23
 *     (self.max_iid = np.max(slates)) and :46 (np.random.randint(self.max_iid + 1, ...)); a table can have more rows than the
 *     slates use (the simulators build n_item + 1 rows, env/response_model.py:30).  <= 0: the table's row count N.  Given sets
 *     (cand != NULL) are bounded by N as before.                                                                              */
/*     seed_dev (here, in pcvae_catalog_ce_sparse_scaled; row_offset_dev in pcvae_catalog_sample_at): NULL, or a device word the
 *     kernel reads its seed (stream position) from at run time instead of the by-value argument - kernel arguments are frozen when
 *     a step is captured into a hipGraph, a device word is not: pcvae_set_words writes it before each replay.                      */
int pcvae_set_words(uint64_t* dst, uint64_t a, uint64_t b, pcvae_stream_t stream);   /* dst[0] = a, dst[1] = b, on the stream */

/* ---------------------------------------------------------------------------------------------
 * K8  Adam over one flat fp32 buffer                    train_generative.py:103,134
 *     torch.optim.Adam defaults (betas 0.9/0.999, eps 1e-8, no weight decay, bias-corrected):
 *        g' = g * grad_scale ; m += (g' - m)(1 - b1) ; v = b2 v + (1 - b2) g'^2
 *        p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
 * ------------------------------------------------------------------------------------------- */
int pcvae_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                    float eps, int step, float grad_scale, pcvae_stream_t stream);
/* the same with torch.optim.Adam's weight_decay (L2 term added to the gradient: g' = g * grad_scale + weight_decay * p);
 * pretrain_env.py:59 trains the click model with it                                                                */
int pcvae_adam_step_l2(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                       float eps, int step, float grad_scale, float weight_decay, pcvae_stream_t stream);

/* Measurement only (bench.py, tools/): per-kernel durations from HIP events attached to the dispatch itself.  While enabled, the
 * instrumented launches (tags below) go out through hipExtLaunchKernelGGL with a start / stop event pair; read returns the number
 * of timed launches and fills their durations (ms) and tags in launch order.  No reference counterpart (the reference times with
 * time.time(), train_generative.py:113).
 * The switch and the list are PER THREAD (round 6; they were process-wide): a thread times, and reads back, its own instrumented
 * launches only - off by default.                                                                                               */
#define PCVAE_TIMER_GATHER 1        /* gather_rows_coal_kernel (D = 64 / 128 / 256) or gather_rows_vec4_kernel (pcvae_gather_rows) */
#define PCVAE_TIMER_ASSEMBLE 2      /* assemble_inputs_vec_kernel   (pcvae_assemble_inputs)  */
#define PCVAE_TIMER_CANDIDATE_CE 3  /* candidate_ce_kernel          (pcvae_candidate_ce)     */
int pcvae_kernel_timer(int enable);
int pcvae_kernel_timer_read(float* ms, int* tags, int cap);

/* optimizer.zero_grad() (train_generative.py:124): the flat gradient buffer (+ its statistics tail) zeroed by ONE fill kernel on the
 * stream.  Deliberately not hipMemsetAsync: as a node of a captured hipGraph the memset is not ordered against its neighbours on
 * ROCm 7.2 (replayed training drifted off the eager trajectory; csrc/elementwise.hip).  Any pointer / byte count.              */
int pcvae_zero(void* p, size_t nbytes, pcvae_stream_t stream);

/* the logged ELBO terms of a step (train_generative.py:62-63, :128): out[0..2] = (rec + beta * kld, rec, kld).  A data-parallel
 * rank writes the record into the tail of its gradient buffer: ONE all-reduce sums gradients and statistics.                  */
int pcvae_elbo_pack(const float* rec, const float* kld, float beta, float* out, pcvae_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Training the click model                       pretrain_env.py:25-139 ; env/response_model.py:76-87
 *   scatter_add_rows    : nn.Embedding backward, dtable[idx[i], :] += g_row(i)  (g laid out like gather_rows' output)
 *   normalize_rows_norm : F.normalize in place, row norms max(||x||, 1e-12) kept;  normalize_rows_bwd : its backward
 *   bce_sigmoid         : nn.BCELoss()(sigmoid(x), t) per element (logs clamped at -100) and dL/dx * grad_scale
 *   relu_bwd            : g *= (y > 0)   (F.relu backward keyed on the activated output)
 * ------------------------------------------------------------------------------------------- */
int pcvae_scatter_add_rows(const float* g, int64_t g_ld, int group, int D, const int64_t* idx, int64_t n_idx,
                           float* dtable, int64_t n_rows, pcvae_stream_t stream);
int pcvae_normalize_rows_norm(float* x, int64_t ldx, int64_t rows, int cols, float* norm, pcvae_stream_t stream);
int pcvae_normalize_rows_bwd(const float* y, int64_t ldy, const float* norm, const float* g, int64_t ldg, float* dx,
                             int64_t lddx, int64_t rows, int cols, pcvae_stream_t stream);
int pcvae_bce_sigmoid(const float* x, const float* t, int64_t n, float* loss, float* dx, float grad_scale,
                      pcvae_stream_t stream);
int pcvae_relu_bwd(float* g, int64_t ldg, const float* y, int64_t ldy, int64_t rows, int cols, pcvae_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Offline metrics of generated slates                          analysis.py:5-30
 *   coverage_count : number of distinct item ids among ids[0..n) (ids outside [0, N) ignored); bits = scratch of
 *                    ceil(N / 32) words.  get_coverage = count / N.
 *   ils            : out[b] = (sum_{i,j} <e_i, e_j> - S) / (S (S - 1)) with e_i = normalize(E[slates[b, i]])  (get_ILS)
 * ------------------------------------------------------------------------------------------- */
int pcvae_coverage_count(const int64_t* ids, int64_t n, int64_t N, unsigned int* bits, int64_t* count,
                         pcvae_stream_t stream);
int pcvae_ils(const float* E, int64_t N, int D, const int64_t* slates, int64_t B, int S, float* out,
              pcvae_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PCVAE_H */
