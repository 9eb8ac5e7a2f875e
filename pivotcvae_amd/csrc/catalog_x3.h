// K5 in fp32-EQUIVALENT arithmetic on the bf16 matrix cores ("bf16x3"): the fused full-catalog softmax cross-entropy of
// catalog_bf16.hip with BOTH operands of both contractions split into bf16 hi + lo halves and three bf16 MFMAs per product
// (hi*hi + hi*lo + lo*hi, fp32 accumulate; the dropped lo*lo term is 2^-18 relative).  Replaces, at reference precision,
// `mm` + `downsample(n_neg = N)` + `CrossEntropyLoss` + their backward (models/pivotcvae.py:274, train_generative.py:59).
//
// This file is included by catalog_bf16.hip INSIDE its anonymous namespace (it reuses that file's LDS geometry, LDS-DMA staging
// and inline-asm MFMA / LDS-read helpers); it is not a translation unit of its own.
//
// Table image (pcvae_split_bf16x2): row n = hi(E_n)[D] | lo(E_n)[D] bf16 = 4 D bytes.  For D = 128 that is exactly the LDS geometry
// of the D = 256 bf16 kernel (FastGeo<256>: 512-byte rows, the 16-chunk XOR swizzle inside each 256-byte half, one 32-item
// subtile per 16 KB ring chunk), so the ring, the seams and both conflict-free read patterns are shared with it:
//   logits    k-steps 0..3 read the hi half, 4..7 the lo half of a row.  Step (s, rt):
//               s < 4 : acc[rt][ct] += A_hi(s) . xh[ct][s]   and   += A_hi(s) . xl[ct][s]
//               s >= 4: acc[rt][ct] += A_lo(s) . xh[ct][s-4]
//   numerators p = exp2(acc) (fp32) -> ph = RNE bf16(p), pl = RNE bf16(p - ph) (p - ph is exact in fp32)
//   row sums  lsum[ct] += ones . ph[ct]  and  += ones . pl[ct]    (the same 16-bit numerators the gradient chain multiplies)
//   gradient  d tiles 0..7 are E_hi^T, 8..15 E_lo^T:
//               DT < 8 : U[DT][ct] += T_hi(DT) . ph[ct]  and  += T_hi(DT) . pl[ct]
//               DT >= 8: U[DT-8][ct] += T_lo(DT) . ph[ct]
// 100 MFMAs per 32-item subtile and 32-row wave (2 x 24 CT + 4 row sums at CT = 2) against the bf16 kernel's 36: ~2.8x its
// time, ~4x faster than the exact f32-MFMA kernel, at the f32 kernel's tolerances (tests/test_hip_x3.py).
//
// Schedule (one wave per SIMD, CT = 2 column tiles of 16 rows per wave, every MFMA / LDS read / VALU op an inline-asm statement
// in schedule order - see catalog_ce_bf16_pipe_kernel for the method and for what hipcc may not place in these loops):
//   slot t:  L(t)    logits chain of subtile t (48 MFMAs)  ||  hi / lo split of subtile t-1's numerators (6 cheap VALU ops per pair)
//            G(t-1)  row sums + gradient chain of subtile t-1 (52 MFMAs)  ||  the exponentials of subtile t, the seam (counted
//                    vmcnt + s_barrier + refill of the ring) and the first A fragments of L(t+1) in the middle of the chain.
// The slot is bound by VECTOR ISSUE, not by the MFMA pipe alone: an MFMA holds the issue port for 8 of its 16 cycles, a v_exp_f32
// for 8, a conversion / LDS read / s_waitcnt for 4-5, and a gap runs max(16, the sum) (MI355X_MICROARCH.md, issue-cost row).  So
// the statement order gives every MFMA gap at most 8 cycles of other work: a step is  wait | MFMA | LDS reads of a later step |
// MFMA | op | MFMA | op | MFMA ..., the 16 exponentials sit one per gap in the two middle gaps of the eight hi tiles of G, the 48
// split ops two per gap in the hi steps of L (the uniform spread this replaced put an exponential, two reads and a wait into one
// gap: 0.75 -> see DESIGN.md for the measured MFMA-pipe utilisation).
// A single accumulation chain of v_mfma_f32_16x16x32_bf16 issues back to back at full rate (MI355X_MICROARCH.md, cycle
// constants), so the two or three MFMAs that update one accumulator need no interleaving.
// Max-free like the bf16 fast kernels (row blocks whose Cauchy-Schwarz logit bound exceeds 90 are flagged by
// catalog_row_bound_kernel and run the exact f32 kernel instead: catalog_ce_x3 below).
#pragma once

#ifndef X3_AD
#define X3_AD 3      // logits chain: A fragments requested ahead (the first X3_AD of a slot are issued at the previous seam)
#endif
#ifndef X3_TD
#define X3_TD 3      // gradient chain: d tiles requested ahead (two transposed reads each)
#endif
// Timing probes (tools/build_variant.sh ... -DX3_PROBE=<mask>; results are garbage, only the time means something): drop from the
// steady-state slots 1: the numerator VALU ops, 2: the A-fragment reads, 4: the transposed reads, 8: the seam (wait + barrier +
// refill), 16: the LDS waits, 32: only the seam's barrier, 64: only the seam's refill
#ifndef X3_PROBE
#define X3_PROBE 0
#endif

// D = 256 (round 3): the table is TWO such images - dims 0..127 and dims 128..255, each [N, 256] bf16 hi | lo with the 512-byte rows
// of the D = 128 kernel - and a slot walks both: the logits chain is 16 k-steps per image into the SAME accumulators, the gradient
// chain 16 transposed tiles per image into U tiles 0..7 / 8..15, with one seam (wait + barrier + refill of ONE 16 KB chunk) in the
// middle of each image's tiles.  Ring chunk c = (subtile c / 2, image c % 2); 8 ring buffers, so image 1 of a subtile always sits
// 16 KB behind image 0 (no wrap between them).  The VALU work of a slot (16 exponentials, 48 split ops per 32-row wave) is what
// it was, spread over twice the MFMAs: 196 MFMAs per 32-item subtile.
template <int D, int CT>
struct X3Geo {
    static_assert(D == 128 || D == 256, "bf16x3 is built for D = 128 (one image) and D = 256 (two images of 128 dims)");
    static constexpr int NIMG = D / 128;                   // images of 128 dims
    static constexpr int DL = 256;                         // row width of ONE image in bf16 elements: hi | lo
    using GL = FastGeo<DL>;
    static_assert(GL::SUB == 1 && GL::KS == 8 && GL::NDT == 16, "layout of FastGeo<256>: a 512-byte row, one 32-item subtile per 16 KB chunk");
    static constexpr int CB = 16384;                       // ring chunk = one 32-item subtile of one image
    static constexpr int KSH = 4;                          // k-steps per half (hi / lo) of an image
    static constexpr int NIL = 2 * GL::KS;                 // logits steps per image: (k-step, row tile)
    static constexpr int NI = NIMG * NIL;                  // steps of the logits chain
    static constexpr int NDTI = GL::NDT;                   // transposed tiles per image: 8 hi then 8 lo
    static constexpr int NDTL = NIMG * NDTI;
    static constexpr int NDT = D / 16;                     // 16-wide d tiles of U
    static constexpr int MLI = 2 * KSH * 2 * CT + 2 * KSH * CT;         // MFMAs of L per image
    static constexpr int ML = NIMG * MLI;
    static constexpr int MGI = 8 * 2 * CT + 8 * CT;                     // MFMAs of G per image
    static constexpr int MG = 2 * CT + NIMG * MGI;                      // (row sums first)
    static constexpr int P = 4 * CT;                       // numerator pairs per slot: (row tile, column tile, half)
    static constexpr int GOPS = 2 * P;                     // during G: the 2 exponentials of every pair
    static constexpr int LOPS = 6 * P;                     // during the next L: hi conversion, shift, mask, 2 subtractions, lo conversion
    static constexpr int ROWS = 4 * 16 * CT;               // rows per workgroup (4 waves)
    static constexpr int PF = 3;                           // chunks requested ahead
    static constexpr int NB = NIMG == 1 ? 6 : 8;           // ring buffers (>= PF + 3 NIMG - 1 live or in flight; a multiple of NIMG)
    static constexpr int TR = NB / NIMG;                   // slots per steady-state trip (every LDS offset an immediate)
    // MFMA positions of a phase <-> (step, j-th MFMA of the step).  Statement order inside a step:
    //     wait(this step's LDS data) | MFMA 0 | LDS reads of a later step | MFMA 1 | ops | ... | MFMA last | ops
    // ---- G: row sums (2 CT), then per image hi tiles (2 CT MFMAs each), lo tiles (CT each).  One exponential behind the MIDDLE
    // MFMAs of a hi tile: the gap behind MFMA 0 carries the two transposed reads (8 cycles), the one behind the last MFMA the next
    // step's wait
    static constexpr int gcap(int m) {
        if (m < 2 * CT || m >= MG) return 0;
        const int mm = (m - 2 * CT) % MGI;
        if (mm >= 8 * 2 * CT) return 0;
        const int j = mm % (2 * CT);
        // CT = 1: a hi tile is two MFMAs, the one exponential shares the gap of the transposed reads (not behind the very first
        // tile: the logits accumulators are fresh there)
        if (CT == 1) return (j == 0 && m > 2 * CT) ? 1 : 0;
        return (j == 0 || j == 2 * CT - 1) ? 0 : 1;
    }
    static constexpr int gfirst(int m) {
        int n = 0;
        for (int i = 0; i < m && i < MG; ++i) n += gcap(i);
        return n > GOPS ? GOPS : n;
    }
    // ---- L: per image hi steps (2 CT MFMAs), lo steps (CT).  Cheap ops (4-5 cycles): one next to the A-fragment read behind MFMA 0,
    // one in front of the next step's wait behind the last MFMA, two in the gaps between; none in the last two gaps of L (the packed
    // numerators are MFMA operands right after it)
    static constexpr int lcap(int m) {
        // nothing behind the MFMAs of step 0: the first split op overwrites the hi numerators (r.wh) that the LAST MFMAs of the
        // gradient chain in front of this L read as their B operand.  hipcc sees those operands dead and the in-order issue would
        // seem to protect them, but a VALU write two MFMAs behind such a read corrupted it (column tile 0 - the first pair
        // written - NaN, depending on where an unrelated ds_read sat: tools/dbg_x3.py): MFMAs queue in front of the
        // matrix pipe and read their operands when they start, not when they issue.  A whole step (2 CT MFMAs) of distance.
        if (m < 2 * CT || m >= ML - 2) return 0;
        const int mm = m % MLI, nh = 2 * KSH * 2 * CT;
        const int len = mm < nh ? 2 * CT : CT, j = mm < nh ? mm % (2 * CT) : (mm - nh) % CT;
        return (j == 0 || j == len - 1) ? 1 : 2;
    }
    static constexpr int lfirst(int m) {
        int n = 0;
        for (int i = 0; i < m && i < ML; ++i) n += lcap(i);
        return n > LOPS ? LOPS : n;
    }
};

template <int CT, int NT = 16>
struct X3Regs {
    f32x4 acc[2][CT];          // logits of the current subtile [row tile][column tile] (log2 domain)
    unsigned wh[CT][4];        // [ct][2 rt + h]: bf16 pair of hi halves; written during L, read by the G right behind it
    unsigned wl[CT][4];        // lo halves, likewise
    float e[4 * CT][2];        // the fp32 numerators: written during G (exponentials), split during the next L
    float tmp[2][2];
    // MFMA operands that must outlive their last MFMA (see x3_keep): the packed numerators and the transposed tiles of the
    // gradient chain, held here so that the NEXT slot's logits chain can still name them
    bf16x8 pbh[CT], pbl[CT];
    s16x4 tl[NT], th[NT];      // (X3Geo::NDTL transposed tiles)
};

// G-phase op V: exponential `V & 1` of pair V / 2 (pair k <-> row tile k / (2 CT), column tile (k / 2) % CT, half k & 1)
template <int CT, int V, class RG>
__device__ __forceinline__ void x3_gop(RG& r) {
    constexpr int k = V / 2, which = V % 2;
    constexpr int rt = k / (2 * CT), ct = (k / 2) % CT, h = k & 1;
    asm volatile("v_exp_f32 %0, %1" : "=v"(r.e[k][which]) : "v"(r.acc[rt][ct][2 * h + which]));
}
// L-phase op V: the split of the numerators into bf16 hi + lo halves, two pairs (a, b) interleaved so that no op reads the result
// of the op right in front of it:  hi = RNE bf16(e);  lo = RNE bf16(e - float(hi))  (the difference is exact in fp32)
//   block of 12 ops: cvt a, cvt b, shl a, shl b, and a, and b, sub0 a, sub0 b, sub1 a, sub1 b, cvt-lo a, cvt-lo b
template <int CT, int V, class RG>
__device__ __forceinline__ void x3_lop(RG& r) {
    constexpr int blk = V / 12, j = (V % 12) / 2, ab = V & 1, k = 2 * blk + ab;
    constexpr int rt = k / (2 * CT), ct = (k / 2) % CT, h = k & 1;
    if constexpr (j == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r.wh[ct][2 * rt + h]) : "v"(r.e[k][0]), "v"(r.e[k][1]));
    else if constexpr (j == 1) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(r.tmp[ab][0]) : "v"(r.wh[ct][2 * rt + h]));
    else if constexpr (j == 2) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(r.tmp[ab][1]) : "v"(r.wh[ct][2 * rt + h]));
    else if constexpr (j == 3) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r.tmp[ab][0]) : "v"(r.e[k][0]));
    else if constexpr (j == 4) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r.tmp[ab][1]) : "v"(r.e[k][1]));
    else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r.wl[ct][2 * rt + h]) : "v"(r.tmp[ab][0]), "v"(r.tmp[ab][1]));
}
template <int D, int CT, int M, int V = X3Geo<D, CT>::gfirst(M), class RG>
__device__ __forceinline__ void x3_gops(RG& r) {
    using XG = X3Geo<D, CT>;
    if constexpr (M < XG::MG && V < XG::gfirst(M + 1)) {
        x3_gop<CT, V>(r);
        x3_gops<D, CT, M, V + 1>(r);
    }
}
template <int D, int CT, int M, int V = X3Geo<D, CT>::lfirst(M), class RG>
__device__ __forceinline__ void x3_lops(RG& r) {
    using XG = X3Geo<D, CT>;
    if constexpr (M < XG::ML && V < XG::lfirst(M + 1)) {
        x3_lop<CT, V>(r);
        x3_lops<D, CT, M, V + 1>(r);
    }
}
// all ops of one phase back to back (fill slot, drain, fenced slots: no MFMAs to hide them under)
template <int D, int CT, int V = 0, class RG>
__device__ __forceinline__ void x3_all_gops(RG& r) {
    if constexpr (V < X3Geo<D, CT>::GOPS) {
        x3_gop<CT, V>(r);
        x3_all_gops<D, CT, V + 1>(r);
    }
}
template <int D, int CT, int V = 0, class RG>
__device__ __forceinline__ void x3_all_lops(RG& r) {
    if constexpr (V < X3Geo<D, CT>::LOPS) {
        x3_lop<CT, V>(r);
        x3_all_lops<D, CT, V + 1>(r);
    }
}

// acc (VGPR) += A . B with the B operand in AGPRs: the lo fragments of rx live there (MFMA operands may be ArchVGPRs or AccVGPRs on
// gfx90a and later), which is what makes 64 rows per wave (CT = 4) fit the 256 architectural VGPRs an asm operand can live in
template <bool COLD>
__device__ __forceinline__ void mfma_v_ab(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    if constexpr (COLD) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 15" : "+v"(acc) : "v"(a), "a"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
}

template <int CT>
__device__ __forceinline__ void x3_pack(const unsigned (&w)[CT][4], bf16x8 (&pb)[CT]) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const u32x4 v = {w[ct][0], w[ct][1], w[ct][2], w[ct][3]};
        pb[ct] = __builtin_bit_cast(bf16x8, v);
    }
}

// ---- operand lifetimes hipcc cannot know about.  The SIMD issues an MFMA every 8 cycles but the matrix pipe starts one every 16:
// in MFMA-dense stretches (the lo tiles at the end of G: two MFMAs per step and little else) MFMAs queue up in front of the pipe,
// and an MFMA reads its A / B operands when it STARTS, not when it issues.  hipcc sees an operand dead right behind its last MFMA
// and hands the register to the next asm result - e.g. the destination of a ds_read issued two MFMAs later, whose data then lands
// (LDS latency ~64+ cycles) BEFORE the queued MFMA has read the old value.  Observed: the first A-fragment request of L(t+1)
// was given the registers of pbh[0], the B operand of the third-last MFMA of G: column tile 0 NaN (tools/dbg_x3.py).  An empty
// asm use keeps an operand reserved for at least a whole step (>= 2 CT MFMAs) behind its last MFMA: fragments and tiles two
// steps, the operands of a chain's tail until the next chain's second step.
template <int CT>
__device__ __forceinline__ void x3_keep_pb(const bf16x8 (&pbh)[CT], const bf16x8 (&pbl)[CT]) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) asm volatile("" ::"v"(pbh[ct]), "v"(pbl[ct]));
}
__device__ __forceinline__ void x3_keep(const bf16x8& a) { asm volatile("" ::"v"(a)); }
__device__ __forceinline__ void x3_keep(const s16x4& a, const s16x4& b) { asm volatile("" ::"v"(a), "v"(b)); }

// first MFMA position of logits step I: per image, steps 0 .. 2 KSH - 1 are hi steps (2 CT MFMAs), the rest lo steps (CT MFMAs)
template <int D, int CT>
__host__ __device__ constexpr int x3_lpos(int I) {
    using XG = X3Geo<D, CT>;
    constexpr int NH = 2 * XG::KSH;
    const int img = I / XG::NIL, i = I % XG::NIL;
    return img * XG::MLI + (i < NH ? i * 2 * CT : NH * 2 * CT + (i - NH) * CT);
}

// A fragment of logits step I (any image): image I / NIL sits IMG_STRIDE bytes behind image 0 of the same subtile
template <int D, int CT, int OFF, int I>
__device__ __forceinline__ void x3_a_issue(const unsigned lbase, const int a0, bf16x8& a) {
    using XG = X3Geo<D, CT>;
    pipe_a_issue<XG::DL, OFF + (I / XG::NIL) * XG::CB, I % XG::NIL>(lbase, a0, a);
}
template <int D, int CT, int OFF, int K>
__device__ __forceinline__ void x3_a_prologue(const unsigned lbase, const int a0, bf16x8 (&af)[X3Geo<D, CT>::NI]) {
    if constexpr (K > 0) {
        x3_a_prologue<D, CT, OFF, K - 1>(lbase, a0, af);
        x3_a_issue<D, CT, OFF, K - 1>(lbase, a0, af[K - 1]);
    }
}
// transposed tile DT (any image)
template <int D, int CT, int OFF, int DT>
__device__ __forceinline__ void x3_tr_issue(const unsigned lbase, const int t0, s16x4& lo, s16x4& hi) {
    using XG = X3Geo<D, CT>;
    tr_issue<XG::DL, OFF + (DT / XG::NDTI) * XG::CB, DT % XG::NDTI>(lbase, t0, lo, hi);
}
template <int D, int CT, int OFF, int K>
__device__ __forceinline__ void x3_tr_prologue(const unsigned lbase, const int t0, s16x4 (&tl)[X3Geo<D, CT>::NDTL],
                                               s16x4 (&th)[X3Geo<D, CT>::NDTL]) {
    if constexpr (K > 0) {
        x3_tr_prologue<D, CT, OFF, K - 1>(lbase, t0, tl, th);
        x3_tr_issue<D, CT, OFF, K - 1>(lbase, t0, tl[K - 1], th[K - 1]);
    }
}

// L(t): logits chain into r.acc with the hi / lo split of the PREVIOUS subtile's numerators (r.e -> r.wh, r.wl) in its gaps.
// A fragment I + X3_AD is requested behind the first MFMA of step I, so X3_AD - 1 younger fragments are in flight at step I's wait.
// xh[ct][img KSH + s] / xl[ct][...]: rx fragments of image img; the hi fragments live in VGPRs, the lo fragments in AGPRs (MFMA B
// operands may): D = 256 then pins 200 AGPRs (U 128, row sums 8, xl 64) and its steady-state loop ~190 VGPRs.
template <int D, int CT, int OFF, int I, bool HAS_PREV, bool COLD, class RG>
__device__ __forceinline__ void x3_logits(const unsigned lbase, const int a0, bf16x8 (&af)[X3Geo<D, CT>::NI],
                                          const bf16x8 (&xh)[CT][X3Geo<D, CT>::NIMG * X3Geo<D, CT>::KSH],
                                          const bf16x8 (&xl)[CT][X3Geo<D, CT>::NIMG * X3Geo<D, CT>::KSH], RG& r) {
    using XG = X3Geo<D, CT>;
    if constexpr (I < XG::NI) {
        if constexpr (COLD || !(X3_PROBE & (2 | 16))) lgkm_wait<(I + X3_AD - 1 < XG::NI ? X3_AD - 1 : XG::NI - 1 - I)>();
        constexpr int img = I / XG::NIL, s = (I % XG::NIL) >> 1, rt = I & 1;
        constexpr int xs = img * XG::KSH + s % XG::KSH;       // rx fragment of this k-step
        constexpr int M0 = x3_lpos<D, CT>(I);
        // operand reservations (x3_keep): an operand stays allocated until >= 4 MFMAs behind its last MFMA - two steps at CT >= 2
        // (every step has >= 2 MFMAs; the statement sits behind the step's second MFMA), four at CT = 1 (a lo step is ONE MFMA)
        constexpr int KD = CT == 1 ? 4 : 2, KPOS = CT == 1 ? 0 : 1;
#define PCVAE_X3_TAIL(POS)                                                                                 \
            if constexpr ((POS) == 0 && I + X3_AD < XG::NI && (COLD || !(X3_PROBE & 2)))                   \
                x3_a_issue<D, CT, OFF, I + X3_AD>(lbase, a0, af[I + X3_AD]);                               \
            if constexpr ((POS) == KPOS && I >= KD) x3_keep(af[I - KD]);                                   \
            if constexpr ((POS) == KPOS && I == 2 && HAS_PREV) {   /* the tail operands of the gradient chain in front of this L */ \
                x3_keep_pb<CT>(r.pbh, r.pbl);                                                              \
                x3_keep(r.tl[XG::NDTL - 1], r.th[XG::NDTL - 1]);                                           \
                x3_keep(r.tl[XG::NDTL - 2], r.th[XG::NDTL - 2]);                                           \
                x3_keep(r.tl[XG::NDTL - 3], r.th[XG::NDTL - 3]);                                           \
                if constexpr (KD > 3) x3_keep(r.tl[XG::NDTL - 4], r.th[XG::NDTL - 4]);                     \
            }                                                                                              \
            if constexpr (HAS_PREV && !COLD && !(X3_PROBE & 1)) x3_lops<D, CT, M0 + (POS)>(r);
#define PCVAE_X3_L(POS, INIT, XB)   /* B operand in VGPRs (image 0's hi fragments) */                      \
        {                                                                                                  \
            constexpr int cti_ = (POS) % CT;                                                               \
            if constexpr (INIT) mfma_v0<COLD>(r.acc[rt][cti_], af[I], XB[cti_][xs]);                       \
            else mfma_v<COLD>(r.acc[rt][cti_], af[I], XB[cti_][xs]);                                       \
            PCVAE_X3_TAIL(POS)                                                                             \
        }
#define PCVAE_X3_LA(POS, XB)   /* B operand lives in AGPRs */                                              \
        {                                                                                                  \
            constexpr int cti_ = (POS) % CT;                                                               \
            mfma_v_ab<COLD>(r.acc[rt][cti_], af[I], XB[cti_][xs]);                                         \
            PCVAE_X3_TAIL(POS)                                                                             \
        }
#define PCVAE_X3_LH(POS, INIT) PCVAE_X3_L(POS, INIT, xh)
        if constexpr (s < XG::KSH) {
            PCVAE_X3_LH(0, s == 0 && img == 0)
            if constexpr (CT > 1) PCVAE_X3_LH(1, s == 0 && img == 0)
            if constexpr (CT > 2) PCVAE_X3_LH(2, s == 0 && img == 0)
            if constexpr (CT > 3) PCVAE_X3_LH(3, s == 0 && img == 0)
            PCVAE_X3_LA(CT, xl)
            if constexpr (CT > 1) PCVAE_X3_LA(CT + 1, xl)
            if constexpr (CT > 2) PCVAE_X3_LA(CT + 2, xl)
            if constexpr (CT > 3) PCVAE_X3_LA(CT + 3, xl)
        } else {
            PCVAE_X3_LH(0, false)
            if constexpr (CT > 1) PCVAE_X3_LH(1, false)
            if constexpr (CT > 2) PCVAE_X3_LH(2, false)
            if constexpr (CT > 3) PCVAE_X3_LH(3, false)
        }
#undef PCVAE_X3_LH
#undef PCVAE_X3_LA
#undef PCVAE_X3_TAIL
#undef PCVAE_X3_L
        x3_logits<D, CT, OFF, I + 1, HAS_PREV, COLD>(lbase, a0, af, xh, xl, r);
    }
}

struct X3Seam {             // all wave-uniform; one entry per image = per seam of a slot
    const uint16_t* E[2];   // image the chunk to request belongs to
    int64_t n_stage[2];     // first item of that chunk
    char* stage_buf[2];     // ring buffer it goes to (fenced slots: fast_stage)
    unsigned stage_lds[2];  // the same buffer as an LDS byte address (steady-state seams: pipe_stage)
    unsigned next_lbase;    // LDS address (minus the immediate) of the NEXT slot's subtile (image 0)
};

// first MFMA position of gradient step DT (after the 2 CT row-sum MFMAs): per image, hi tiles 2 CT MFMAs, lo tiles CT
template <int D, int CT>
__host__ __device__ constexpr int x3_gpos(int DT) {
    using XG = X3Geo<D, CT>;
    const int img = DT / XG::NDTI, d = DT % XG::NDTI;
    return 2 * CT + img * XG::MGI + (d < 8 ? d * 2 * CT : 8 * 2 * CT + (d - 8) * CT);
}

// G(t-1): gradient chain of the previous subtile (numerators pbh / pbl) with the exponentials of subtile t (r.acc -> r.e) in its
// gaps; in the middle of EACH image's tiles a seam (counted wait + barrier + refill of one ring chunk), and behind the last seam
// the first A fragments of the next slot.  The transposed reads of tile DT + X3_TD are requested behind the first MFMA of step
// DT, so 2 (X3_TD - 1) younger reads are in flight at step DT's wait.
template <int D, int CT, int OFFG, int OFFL_NEXT, int DT, bool HAS_G, int VM, bool COLD, class RG>
__device__ __forceinline__ void x3_grad(const unsigned lbase_g, const int t0, s16x4 (&tl)[X3Geo<D, CT>::NDTL],
                                        s16x4 (&th)[X3Geo<D, CT>::NDTL], const bf16x8 (&pbh)[CT], const bf16x8 (&pbl)[CT],
                                        RG& r, f32x4 (&U)[X3Geo<D, CT>::NDT][CT],
                                        const X3Seam& sm, const int wave_u, const int (&lane_off)[4], const int a0,
                                        bf16x8 (&af)[X3Geo<D, CT>::NI]) {
    using XG = X3Geo<D, CT>;
    constexpr int NDTL = XG::NDTL, NDTI = XG::NDTI;
    constexpr int SEAM_LAST = (XG::NIMG - 1) * NDTI + NDTI / 2;     // the seam that also issues the next slot's first A fragments
    if constexpr (DT < NDTL) {
        constexpr int img = DT / NDTI, d = DT % NDTI;
        if constexpr (d == NDTI / 2) {
            if constexpr (COLD) {
                pipe_fence();
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                if (sm.n_stage[img] >= 0) fast_stage<XG::DL, 4>(sm.E[img], sm.n_stage[img], sm.stage_buf[img], wave_u, lane_off);
                pipe_fence();
            } else if constexpr (!(X3_PROBE & 8)) {
                if constexpr (X3_PROBE & 32) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");   // probe: no barrier
                else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(VM) : "memory");
                if constexpr (!(X3_PROBE & 64)) pipe_stage<XG::DL>(sm.E[img], sm.n_stage[img], sm.stage_lds[img], wave_u, lane_off);   // probe: no refill
            }
            if constexpr (DT == SEAM_LAST && (COLD || !(X3_PROBE & 2)))
                x3_a_prologue<D, CT, OFFL_NEXT, X3_AD>(sm.next_lbase, a0, af);   // first A fragments of the next slot
        }
        if constexpr (HAS_G) {
            // younger than tile DT's reads: tiles DT + 1 .. DT + X3_TD - 1, and the next slot's first A fragments once they are out
            constexpr int young = (DT + X3_TD - 1 < NDTL ? X3_TD - 1 : NDTL - 1 - DT);
            constexpr int extra = (DT >= SEAM_LAST && DT - SEAM_LAST < X3_TD) ? X3_AD : 0;   // issued behind a read that is still awaited
            if constexpr (COLD || !(X3_PROBE & (2 | 4 | 16))) lgkm_wait<2 * young + extra>();
        }
        const s16x8 a16 = __builtin_shufflevector(tl[DT], th[DT], 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 a = __builtin_bit_cast(bf16x8, a16);
        constexpr int M0 = x3_gpos<D, CT>(DT);
        constexpr int UD = img * 8 + d % 8;
        constexpr int KD = CT == 1 ? 4 : 2, KPOS = CT == 1 ? 0 : 1;   // operand reservations: see x3_logits
#define PCVAE_X3_G(POS, PB)                                                                                \
        {                                                                                                  \
            if constexpr (HAS_G) mfma_a<COLD>(U[UD][(POS) % CT], a, PB[(POS) % CT]);                       \
            if constexpr (HAS_G && (POS) == 0 && DT + X3_TD < NDTL && (COLD || !(X3_PROBE & 4)))           \
                x3_tr_issue<D, CT, OFFG, DT + X3_TD>(lbase_g, t0, tl[DT + X3_TD], th[DT + X3_TD]);         \
            if constexpr (HAS_G && (POS) == KPOS && DT >= KD) x3_keep(tl[DT - KD], th[DT - KD]);           \
            if constexpr (HAS_G && (POS) == 1 && DT == 1) {   /* the last fragments of the logits chain in front of this G */ \
                x3_keep(af[XG::NI - 1]); x3_keep(af[XG::NI - 2]); x3_keep(af[XG::NI - 3]);                  \
                if constexpr (KD > 3) x3_keep(af[XG::NI - 4]);                                             \
            }                                                                                              \
            if constexpr (!COLD && !(X3_PROBE & 1)) x3_gops<D, CT, M0 + (POS)>(r);                         \
        }
        if constexpr (d < 8) {
            PCVAE_X3_G(0, pbh)
            if constexpr (CT > 1) PCVAE_X3_G(1, pbh)
            if constexpr (CT > 2) PCVAE_X3_G(2, pbh)
            if constexpr (CT > 3) PCVAE_X3_G(3, pbh)
            PCVAE_X3_G(CT, pbl)
            if constexpr (CT > 1) PCVAE_X3_G(CT + 1, pbl)
            if constexpr (CT > 2) PCVAE_X3_G(CT + 2, pbl)
            if constexpr (CT > 3) PCVAE_X3_G(CT + 3, pbl)
        } else {
            PCVAE_X3_G(0, pbh)
            if constexpr (CT > 1) PCVAE_X3_G(1, pbh)
            if constexpr (CT > 2) PCVAE_X3_G(2, pbh)
            if constexpr (CT > 3) PCVAE_X3_G(3, pbh)
        }
#undef PCVAE_X3_G
        x3_grad<D, CT, OFFG, OFFL_NEXT, DT + 1, HAS_G, VM, COLD>(lbase_g, t0, tl, th, pbh, pbl, r, U, sm, wave_u, lane_off, a0, af);
    } else {
        if constexpr (HAS_G) x3_keep_pb<CT>(pbh, pbl);   // (and on into the next logits chain: x3_logits, step 2)
        if constexpr (COLD) pipe_fence();
    }
}

// one slot: L(t) || split of subtile t-1's numerators, then G(t-1) || exponentials of subtile t
template <int D, int CT, int OFFL, int OFFG, int OFFL_NEXT, bool HAS_G, int VM, bool COLD, class RG>
__device__ __forceinline__ void x3_slot(const unsigned lbase_l, const unsigned lbase_g, const FastLane& L,
                                        const bf16x8 (&xh)[CT][X3Geo<D, CT>::NIMG * X3Geo<D, CT>::KSH],
                                        const bf16x8 (&xl)[CT][X3Geo<D, CT>::NIMG * X3Geo<D, CT>::KSH],
                                        bf16x8 (&af)[X3Geo<D, CT>::NI], RG& r, f32x4 (&U)[X3Geo<D, CT>::NDT][CT],
                                        f32x4 (&lsum)[CT], const X3Seam& sm, const int wave_u, const int (&lane_off)[4]) {
    using XG = X3Geo<D, CT>;
    if constexpr (COLD) pipe_fence();
    x3_logits<D, CT, OFFL, 0, HAS_G, COLD>(lbase_l, L.a0, af, xh, xl, r);
    if constexpr (COLD && HAS_G) {      // fenced slots: nothing overlapped - the split of the previous subtile's numerators now
        pipe_fence();
        x3_all_lops<D, CT>(r);
        asm volatile("s_nop 1" ::: "memory");
    }
    x3_pack<CT>(r.wh, r.pbh);
    x3_pack<CT>(r.wl, r.pbl);
    if constexpr (HAS_G) {
        if constexpr (COLD || !(X3_PROBE & 4)) x3_tr_prologue<D, CT, OFFG, X3_TD>(lbase_g, L.t0, r.tl, r.th);
#define PCVAE_X3_ONES(POS, PB) mfma_a<COLD>(lsum[(POS) % CT], L.ones, PB[(POS) % CT]);
        PCVAE_X3_ONES(0, r.pbh)
        if constexpr (CT > 1) PCVAE_X3_ONES(1, r.pbh)
        if constexpr (CT > 2) PCVAE_X3_ONES(2, r.pbh)
        if constexpr (CT > 3) PCVAE_X3_ONES(3, r.pbh)
        PCVAE_X3_ONES(CT, r.pbl)
        if constexpr (CT > 1) PCVAE_X3_ONES(CT + 1, r.pbl)
        if constexpr (CT > 2) PCVAE_X3_ONES(CT + 2, r.pbl)
        if constexpr (CT > 3) PCVAE_X3_ONES(CT + 3, r.pbl)
#undef PCVAE_X3_ONES
    }
    if constexpr (COLD) {               // exponentials of this subtile, nothing overlapped
        pipe_fence();
        x3_all_gops<D, CT>(r);
        asm volatile("s_nop 1" ::: "memory");
    }
    x3_grad<D, CT, OFFG, OFFL_NEXT, 0, HAS_G, VM, COLD>(lbase_g, L.t0, r.tl, r.th, r.pbh, r.pbl, r, U, sm, wave_u, lane_off, L.a0, af);
}

// ---- fenced gradient of one subtile (drain, ragged tail): nothing overlapped.  `istride`: bytes from image 0 to image 1 of the
// subtile (the ring: one chunk; the synchronously staged tail: its own image area)
template <int D, int CT, int DT = 0>
__device__ __forceinline__ void x3_cold_grad(const unsigned lbase_g, const unsigned istride, const int t0,
                                             s16x4 (&tl)[X3Geo<D, CT>::NDTL], s16x4 (&th)[X3Geo<D, CT>::NDTL],
                                             const bf16x8 (&pbh)[CT], const bf16x8 (&pbl)[CT], f32x4 (&U)[X3Geo<D, CT>::NDT][CT]) {
    using XG = X3Geo<D, CT>;
    if constexpr (DT < XG::NDTL) {
        if constexpr (DT + 2 < XG::NDTL)
            tr_issue<XG::DL, 0, (DT + 2) % XG::NDTI>(lbase_g + ((DT + 2) / XG::NDTI) * istride, t0, tl[DT + 2], th[DT + 2]);
        lgkm_wait<2 * ((DT + 2 < XG::NDTL ? DT + 2 : XG::NDTL - 1) - DT)>();
        const s16x8 a16 = __builtin_shufflevector(tl[DT], th[DT], 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 a = __builtin_bit_cast(bf16x8, a16);
        constexpr int UD = (DT / XG::NDTI) * 8 + (DT % XG::NDTI) % 8;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            mfma_a<true>(U[UD][ct], a, pbh[ct]);
            if constexpr (DT % XG::NDTI < 8) mfma_a<true>(U[UD][ct], a, pbl[ct]);
        }
        x3_cold_grad<D, CT, DT + 1>(lbase_g, istride, t0, tl, th, pbh, pbl, U);
    }
}
template <int D, int CT>
__device__ __forceinline__ void x3_cold_gradient(const unsigned lbase_g, const unsigned istride, const FastLane& L,
                                                 const bf16x8 (&pbh)[CT], const bf16x8 (&pbl)[CT],
                                                 f32x4 (&U)[X3Geo<D, CT>::NDT][CT], f32x4 (&lsum)[CT]) {
    using XG = X3Geo<D, CT>;
    s16x4 tl[XG::NDTL], th[XG::NDTL];
    pipe_fence();
    tr_issue<XG::DL, 0, 0>(lbase_g, L.t0, tl[0], th[0]);
    tr_issue<XG::DL, 0, 1>(lbase_g, L.t0, tl[1], th[1]);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        mfma_a<true>(lsum[ct], L.ones, pbh[ct]);
        mfma_a<true>(lsum[ct], L.ones, pbl[ct]);
    }
    x3_cold_grad<D, CT>(lbase_g, istride, L.t0, tl, th, pbh, pbl, U);
    pipe_fence();
}

// one subtile on its own (ragged tail): logits, bound check, numerators, gradient - nothing overlapped, every MFMA fenced
template <int D, int CT, int I = 0>
__device__ __forceinline__ void x3_cold_logits(const unsigned lbase, const unsigned istride, const int a0,
                                               bf16x8 (&af)[X3Geo<D, CT>::NI],
                                               const bf16x8 (&xh)[CT][X3Geo<D, CT>::NIMG * X3Geo<D, CT>::KSH],
                                               const bf16x8 (&xl)[CT][X3Geo<D, CT>::NIMG * X3Geo<D, CT>::KSH], f32x4 (&acc)[2][CT]) {
    using XG = X3Geo<D, CT>;
    if constexpr (I < XG::NI) {
        constexpr int img = I / XG::NIL, i = I % XG::NIL;
        pipe_a_issue<XG::DL, 0, i>(lbase + img * istride, a0, af[I]);
        lgkm_wait<0>();
        constexpr int s = i >> 1, rt = i & 1, xs = img * XG::KSH + s % XG::KSH;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            if constexpr (s == 0 && img == 0) mfma_v0<true>(acc[rt][ct], af[I], xh[ct][xs]);
            else mfma_v<true>(acc[rt][ct], af[I], xh[ct][xs]);
            if constexpr (s < XG::KSH) mfma_v_ab<true>(acc[rt][ct], af[I], xl[ct][xs]);
        }
        x3_cold_logits<D, CT, I + 1>(lbase, istride, a0, af, xh, xl, acc);
    }
}
__device__ __forceinline__ unsigned x3_pack_rne(float a, float b) {   // two fp32 -> packed bf16 pair (RNE), a in the low half
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int D, int CT>
__device__ __forceinline__ void x3_solo(const unsigned lbase, const unsigned istride, const int64_t n0, const int64_t N, const FastLane& L,
                                        const bf16x8 (&xh)[CT][X3Geo<D, CT>::NIMG * X3Geo<D, CT>::KSH],
                                        const bf16x8 (&xl)[CT][X3Geo<D, CT>::NIMG * X3Geo<D, CT>::KSH],
                                        f32x4 (&U)[X3Geo<D, CT>::NDT][CT], f32x4 (&lsum)[CT]) {
    using XG = X3Geo<D, CT>;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc[2][CT];
    bf16x8 af[XG::NI];
    pipe_fence();
    x3_cold_logits<D, CT>(lbase, istride, L.a0, af, xh, xl, acc);
    pipe_fence();
    bf16x8 pbh[CT], pbl[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        u32x4 wh, wl;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = 2 * h;
                const bool ok0 = n0 + 16 * rt + 4 * L.g + i < N, ok1 = n0 + 16 * rt + 4 * L.g + i + 1 < N;
                const float e0 = ok0 ? __builtin_amdgcn_exp2f(acc[rt][ct][i]) : 0.f;
                const float e1 = ok1 ? __builtin_amdgcn_exp2f(acc[rt][ct][i + 1]) : 0.f;
                const unsigned hi = x3_pack_rne(e0, e1);
                wh[2 * rt + h] = hi;
                wl[2 * rt + h] = x3_pack_rne(e0 - __uint_as_float(hi << 16), e1 - __uint_as_float(hi & 0xffff0000u));
            }
        pbh[ct] = __builtin_bit_cast(bf16x8, wh);
        pbl[ct] = __builtin_bit_cast(bf16x8, wl);
    }
    asm volatile("s_nop 1" ::: "memory");
    x3_cold_gradient<D, CT>(lbase, istride, L, pbh, pbl, U, lsum);
}

// rx row fragments: xh = RNE bf16(rx * log2 e), xl = RNE bf16(rx * log2 e - xh), laid out like the bf16 kernels' B operand;
// fragment img * 4 + s multiplies k-step s of image img (dims 128 img ..)
template <int D>
__device__ __forceinline__ void x3_load_x(const float* __restrict__ rx, const int64_t row, const int g, bf16x8 (&xh)[D / 32],
                                          bf16x8 (&xl)[D / 32]) {
#pragma unroll
    for (int f = 0; f < D / 32; ++f) {
        const int img = f >> 2, s = f & 3;
        const int c0 = img * 128 + 8 * fchunk<256>(s, g);   // hi-half chunk of k-step s (the lo k-step s + 4 multiplies the same columns)
        const float4 v0 = *reinterpret_cast<const float4*>(rx + row * D + c0);
        const float4 v1 = *reinterpret_cast<const float4*>(rx + row * D + c0 + 4);
        const float v[8] = {v0.x * kLog2e, v0.y * kLog2e, v0.z * kLog2e, v0.w * kLog2e,
                            v1.x * kLog2e, v1.y * kLog2e, v1.z * kLog2e, v1.w * kLog2e};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const __bf16 h = (__bf16)v[j];
            xh[f][j] = h;
            xl[f][j] = (__bf16)(v[j] - (float)h);
        }
    }
}

template <int D, int CT>
__global__ void __launch_bounds__(256, 1) catalog_ce_x3_pipe_kernel(CatParamsB p) {
    using XG = X3Geo<D, CT>;
    using GL = typename XG::GL;
    constexpr int CB = XG::CB, NW = 4, ROWS = XG::ROWS, TR = XG::TR, NB = XG::NB, PF = XG::PF, DL = XG::DL, NIMG = XG::NIMG;
    constexpr int NX = NIMG * XG::KSH;                                   // rx fragments per column tile and half
    static_assert(XG::gfirst(XG::MG) == XG::GOPS && XG::gfirst(XG::MG - 1) == XG::GOPS, "every exponential has a gap");
    static_assert(XG::lfirst(XG::ML - 2) == XG::LOPS, "every split op has a gap, the last two gaps of L stay free");
    static_assert(XG::gfirst(2 * CT + 1) == 0, "no exponential before the row-sum MFMAs are out: the accumulators are fresh");
    // seam h requests chunk h + NIMG + PF into the buffer of chunk h + NIMG + PF - NB, which must be dead: at seam h the gradient
    // chain still reads chunks >= h - NIMG, so NB >= PF + 2 NIMG + 1; NB a multiple of NIMG: image 1 never wraps away from image 0
    static_assert(NB % NIMG == 0 && NB >= PF + 2 * NIMG + 1 && TR >= 4 && TR <= 6, "ring geometry");
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int nrb = (int)((p.R + ROWS - 1) / ROWS);
    const int split = logical / nrb, rb = logical % nrb;
    if (p.safe_flags[(int)(((int64_t)rb * ROWS) / ROWS_WG)] != 0) return;   // large |rx|: the exact f32 kernel handles this block
    const int t_beg = split * p.tiles_per_split;
    const int t_end = min(t_beg + p.tiles_per_split, p.ntiles);
    const int64_t nbase = (int64_t)t_beg * 32;
    int T = (int)min((int64_t)(t_end - t_beg), (p.N - nbase) / GL::BNF);    // full 32-item subtiles = slots (NIMG ring chunks each)
    T = max(T, 0);
    // ring chunk q = (subtile q / NIMG, image q % NIMG); image i of the table starts i * N rows behind image 0
    auto img_of = [&](int q) { return p.E + (int64_t)(q % NIMG) * p.N * DL; };
    auto item_of = [&](int q) { return nbase + (int64_t)min(q / NIMG, T - 1) * GL::BNF; };   // beyond the end: the last subtile again

    const int64_t rw = (int64_t)rb * ROWS + wave * 16 * CT;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int lane_off[4];
    fast_lane_off<DL, NW>(lane, wave, lane_off);
#pragma unroll
    for (int q = 0; q < PF + NIMG; ++q)   // a constant number of chunks in flight from here on
        if (T > 0) fast_stage<DL, NW>(img_of(q), item_of(q), smem + q * CB, wave_u, lane_off);

    bf16x8 xh[CT][NX], xl[CT][NX];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int64_t r = rw + 16 * ct + c;
        x3_load_x<D>(p.rx, r < p.R ? r : p.R - 1, g, xh[ct], xl[ct]);
#pragma unroll
        for (int s = 0; s < NX; ++s) asm volatile("" : "+a"(xl[ct][s]));   // the lo fragments live in AGPRs (mfma_v_ab)
    }
    f32x4 U[XG::NDT][CT];
    f32x4 lsum[CT];
#pragma unroll
    for (int dt = 0; dt < XG::NDT; ++dt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int i = 0; i < 4; ++i) U[dt][ct][i] = 0.f;
            asm volatile("" : "+a"(U[dt][ct]));
        }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int i = 0; i < 4; ++i) lsum[ct][i] = 0.f;
        asm volatile("" : "+a"(lsum[ct]));
    }
    const FastLane L = fast_lane<DL>(lane);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    X3Regs<CT, XG::NDTL> r;
    bf16x8 af[XG::NI];

    auto lds_of = [&](int t) { return lds0 + (unsigned)(((NIMG * t) % NB) * CB); };   // image 0 of slot t's subtile
    auto seam_of = [&](int t) {          // seam k of slot t is seam h = NIMG t + k of the range: it requests chunk h + NIMG + PF
        X3Seam sm;
#pragma unroll
        for (int k = 0; k < NIMG; ++k) {
            const int q = NIMG * t + k + NIMG + PF;
            sm.E[k] = img_of(q);
            sm.n_stage[k] = item_of(q);
            sm.stage_buf[k] = smem + (q % NB) * CB;
            sm.stage_lds[k] = lds0 + (unsigned)((q % NB) * CB);
        }
        sm.next_lbase = lds0;
        return sm;
    };

    int t = 0;
    if (T > 0) {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PF * 4) : "memory");   // the chunks of subtile 0 landed
        x3_a_prologue<D, CT, 0, X3_AD>(lds0, L.a0, af);
        {   // slot 0: nothing to drain yet
            X3Seam sm = seam_of(0);
            sm.next_lbase = lds_of(T > 1 ? 1 : 0);
            x3_slot<D, CT, 0, 0, 0, false, 0, true>(lds0, lds0, L, xh, xl, af, r, U, lsum, sm, wave_u, lane_off);
        }
        t = 1;
        // steady state: TR slots per trip, every LDS offset an immediate
        for (; t + TR <= T; t += TR) {
            // hipcc copies the loop-carried values (the prologue's A fragments, the numerators of the fill slot) into the loop's
            // registers in the PREHEADER - plain v_mov's, and the first statement of a trip is an asm MFMA that reads one of them
            // (af[0]) with no wait states of its own: a VALU write needs >= 2 wait states before an MFMA reads the register.
            // Round 3 met it: with the copies in another order (a rebuild with one more instantiation in the translation unit)
            // the first MFMA of the first trip read a stale fragment - column tile 0 of every wave wrong.  Fenced here, per trip.
            pipe_fence();
#define PCVAE_X3S(UU)                                                                                                     \
            {                                                                                                             \
                constexpr int TL = 1 + UU, TG = UU, TN = 2 + UU;                                                          \
                constexpr int OL = ((NIMG * TL) % NB) * CB, OG = ((NIMG * TG) % NB) * CB, ON = ((NIMG * TN) % NB) * CB;   \
                X3Seam s2 = seam_of(t + UU);                                                                              \
                s2.stage_lds[0] = lds0 + ((NIMG * TL + NIMG + PF) % NB) * CB;   /* t = 1 (mod TR): constants */           \
                if constexpr (NIMG > 1) s2.stage_lds[NIMG - 1] = lds0 + ((NIMG * TL + NIMG - 1 + NIMG + PF) % NB) * CB;   \
                x3_slot<D, CT, OL, OG, ON, true, (PF - 1) * 4, false>(lds0, lds0, L, xh, xl, af, r, U, lsum, s2, wave_u, lane_off); \
            }
            PCVAE_X3S(0) PCVAE_X3S(1) PCVAE_X3S(2) PCVAE_X3S(3)
            if constexpr (TR > 4) { PCVAE_X3S(4) }
            if constexpr (TR > 5) { PCVAE_X3S(5) }
#undef PCVAE_X3S
            pipe_fence();  // latch
        }
        // at most TR - 1 slots are left: fenced slots with runtime ring offsets (their seams drain the ring: vmcnt(0))
        for (; t < T; ++t) {
            X3Seam s2 = seam_of(t);
            s2.next_lbase = lds_of(t + 1 < T ? t + 1 : t);
            x3_slot<D, CT, 0, 0, 0, true, 0, true>(lds_of(t), lds_of(t - 1), L, xh, xl, af, r, U, lsum, s2, wave_u, lane_off);
        }
        {   // drain: the lo halves of the last subtile's numerators, then its gradient chain
            bf16x8 pbh[CT], pbl[CT];
            pipe_fence();
            x3_all_lops<D, CT>(r);
            x3_pack<CT>(r.wh, pbh);
            x3_pack<CT>(r.wl, pbl);
            asm volatile("s_nop 1" ::: "memory");
            x3_cold_gradient<D, CT>(lds_of(T - 1), (unsigned)CB, L, pbh, pbl, U, lsum);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive its wave
    // ---- tail: the ragged last subtile of the catalog (and ranges shorter than one chunk), staged synchronously: up to 4 subtiles
    // of image 0 at smem, of image 1 at smem + 64 KB
    constexpr unsigned TAIL_IMG = 128 * GL::RB;   // 64 KB
    static_assert(NIMG * TAIL_IMG <= (unsigned)(NB * CB), "the tail images fit the ring's LDS");
    for (int tt = t_beg + T; tt < t_end; tt += 4) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NIMG; ++i) fast_stage_tail<DL, NW>(p.E + (int64_t)i * p.N * DL, p.N, (int64_t)tt * 32, smem + i * TAIL_IMG);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int nsub = min(4, t_end - tt);
        for (int st = 0; st < nsub; ++st) x3_solo<D, CT>(lds0 + st * GL::ST, TAIL_IMG, (int64_t)(tt + st) * 32, p.N, L, xh, xl, U, lsum);
    }
    pipe_fence();
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const float l = lsum[ct][0];
        const int64_t row = rw + 16 * ct + c;
        if (row < p.R) {
            const int64_t o = (int64_t)split * p.R + row;
            if (g == 0) { p.pm[o] = 0.f; p.pl[o] = l; }
#pragma unroll
            for (int dt = 0; dt < XG::NDT; ++dt)
                *reinterpret_cast<float4*>(p.pU + o * D + 16 * dt + 4 * g) =
                    make_float4(U[dt][ct][0], U[dt][ct][1], U[dt][ct][2], U[dt][ct][3]);
        }
    }
}

// one wave per row: sum the split partials (all max-free: pm = 0), exact fp32 target logit and target row from the fp32 table
template <int D>
__global__ void __launch_bounds__(256) catalog_ce_merge_x3_kernel(CatParamsB p, const float* __restrict__ Ef,
                                                                  float* __restrict__ nll, float* __restrict__ lse,
                                                                  float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    // the row is wave-uniform, and known to be: the target index, the table row and the rx row then come through the SCALAR
    // cache (s_load) instead of 64 vector loads of one address per step of the logit chain - as vector loads those 64 broadcast
    // requests per wave kept the address unit busy for ~150 of this kernel's 196 us at config 4
    const int64_t r = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (r >= p.R) return;
    if (p.safe_flags[r / ROWS_WG] != 0) return;   // this row block ran the exact f32 kernel (its merge writes the row)
    // everything that does not depend on the target index is requested first: the gradient partials (two columns per lane) ride
    // along with the index -> table row -> logit chain instead of queueing behind it
    constexpr int CPL = D / 64;   // columns per lane (2 at D = 128, 4 at D = 256): one 8- or 16-byte access per partial
    typedef float fcpl __attribute__((ext_vector_type(CPL)));
    const int64_t t = p.target[r];
    float L = lane < p.nsplit ? p.pl[(int64_t)lane * p.R + r] : 0.f;
    fcpl u;
#pragma unroll
    for (int i = 0; i < CPL; ++i) u[i] = 0.f;
    if (dx)   // eight ranges requested together, added in range order (the same sums as one load per trip, an eighth of the round trips)
        for (int j0 = 0; j0 < p.nsplit; j0 += 8) {
            fcpl v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int j = j0 + q < p.nsplit ? j0 + q : p.nsplit - 1;
                v[q] = *reinterpret_cast<const fcpl*>(p.pU + ((int64_t)j * p.R + r) * D + CPL * lane);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (j0 + q < p.nsplit) u += v[q];
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) L += __shfl_xor(L, o, 64);
    const bool t_ok = t >= 0 && t < p.N;
    float zt = 0.f;   // the k-ordered fmaf chain of the f32 kernel / oracle (16-byte loads, the same order)
    if (t_ok) {
        typedef float f32x4m __attribute__((ext_vector_type(4)));
        const f32x4m* e4 = reinterpret_cast<const f32x4m*>(Ef + t * D);
        const f32x4m* x4 = reinterpret_cast<const f32x4m*>(p.rx + r * D);
#pragma unroll 8
        for (int k = 0; k < D / 4; ++k) {
            const f32x4m e = e4[k], x = x4[k];
            zt = fmaf(e[0], x[0], zt);
            zt = fmaf(e[1], x[1], zt);
            zt = fmaf(e[2], x[2], zt);
            zt = fmaf(e[3], x[3], zt);
        }
    }
    const float lse_r = log2f(L) * kLn2;
    if (lane == 0) {
        nll[r] = t_ok ? lse_r - zt : NAN;
        if (lse) lse[r] = lse_r;
    }
    if (dx) {
        const float invL = 1.f / L;
        fcpl o;
#pragma unroll
        for (int i = 0; i < CPL; ++i) o[i] = NAN;
        if (t_ok) {
            const fcpl e = *reinterpret_cast<const fcpl*>(Ef + t * D + CPL * lane);
#pragma unroll
            for (int i = 0; i < CPL; ++i) o[i] = (u[i] * invL - e[i]) * p.dx_scale;
        }
        *reinterpret_cast<fcpl*>(dx + r * D + CPL * lane) = o;
    }
}
