// K5 in SPLIT-bf16 arithmetic on the bf16 matrix cores: the fused full-catalog softmax cross-entropy of catalog_bf16.hip with BOTH
// operands of both contractions written as a sum of NC bf16 components, x = c0 + c1 (+ c2), c0 = RNE bf16(x), c1 = RNE bf16(x - c0),
// c2 = RNE bf16(x - c0 - c1) (the differences are exact in fp32), and one bf16 MFMA per kept component pair (i, j), i + j < NC,
// fp32 accumulate:
//   NC = 2  "bf16x3"  16-bit-mantissa operands, 3 MFMAs per product (c0 c0 + c0 c1 + c1 c0; the dropped c1 c1 is 2^-18 relative).
//                     A stated-tolerance fast path: NARROWER than the reference's fp32.
//   NC = 3  "bf16x6"  x = c0 + c1 + c2 EXACTLY (3 x 8 = 24 significand bits: the fp32 operand itself), 6 MFMAs per product
//                     (c0 c0, c0 c1, c1 c0, c1 c1, c0 c2, c2 c0; the dropped c1 c2, c2 c1, c2 c2 are <= 2^-25 relative - below the
//                     rounding of an fp32 product), every partial product exact in the fp32 accumulator's input: the reference's
//                     own arithmetic (fp32 operands, fp32 accumulation) on the 16x faster pipe.  (round 4)
// Replaces `mm` + `downsample(n_neg = N)` + `CrossEntropyLoss` + their backward (models/pivotcvae.py:274, train_generative.py:59).
//
// This file is included by catalog_bf16.hip INSIDE its anonymous namespace (it reuses that file's LDS geometry, LDS-DMA staging
// and inline-asm MFMA / LDS-read helpers); it is not a translation unit of its own.
//
// Table image (pcvae_split_bf16x2 / pcvae_split_bf16x3): row n = c0(E_n)[128] | c1(E_n)[128] (| c2(E_n)[128]) bf16 = NC parts of
// 256 bytes.  A part is one LDS bank row, so the 16-chunk XOR swizzle of the D = 128 bf16 kernel applies inside each part and both
// read patterns (ds_read_b128 rows, ds_read_b64_tr_b16 transposed) keep their conflict-free banking whatever NC is; for NC = 2
// the row is exactly the geometry of the D = 256 bf16 kernel (FastGeo<256>).  One 32-item subtile = one ring chunk of 8 NC KB.
//   logits    k-steps 4 i .. 4 i + 3 read part i of a row.  Step (s, rt), part i = s / 4:
//               acc[rt][ct] += A_i(s) . x_j[ct][s % 4]   for j = 0 .. NC - 1 - i
//   numerators p = exp2(acc) (fp32) -> p_0 = RNE bf16(p), p_1 = RNE bf16(p - p_0) (, p_2 = RNE bf16(p - p_0 - p_1))
//   row sums  lsum[ct] += ones . p_j[ct]  for every j   (the same numerators the gradient chain multiplies)
//   gradient  d tiles 8 i .. 8 i + 7 are E_i^T:
//               U[DT % 8][ct] += T_i(DT) . p_j[ct]   for j = 0 .. NC - 1 - i
// NC = 2: 100 MFMAs per 32-item subtile and 32-row wave (2 x 24 CT + 4 row sums at CT = 2) against the bf16 kernel's 36;
// NC = 3: 198 (2 x 48 CT + 6).
//
// Schedule (one wave per SIMD, CT = 2 column tiles of 16 rows per wave, every MFMA / LDS read / VALU op an inline-asm statement
// in schedule order - see catalog_ce_bf16_pipe_kernel for the method and for what hipcc may not place in these loops):
//   slot t:  L(t)    logits chain of subtile t  ||  the split of subtile t-1's numerators (6 / 11 cheap VALU ops per pair)
//            G(t-1)  row sums + gradient chain of subtile t-1  ||  the exponentials of subtile t, the seam (counted
//                    vmcnt + s_barrier + refill of the ring) and the first A fragments of L(t+1) in the middle of the chain.
// The slot is bound by VECTOR ISSUE, not by the MFMA pipe alone: an MFMA holds the issue port for 8 of its 16 cycles, a v_exp_f32
// for 8, a conversion / LDS read / s_waitcnt for 4-5, and a gap runs max(16, the sum) (MI355X_MICROARCH.md, issue-cost row).  So
// the statement order gives every MFMA gap at most 8 cycles of other work: a step is  wait | MFMA | LDS reads of a later step |
// MFMA | op | MFMA | op | MFMA ..., the exponentials sit one per gap in middle gaps of the part-0 tiles of G, the split ops at most
// two (NC = 2) / one (NC = 3) per gap in L.
// A single accumulation chain of v_mfma_f32_16x16x32_bf16 issues back to back at full rate (MI355X_MICROARCH.md, cycle
// constants), so the MFMAs that update one accumulator need no interleaving.
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
#ifndef X3_DMA_SPREAD
#define X3_DMA_SPREAD 3      // the seam's LDS-DMA refill in groups of X3_DMA_PPS pieces, one group per X3_DMA_SPREAD gradient steps
                             // (0: a burst of 2 NC pieces at the seam).  Round 4, bf16x6 at config 4, same box: burst 148.0 ms; pairs
                             // every 1 / 3 steps 146.8 / 146.5 ms with the L2 misses unchanged (121 M per launch = one per 32
                             // workgroups); SINGLE pieces per step 146.5 ms but 380 M misses (L2 hit rate 96.8 -> 90.1 %: 44 GB
                             // instead of 15.5 GB of memory-side reads per launch) - profiles/r04_dma_spread.txt
#endif
#ifndef X3_DMA_PPS
#define X3_DMA_PPS 2         // pieces per group of the spread refill (adjacent 1 KiB pieces: 2 KiB of one table row range)
#endif
#ifndef X3_PROBE
#define X3_PROBE 0
#endif
// bf16x6: the logits of a subtile in TWO accumulators - the c0 c0 products (16-bit, never truncated against the sum) in one, the
// five small component products (2^-8 .. 2^-17 of it) in the other - added once per slot in fp32.  The matrix core aligns the 33
// addends of an MFMA to the largest and keeps 27 bits of each (tools/mfma_round_probe.hip): small products added to a big
// accumulator lose their low bits toward zero, in their own accumulator they do not.  (rx component 0 then lives in AGPRs too.)
#ifndef X6_SPLIT_ACC
#define X6_SPLIT_ACC 1
#endif
// bf16x6: the row sums of a catalog range in CHUNKS of one steady-state trip (6 slots = 192 items): the accumulator restarts with
// every trip (its first row-sum MFMA takes C = 0) and the finished chunk is added to a running fp32 total by one v_add per column
// tile and trip.  A single accumulator over 10^5 .. 10^6 items grows far above the numerators it is fed; the matrix core then
// truncates their low bits toward zero (27-bit alignment) and the sum of POSITIVE terms comes out low: -6.6e-6 relative on rows
// with logits of +-10 over 20 000 items in an emulation of that arithmetic, +2e-8 with chunks (a plain fp32 chain: +2e-7).
#ifndef X6_CHUNK_LSUM
#define X6_CHUNK_LSUM 1
#endif

// D = 256 (round 3, NC = 2 only): the table is TWO images - dims 0..127 and dims 128..255, each [N, 256] bf16 c0 | c1 with the
// 512-byte rows of the D = 128 kernel - and a slot walks both: the logits chain is 16 k-steps per image into the SAME accumulators,
// the gradient chain 16 transposed tiles per image into U tiles 0..7 / 8..15, with one seam (wait + barrier + refill of ONE 16 KB
// chunk) in the middle of each image's tiles.  Ring chunk c = (subtile c / 2, image c % 2); 8 ring buffers, so image 1 of a subtile
// always sits 16 KB behind image 0 (no wrap between them).  (NC = 3 at D = 256 would need 8 x 24 KB of ring: more LDS than a CU has.)
template <int D, int CT, int NC = 2>
struct X3Geo {
    static_assert(D == 128 || D == 256, "built for D = 128 (one image) and D = 256 (two images of 128 dims)");
    static_assert(NC == 2 || (NC == 3 && D == 128), "bf16x3: D = 128 / 256; bf16x6: D = 128");
    static constexpr int NIMG = D / 128;                   // images of 128 dims
    static constexpr int DL = 128 * NC;                    // row width of ONE image in bf16 elements: c0 | c1 (| c2)
    using GL = FastGeo<DL>;                                // (only its row geometry is used: RB, RT, ST, KS, NDT, KMASK)
    static constexpr int RB = GL::RB;                      // bytes per table row
    static_assert(RB == 256 * NC && GL::KS == 4 * NC && GL::NDT == 8 * NC && GL::KMASK == 3, "row geometry: NC parts of 256 bytes");
    static constexpr int BNF = 32;                         // items per ring chunk = one subtile
    static constexpr int CB = BNF * RB;                    // ring chunk = one 32-item subtile of one image: 16 / 24 KB
    static constexpr int PPW = CB / 4096;                  // 1 KiB LDS-DMA pieces per wave (4 waves) and ring chunk
    static constexpr int KSH = 4;                          // k-steps per part of an image
    static constexpr int NIL = 2 * GL::KS;                 // logits steps per image: (k-step, row tile)
    static constexpr int NI = NIMG * NIL;                  // steps of the logits chain
    static constexpr int NDTI = GL::NDT;                   // transposed tiles per image: 8 per part
    static constexpr int NDTL = NIMG * NDTI;
    static constexpr int NDT = D / 16;                     // 16-wide d tiles of U
    static constexpr int TRI = NC * (NC + 1) / 2;          // kept component pairs = MFMAs per algorithmic multiply-add
    static constexpr int MLI = 2 * KSH * CT * TRI;         // MFMAs of L per image
    static constexpr int ML = NIMG * MLI;
    static constexpr int MGI = 8 * CT * TRI;               // MFMAs of G per image
    static constexpr int NRS = NC * CT;                    // row-sum MFMAs (first in G)
    static constexpr int MG = NRS + NIMG * MGI;
    static constexpr int P = 4 * CT;                       // numerator pairs per slot: (row tile, column tile, half)
    static constexpr bool SPLIT = NC == 3 && X6_SPLIT_ACC; // logits in two accumulators (see X6_SPLIT_ACC)
    static constexpr bool CHUNK = NC == 3 && X6_CHUNK_LSUM; // row sums restart every steady-state trip (see X6_CHUNK_LSUM)
    static constexpr int GOPS = (SPLIT ? 4 : 2) * P;       // during G: (the 2 additions acc += acc_lo and) the 2 exponentials of every pair
    static constexpr int OPP = NC == 2 ? 6 : 11;           // split ops per pair: conversion, shift, mask, 2 subtractions, per component after the first
    static constexpr int LOPS = OPP * P;                   // during the next L
    static constexpr int ROWS = 4 * 16 * CT;               // rows per workgroup (4 waves)
    static constexpr int PF = 3;                           // chunks requested ahead
    static constexpr int NB = NIMG == 1 ? 6 : 8;           // ring buffers (>= PF + 3 NIMG - 1 live or in flight; a multiple of NIMG)
    static constexpr int TR = NB / NIMG;                   // slots per steady-state trip (every LDS offset an immediate)
    static_assert(NB * CB <= 160 * 1024, "the ring fits a CU's LDS");
    // ---- positions.  A chain (L: steps of (k-step, row tile); G: transposed tiles) walks the parts of an image in order; a step of
    // part i is (NC - i) CT MFMAs: x / p component j = POS / CT, column tile POS % CT.
    static constexpr int part_base(int per_part, int part) {     // MFMAs of the parts in front of `part` (per_part steps each)
        int n = 0;
        for (int q = 0; q < part; ++q) n += per_part * (NC - q) * CT;
        return n;
    }
    struct Pos { int part, len, j; };                       // of MFMA m of one image's chain: its step's part, length, index in the step
    static constexpr Pos pos_of(int mm, int per_part) {
        int part = 0;
        while (part + 1 < NC && mm >= part_base(per_part, part + 1)) ++part;
        const int len = (NC - part) * CT;
        return Pos{part, len, (mm - part_base(per_part, part)) % len};
    }
    // Statement order inside a step:  wait(this step's LDS data) | MFMA 0 | LDS reads of a later step | MFMA 1 | ops | ... | MFMA last | ops
    // ---- G: row sums, then per image the tiles.  One exponential behind MIDDLE MFMAs of a part-0 tile: the gap behind MFMA 0
    // carries the two transposed reads (8 cycles), the one behind the last MFMA the next step's wait.  NC = 3: a part-0 tile has
    // 3 CT MFMAs; every other middle gap takes one (2 per tile at CT = 2, as for NC = 2)
    static constexpr int gcap(int m) {
        if (m < NRS || m >= MG) return 0;
        const Pos p = pos_of((m - NRS) % MGI, 8);
        if (p.part != 0) return 0;
        // CT = 1 (NC = 2): a part-0 tile is two MFMAs, the one exponential shares the gap of the transposed reads (not behind the very
        // first tile: the logits accumulators are fresh there)
        if (CT == 1) return (p.j == 0 && m > NRS) ? 1 : 0;
        if (p.j == 0 || p.j == p.len - 1) return 0;
        return (NC == 2 || SPLIT) ? 1 : (p.j & 1);
    }
    static constexpr int gfirst(int m) {
        int n = 0;
        for (int i = 0; i < m && i < MG; ++i) n += gcap(i);
        return n > GOPS ? GOPS : n;
    }
    // ---- L.  Cheap ops (4-5 cycles).  NC = 2: one next to the A-fragment read behind MFMA 0, one in front of the next step's wait
    // behind the last MFMA, two in the gaps between.  NC = 3: one per gap (88 ops, 96 MFMAs).  None in the last two gaps of L (the
    // packed numerators are MFMA operands right after it)
    static constexpr int lcap(int m) {
        // nothing behind the first 2 CT MFMAs: the first split op overwrites the c0 numerators (r.w[0]) that the LAST MFMAs of the
        // gradient chain in front of this L read as their B operand.  hipcc sees those operands dead and the in-order issue would
        // seem to protect them, but a VALU write two MFMAs behind such a read corrupted it (column tile 0 - the first pair
        // written - NaN, depending on where an unrelated ds_read sat: experiments/tools/dbg_x3.py, HISTORY.md 3.1b): MFMAs queue in front of the
        // matrix pipe and read their operands when they start, not when they issue.  2 CT MFMAs of distance.
        if (m < 2 * CT || m >= ML - 2) return 0;
        const Pos p = pos_of(m % MLI, 2 * KSH);
        if (NC == 3) return 1;
        return (p.j == 0 || p.j == p.len - 1) ? 1 : 2;
    }
    static constexpr int lfirst(int m) {
        int n = 0;
        for (int i = 0; i < m && i < ML; ++i) n += lcap(i);
        return n > LOPS ? LOPS : n;
    }
    // first MFMA position of logits step I / of gradient step DT (after the row-sum MFMAs)
    static constexpr int lpos(int I) {
        const int img = I / NIL, i = I % NIL, part = i / (2 * KSH);
        return img * MLI + part_base(2 * KSH, part) + (i % (2 * KSH)) * (NC - part) * CT;
    }
    static constexpr int gpos(int DT) {
        const int img = DT / NDTI, d = DT % NDTI, part = d / 8;
        return NRS + img * MGI + part_base(8, part) + (d % 8) * (NC - part) * CT;
    }
};

template <int CT, int NT, int NC>
struct X3Regs {
    f32x4 acc[2][CT];          // logits of the current subtile [row tile][column tile] (log2 domain)
    f32x4 accl[2][CT];         // (bf16x6, X6_SPLIT_ACC) the small component products of the same logits
    unsigned w[NC][CT][4];     // [component][ct][2 rt + h]: bf16 pairs; written during L, read by the G right behind it
    float e[4 * CT][2];        // the fp32 numerators: written during G (exponentials), split during the next L
    float tmp[2][2];           // running residuals of the two pairs in flight
    float tm2[2][2];           // (NC = 3) the second component as fp32
    // MFMA operands that must outlive their last MFMA (see x3_keep): the packed numerators and the transposed tiles of the
    // gradient chain, held here so that the NEXT slot's logits chain can still name them
    bf16x8 pb[NC][CT];
    s16x4 tl[NT], th[NT];      // (X3Geo::NDTL transposed tiles)
};

// G-phase op V: exponential `V & 1` of pair V / 2 (pair k <-> row tile k / (2 CT), column tile (k / 2) % CT, half k & 1).
// SPLIT: blocks of 8 ops for two pairs (a, b): add0 a, add1 a, add0 b, add1 b (acc += acc_lo), exp0 a, exp1 a, exp0 b, exp1 b
template <int CT, bool SPLIT, int V, class RG>
__device__ __forceinline__ void x3_gop(RG& r) {
    if constexpr (!SPLIT) {
        constexpr int k = V / 2, which = V % 2;
        constexpr int rt = k / (2 * CT), ct = (k / 2) % CT, h = k & 1;
        asm volatile("v_exp_f32 %0, %1" : "=v"(r.e[k][which]) : "v"(r.acc[rt][ct][2 * h + which]));
    } else {
        constexpr int blk = V / 8, o = V % 8, k = 2 * blk + ((o >> 1) & 1), which = o & 1;
        constexpr int rt = k / (2 * CT), ct = (k / 2) % CT, h = k & 1;
        if constexpr (o < 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r.acc[rt][ct][2 * h + which]) : "v"(r.accl[rt][ct][2 * h + which]));
        else asm volatile("v_exp_f32 %0, %1" : "=v"(r.e[k][which]) : "v"(r.acc[rt][ct][2 * h + which]));
    }
}
// L-phase op V: the split of the numerators into bf16 components, two pairs (a, b) interleaved so that no op reads the result
// of the op right in front of it:  c0 = RNE bf16(e);  c1 = RNE bf16(e - float(c0));  c2 = RNE bf16(e - float(c0) - float(c1))
// (the differences are exact in fp32)
//   NC = 2, block of 12 ops: cvt a, cvt b, shl a, shl b, and a, and b, sub0 a, sub0 b, sub1 a, sub1 b, cvt-c1 a, cvt-c1 b
//   NC = 3, block of 22 ops: ... the same, then shl / and of c1, the two subtractions from the residual, cvt-c2
template <int CT, int NC, int V, class RG>
__device__ __forceinline__ void x3_lop(RG& r) {
    constexpr int OPP = NC == 2 ? 6 : 11;
    constexpr int blk = V / (2 * OPP), j = (V % (2 * OPP)) / 2, ab = V & 1, k = 2 * blk + ab;
    constexpr int rt = k / (2 * CT), ct = (k / 2) % CT, h = k & 1;
    if constexpr (j == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r.w[0][ct][2 * rt + h]) : "v"(r.e[k][0]), "v"(r.e[k][1]));
    else if constexpr (j == 1) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(r.tmp[ab][0]) : "v"(r.w[0][ct][2 * rt + h]));
    else if constexpr (j == 2) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(r.tmp[ab][1]) : "v"(r.w[0][ct][2 * rt + h]));
    else if constexpr (j == 3) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r.tmp[ab][0]) : "v"(r.e[k][0]));
    else if constexpr (j == 4) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r.tmp[ab][1]) : "v"(r.e[k][1]));
    else if constexpr (j == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r.w[1][ct][2 * rt + h]) : "v"(r.tmp[ab][0]), "v"(r.tmp[ab][1]));
    else if constexpr (j == 6) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(r.tm2[ab][0]) : "v"(r.w[1][ct][2 * rt + h]));
    else if constexpr (j == 7) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(r.tm2[ab][1]) : "v"(r.w[1][ct][2 * rt + h]));
    else if constexpr (j == 8) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r.tmp[ab][0]) : "v"(r.tm2[ab][0]));
    else if constexpr (j == 9) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r.tmp[ab][1]) : "v"(r.tm2[ab][1]));
    else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r.w[NC - 1][ct][2 * rt + h]) : "v"(r.tmp[ab][0]), "v"(r.tmp[ab][1]));
}
template <int D, int CT, int NC, int M, int V = X3Geo<D, CT, NC>::gfirst(M), class RG>
__device__ __forceinline__ void x3_gops(RG& r) {
    using XG = X3Geo<D, CT, NC>;
    if constexpr (M < XG::MG && V < XG::gfirst(M + 1)) {
        x3_gop<CT, XG::SPLIT, V>(r);
        x3_gops<D, CT, NC, M, V + 1>(r);
    }
}
template <int D, int CT, int NC, int M, int V = X3Geo<D, CT, NC>::lfirst(M), class RG>
__device__ __forceinline__ void x3_lops(RG& r) {
    using XG = X3Geo<D, CT, NC>;
    if constexpr (M < XG::ML && V < XG::lfirst(M + 1)) {
        x3_lop<CT, NC, V>(r);
        x3_lops<D, CT, NC, M, V + 1>(r);
    }
}
// all ops of one phase back to back (fill slot, drain, fenced slots: no MFMAs to hide them under)
template <int D, int CT, int NC, int V = 0, class RG>
__device__ __forceinline__ void x3_all_gops(RG& r) {
    if constexpr (V < X3Geo<D, CT, NC>::GOPS) {
        x3_gop<CT, X3Geo<D, CT, NC>::SPLIT, V>(r);
        x3_all_gops<D, CT, NC, V + 1>(r);
    }
}
template <int D, int CT, int NC, int V = 0, class RG>
__device__ __forceinline__ void x3_all_lops(RG& r) {
    if constexpr (V < X3Geo<D, CT, NC>::LOPS) {
        x3_lop<CT, NC, V>(r);
        x3_all_lops<D, CT, NC, V + 1>(r);
    }
}

// acc (VGPR) += A . B with the B operand in AGPRs: the c1 / c2 fragments of rx live there (MFMA operands may be ArchVGPRs or
// AccVGPRs on gfx90a and later), which keeps the loop inside the 256 architectural VGPRs an asm operand can live in
template <bool COLD>
__device__ __forceinline__ void mfma_v_ab(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    if constexpr (COLD) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 15" : "+v"(acc) : "v"(a), "a"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
}

template <bool COLD>
__device__ __forceinline__ void mfma_v0_ab(f32x4& acc, const bf16x8& a, const bf16x8& b) {   // acc (VGPR) = A . B, B in AGPRs
    if constexpr (COLD) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0\n\ts_nop 15" : "=&v"(acc) : "v"(a), "a"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));
}

template <bool COLD>
__device__ __forceinline__ void mfma_a0(f32x4& acc, const bf16x8& a, const bf16x8& b) {   // acc (AGPR) = A . B
    if constexpr (COLD) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0\n\ts_nop 15" : "=&a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&a"(acc) : "v"(a), "v"(b));
}

template <int CT, int NC>
__device__ __forceinline__ void x3_pack(const unsigned (&w)[NC][CT][4], bf16x8 (&pb)[NC][CT]) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const u32x4 v = {w[j][ct][0], w[j][ct][1], w[j][ct][2], w[j][ct][3]};
            pb[j][ct] = __builtin_bit_cast(bf16x8, v);
        }
}

// ---- operand lifetimes hipcc cannot know about.  The SIMD issues an MFMA every 8 cycles but the matrix pipe starts one every 16:
// in MFMA-dense stretches (the last part's tiles at the end of G: CT MFMAs per step and little else) MFMAs queue up in front of the
// pipe, and an MFMA reads its A / B operands when it STARTS, not when it issues.  hipcc sees an operand dead right behind its last
// MFMA and hands the register to the next asm result - e.g. the destination of a ds_read issued two MFMAs later, whose data then
// lands (LDS latency ~64+ cycles) BEFORE the queued MFMA has read the old value.  Observed: the first A-fragment request of L(t+1)
// was given the registers of pb[0][0], the B operand of the third-last MFMA of G: column tile 0 NaN (experiments/tools/dbg_x3.py).  An empty
// asm use keeps an operand reserved for at least a whole step (>= 2 CT MFMAs) behind its last MFMA: fragments and tiles two
// steps, the operands of a chain's tail until the next chain's second step.
template <int CT, int NC>
__device__ __forceinline__ void x3_keep_pb(const bf16x8 (&pb)[NC][CT]) {
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) asm volatile("" ::"v"(pb[j][ct]));
}
__device__ __forceinline__ void x3_keep(const bf16x8& a) { asm volatile("" ::"v"(a)); }
__device__ __forceinline__ void x3_keep(const s16x4& a, const s16x4& b) { asm volatile("" ::"v"(a), "v"(b)); }

// A fragment of logits step I (any image): image I / NIL sits one chunk behind image 0 of the same subtile
template <int D, int CT, int NC, int OFF, int I>
__device__ __forceinline__ void x3_a_issue(const unsigned lbase, const int a0, bf16x8& a) {
    using XG = X3Geo<D, CT, NC>;
    pipe_a_issue<XG::DL, OFF + (I / XG::NIL) * XG::CB, I % XG::NIL>(lbase, a0, a);
}
template <int D, int CT, int NC, int OFF, int K>
__device__ __forceinline__ void x3_a_prologue(const unsigned lbase, const int a0, bf16x8 (&af)[X3Geo<D, CT, NC>::NI]) {
    if constexpr (K > 0) {
        x3_a_prologue<D, CT, NC, OFF, K - 1>(lbase, a0, af);
        x3_a_issue<D, CT, NC, OFF, K - 1>(lbase, a0, af[K - 1]);
    }
}
// transposed tile DT (any image)
template <int D, int CT, int NC, int OFF, int DT>
__device__ __forceinline__ void x3_tr_issue(const unsigned lbase, const int t0, s16x4& lo, s16x4& hi) {
    using XG = X3Geo<D, CT, NC>;
    tr_issue<XG::DL, OFF + (DT / XG::NDTI) * XG::CB, DT % XG::NDTI>(lbase, t0, lo, hi);
}
template <int D, int CT, int NC, int OFF, int K>
__device__ __forceinline__ void x3_tr_prologue(const unsigned lbase, const int t0, s16x4 (&tl)[X3Geo<D, CT, NC>::NDTL],
                                               s16x4 (&th)[X3Geo<D, CT, NC>::NDTL]) {
    if constexpr (K > 0) {
        x3_tr_prologue<D, CT, NC, OFF, K - 1>(lbase, t0, tl, th);
        x3_tr_issue<D, CT, NC, OFF, K - 1>(lbase, t0, tl[K - 1], th[K - 1]);
    }
}

// rx fragments: x[j][ct][img KSH + s] multiplies k-step s of every part of image img; component 0 lives in VGPRs, the others in
// AGPRs (MFMA B operands may): NC = 2 at D = 256 pins 200 AGPRs (U 128, row sums 8, c1 64), NC = 3 at D = 128 136 (U 64, 8, 2 x 32)
#define PCVAE_X3_XARGS(D, CT, NC) const bf16x8 (&x)[NC][CT][X3Geo<D, CT, NC>::NIMG * X3Geo<D, CT, NC>::KSH]

// MFMAs POS .. LEN - 1 of logits step I, each followed by what the schedule puts into its gap
template <int D, int CT, int NC, int OFF, int I, int POS, bool HAS_PREV, bool COLD, class RG>
__device__ __forceinline__ void x3_lstep(const unsigned lbase, const int a0, bf16x8 (&af)[X3Geo<D, CT, NC>::NI],
                                         PCVAE_X3_XARGS(D, CT, NC), RG& r) {
    using XG = X3Geo<D, CT, NC>;
    constexpr int img = I / XG::NIL, s = (I % XG::NIL) >> 1, rt = I & 1, part = s / XG::KSH;
    constexpr int LEN = (NC - part) * CT;
    if constexpr (POS < LEN) {
        constexpr int xs = img * XG::KSH + s % XG::KSH;       // rx fragment of this k-step
        constexpr int j = POS / CT, cti = POS % CT;
        constexpr int M0 = XG::lpos(I);
        // operand reservations (x3_keep): an operand stays allocated until >= 4 MFMAs behind its last MFMA - two steps at CT >= 2
        // (every step has >= 2 MFMAs; the statement sits behind the step's second MFMA), four at CT = 1 (a last-part step is ONE MFMA)
        constexpr int KD = CT == 1 ? 4 : 2, KPOS = CT == 1 ? 0 : 1;
        if constexpr (XG::SPLIT) {   // every rx component in AGPRs; component pairs (i, 0) -> acc, the small ones -> accl
            f32x4& dst = (part == 0 && j == 0) ? r.acc[rt][cti] : r.accl[rt][cti];
            if constexpr (s == 0 && img == 0 && j <= 1) mfma_v0_ab<COLD>(dst, af[I], x[j][cti][xs]);
            else mfma_v_ab<COLD>(dst, af[I], x[j][cti][xs]);
        } else if constexpr (j == 0) {
            if constexpr (s == 0 && img == 0) mfma_v0<COLD>(r.acc[rt][cti], af[I], x[0][cti][xs]);
            else mfma_v<COLD>(r.acc[rt][cti], af[I], x[0][cti][xs]);
        } else {
            mfma_v_ab<COLD>(r.acc[rt][cti], af[I], x[j][cti][xs]);
        }
        if constexpr (POS == 0 && I + X3_AD < XG::NI && (COLD || !(X3_PROBE & 2)))
            x3_a_issue<D, CT, NC, OFF, I + X3_AD>(lbase, a0, af[I + X3_AD]);
        if constexpr (POS == KPOS && I >= KD) x3_keep(af[I - KD]);
        if constexpr (POS == KPOS && I == 2 && HAS_PREV) {   // the tail operands of the gradient chain in front of this L
            x3_keep_pb<CT, NC>(r.pb);
            x3_keep(r.tl[XG::NDTL - 1], r.th[XG::NDTL - 1]);
            x3_keep(r.tl[XG::NDTL - 2], r.th[XG::NDTL - 2]);
            x3_keep(r.tl[XG::NDTL - 3], r.th[XG::NDTL - 3]);
            if constexpr (KD > 3) x3_keep(r.tl[XG::NDTL - 4], r.th[XG::NDTL - 4]);
        }
        if constexpr (HAS_PREV && !COLD && !(X3_PROBE & 1)) x3_lops<D, CT, NC, M0 + POS>(r);
        x3_lstep<D, CT, NC, OFF, I, POS + 1, HAS_PREV, COLD>(lbase, a0, af, x, r);
    }
}

// L(t): logits chain into r.acc with the split of the PREVIOUS subtile's numerators (r.e -> r.w) in its gaps.
// A fragment I + X3_AD is requested behind the first MFMA of step I, so X3_AD - 1 younger fragments are in flight at step I's wait.
template <int D, int CT, int NC, int OFF, int I, bool HAS_PREV, bool COLD, class RG>
__device__ __forceinline__ void x3_logits(const unsigned lbase, const int a0, bf16x8 (&af)[X3Geo<D, CT, NC>::NI],
                                          PCVAE_X3_XARGS(D, CT, NC), RG& r) {
    using XG = X3Geo<D, CT, NC>;
    if constexpr (I < XG::NI) {
        if constexpr (COLD || !(X3_PROBE & (2 | 16))) lgkm_wait<(I + X3_AD - 1 < XG::NI ? X3_AD - 1 : XG::NI - 1 - I)>();
        x3_lstep<D, CT, NC, OFF, I, 0, HAS_PREV, COLD>(lbase, a0, af, x, r);
        x3_logits<D, CT, NC, OFF, I + 1, HAS_PREV, COLD>(lbase, a0, af, x, r);
    }
}

struct X3Seam {             // all wave-uniform; one entry per image = per seam of a slot
    const uint16_t* E[2];   // image the chunk to request belongs to
    int64_t n_stage[2];     // first item of that chunk
    char* stage_buf[2];     // ring buffer it goes to (fenced slots: x3_stage)
    unsigned stage_lds[2];  // the same buffer as an LDS byte address (steady-state seams: x3_pipe_stage)
    unsigned next_lbase;    // LDS address (minus the immediate) of the NEXT slot's subtile (image 0)
};

// ---- LDS-DMA staging of one ring chunk (32 rows of NC 256-byte parts = 8 NC pieces of 1 KiB; a wave stages PPW = 2 NC of them).
// A piece is four consecutive parts; lane l writes 16-byte position l & 15 of part l >> 4 of its piece, whose SOURCE chunk is that
// position XOR-swizzled by the part's row (the swizzle of the D = 128 bf16 kernel inside every part; NC = 2: exactly
// fast_lane_off<256> / fast_stage<256> / pipe_stage<256>).
template <int NC>
__device__ __forceinline__ void x3_lane_off(const int lane, const int wave, int (&lane_off)[2 * NC]) {
#pragma unroll
    for (int i = 0; i < 2 * NC; ++i) {
        const int part = (wave * 2 * NC + i) * 4 + (lane >> 4);   // part of the chunk this lane writes
        const int row = part / NC;
        lane_off[i] = (lane >> 4) * 256 + (fswz<256>(row & 15, lane & 15) << 4);
    }
}
template <int NC>
__device__ __forceinline__ void x3_stage(const uint16_t* __restrict__ E, const int64_t n0, char* buf, const int wave_u,
                                         const int (&lane_off)[2 * NC]) {
#pragma unroll
    for (int i = 0; i < 2 * NC; ++i) {
        const int pc = wave_u * 2 * NC + i;
        const char* base = reinterpret_cast<const char*>(E) + n0 * (256 * NC) + pc * 1024;  // wave-uniform (SGPR pair)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(base + (uint64_t)(uint32_t)lane_off[i]),
            (__attribute__((address_space(3))) void*)(buf + pc * 1024), 16, 0, 0);
    }
}
// the same pieces of a wave as ONE asm statement for the steady-state seams (see pipe_stage: scalar base + 32-bit lane offset, the
// piece stride as the instruction's immediate, M0 written once)
template <int NC>
__device__ __forceinline__ void x3_pipe_stage(const uint16_t* __restrict__ E, const int64_t n0, const unsigned lds_dst,
                                              const int wave_u, const int (&lane_off)[2 * NC]) {
    const char* base = reinterpret_cast<const char*>(E) + n0 * (256 * NC) + (int64_t)wave_u * (2 * NC * 1024);
    const unsigned m0v = lds_dst + (unsigned)wave_u * (unsigned)(2 * NC * 1024);
    if constexpr (NC == 2)
        asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %4\n\t"
                     "global_load_lds_dwordx4 %1, %4 offset:1024\n\t"
                     "global_load_lds_dwordx4 %2, %4 offset:2048\n\t"
                     "global_load_lds_dwordx4 %3, %4 offset:3072"
                     ::"v"(lane_off[0]), "v"(lane_off[1]), "v"(lane_off[2]), "v"(lane_off[3]), "s"(base), "s"(m0v)
                     : "memory", "m0");
    else   // the immediate is 13 bits signed: pieces 4, 5 go out from a second base (global and M0) 4 KiB further
        asm volatile("s_mov_b32 m0, %8\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %6\n\t"
                     "global_load_lds_dwordx4 %1, %6 offset:1024\n\t"
                     "global_load_lds_dwordx4 %2, %6 offset:2048\n\t"
                     "global_load_lds_dwordx4 %3, %6 offset:3072\n\t"
                     "s_mov_b32 m0, %9\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %4, %7\n\t"
                     "global_load_lds_dwordx4 %5, %7 offset:1024"
                     ::"v"(lane_off[0]), "v"(lane_off[1]), "v"(lane_off[2]), "v"(lane_off[3]), "v"(lane_off[4]), "v"(lane_off[5]),
                       "s"(base), "s"(base + 4096), "s"(m0v), "s"(m0v + 4096u)
                     : "memory", "m0");
}
// piece I of a wave's 2 NC, alone (X3_DMA_SPREAD: one piece per gradient step behind the seam instead of a burst of 2 NC at it)
template <int NC, int I>
__device__ __forceinline__ void x3_pipe_stage_piece(const uint16_t* __restrict__ E, const int64_t n0, const unsigned lds_dst,
                                                    const int wave_u, const int (&lane_off)[2 * NC]) {
    constexpr int HI = I / 4, LO = I % 4;   // the immediate is 13 bits signed: pieces 4, 5 from a base 4 KiB further
    const char* base = reinterpret_cast<const char*>(E) + n0 * (256 * NC) + (int64_t)wave_u * (2 * NC * 1024) + HI * 4096;
    const unsigned m0v = lds_dst + (unsigned)wave_u * (unsigned)(2 * NC * 1024) + (unsigned)(HI * 4096);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3"
                 ::"v"(lane_off[I]), "s"(base), "s"(m0v), "n"(LO * 1024) : "memory", "m0");
}
// pieces I0 .. I0 + X3_DMA_PPS - 1
template <int NC, int I0, int K = 0>
__device__ __forceinline__ void x3_pipe_stage_group(const uint16_t* __restrict__ E, const int64_t n0, const unsigned lds_dst,
                                                    const int wave_u, const int (&lane_off)[2 * NC]) {
    if constexpr (K < X3_DMA_PPS && I0 + K < 2 * NC) {
        x3_pipe_stage_piece<NC, I0 + K>(E, n0, lds_dst, wave_u, lane_off);
        x3_pipe_stage_group<NC, I0, K + 1>(E, n0, lds_dst, wave_u, lane_off);
    }
}
// ragged tail: up to 128 items staged synchronously with clamped rows (same image: row * RB + part * 256, swizzled chunks)
template <int NC>
__device__ __forceinline__ void x3_stage_tail(const uint16_t* __restrict__ E, const int64_t N, const int64_t n0, char* buf) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PIECES = 128 * 256 * NC / 1024;
#pragma unroll
    for (int i = 0; i < PIECES / 4; ++i) {
        const int pc = wave * (PIECES / 4) + i;
        const int part = pc * 4 + (lane >> 4), row = part / NC, comp = part % NC;
        int64_t n = n0 + row;
        n = n < N ? n : N - 1;
        const char* src = reinterpret_cast<const char*>(E) + n * (256 * NC) + comp * 256 + (fswz<256>(row & 15, lane & 15) << 4);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(buf + pc * 1024), 16, 0, 0);
    }
}

// MFMAs POS .. LEN - 1 of gradient step DT
template <int D, int CT, int NC, int OFFG, int DT, int POS, bool HAS_G, bool COLD, class RG>
__device__ __forceinline__ void x3_gstep(const unsigned lbase_g, const int t0, s16x4 (&tl)[X3Geo<D, CT, NC>::NDTL],
                                         s16x4 (&th)[X3Geo<D, CT, NC>::NDTL], const bf16x8& a, const bf16x8 (&pb)[NC][CT], RG& r,
                                         f32x4 (&U)[X3Geo<D, CT, NC>::NDT][CT], bf16x8 (&af)[X3Geo<D, CT, NC>::NI]) {
    using XG = X3Geo<D, CT, NC>;
    constexpr int NDTL = XG::NDTL, img = DT / XG::NDTI, d = DT % XG::NDTI, part = d / 8;
    constexpr int LEN = (NC - part) * CT;
    if constexpr (POS < LEN) {
        constexpr int M0 = XG::gpos(DT);
        constexpr int UD = img * 8 + d % 8;
        constexpr int KD = CT == 1 ? 4 : 2, KPOS = CT == 1 ? 0 : 1;   // operand reservations: see x3_lstep
        if constexpr (HAS_G) mfma_a<COLD>(U[UD][POS % CT], a, pb[POS / CT][POS % CT]);
        if constexpr (HAS_G && POS == 0 && DT + X3_TD < NDTL && (COLD || !(X3_PROBE & 4)))
            x3_tr_issue<D, CT, NC, OFFG, DT + X3_TD>(lbase_g, t0, tl[DT + X3_TD], th[DT + X3_TD]);
        if constexpr (HAS_G && POS == KPOS && DT >= KD) x3_keep(tl[DT - KD], th[DT - KD]);
        if constexpr (HAS_G && POS == 1 && DT == 1) {   // the last fragments of the logits chain in front of this G
            x3_keep(af[XG::NI - 1]); x3_keep(af[XG::NI - 2]); x3_keep(af[XG::NI - 3]);
            if constexpr (KD > 3) x3_keep(af[XG::NI - 4]);
        }
        if constexpr (!COLD && !(X3_PROBE & 1)) x3_gops<D, CT, NC, M0 + POS>(r);
        x3_gstep<D, CT, NC, OFFG, DT, POS + 1, HAS_G, COLD>(lbase_g, t0, tl, th, a, pb, r, U, af);
    }
}

// G(t-1): gradient chain of the previous subtile (numerators pb) with the exponentials of subtile t (r.acc -> r.e) in its
// gaps; in the middle of EACH image's tiles a seam (counted wait + barrier + refill of one ring chunk), and behind the last seam
// the first A fragments of the next slot.  The transposed reads of tile DT + X3_TD are requested behind the first MFMA of step
// DT, so 2 (X3_TD - 1) younger reads are in flight at step DT's wait.
template <int D, int CT, int NC, int OFFG, int OFFL_NEXT, int DT, bool HAS_G, int VM, bool COLD, class RG>
__device__ __forceinline__ void x3_grad(const unsigned lbase_g, const int t0, s16x4 (&tl)[X3Geo<D, CT, NC>::NDTL],
                                        s16x4 (&th)[X3Geo<D, CT, NC>::NDTL], const bf16x8 (&pb)[NC][CT],
                                        RG& r, f32x4 (&U)[X3Geo<D, CT, NC>::NDT][CT],
                                        const X3Seam& sm, const int wave_u, const int (&lane_off)[2 * NC], const int a0,
                                        bf16x8 (&af)[X3Geo<D, CT, NC>::NI]) {
    using XG = X3Geo<D, CT, NC>;
    constexpr int NDTL = XG::NDTL, NDTI = XG::NDTI;
    constexpr int SEAM_LAST = (XG::NIMG - 1) * NDTI + NDTI / 2;     // the seam that also issues the next slot's first A fragments
    if constexpr (DT < NDTL) {
        constexpr int img = DT / NDTI, d = DT % NDTI;
        if constexpr (d == NDTI / 2) {
            if constexpr (COLD) {
                pipe_fence();
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                if (sm.n_stage[img] >= 0) x3_stage<NC>(sm.E[img], sm.n_stage[img], sm.stage_buf[img], wave_u, lane_off);
                pipe_fence();
            } else if constexpr (!(X3_PROBE & 8)) {
                if constexpr (X3_PROBE & 32) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");   // probe: no barrier
                else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(VM) : "memory");
                if constexpr (!(X3_PROBE & 64)) {   // probe: no refill
                    if constexpr (X3_DMA_SPREAD) x3_pipe_stage_group<NC, 0>(sm.E[img], sm.n_stage[img], sm.stage_lds[img], wave_u, lane_off);
                    else x3_pipe_stage<NC>(sm.E[img], sm.n_stage[img], sm.stage_lds[img], wave_u, lane_off);
                }
            }
            if constexpr (DT == SEAM_LAST && (COLD || !(X3_PROBE & 2)))
                x3_a_prologue<D, CT, NC, OFFL_NEXT, X3_AD>(sm.next_lbase, a0, af);   // first A fragments of the next slot
        }
        // the other pieces of the seam's refill, one in front of each following step (all of them are out before the next seam, so
        // the counted vmcnt there sees the same queue as with the burst)
        if constexpr (!COLD && X3_DMA_SPREAD && d > NDTI / 2 && (d - NDTI / 2) % X3_DMA_SPREAD == 0 &&
                      (d - NDTI / 2) / X3_DMA_SPREAD * X3_DMA_PPS < 2 * NC && !(X3_PROBE & (8 | 64)))
            x3_pipe_stage_group<NC, (d - NDTI / 2) / X3_DMA_SPREAD * X3_DMA_PPS>(sm.E[img], sm.n_stage[img], sm.stage_lds[img], wave_u, lane_off);
        if constexpr (HAS_G) {
            // younger than tile DT's reads: tiles DT + 1 .. DT + X3_TD - 1, and the next slot's first A fragments once they are out
            constexpr int young = (DT + X3_TD - 1 < NDTL ? X3_TD - 1 : NDTL - 1 - DT);
            constexpr int extra = (DT >= SEAM_LAST && DT - SEAM_LAST < X3_TD) ? X3_AD : 0;   // issued behind a read that is still awaited
            if constexpr (COLD || !(X3_PROBE & (2 | 4 | 16))) lgkm_wait<2 * young + extra>();
        }
        const s16x8 a16 = __builtin_shufflevector(tl[DT], th[DT], 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 a = __builtin_bit_cast(bf16x8, a16);
        x3_gstep<D, CT, NC, OFFG, DT, 0, HAS_G, COLD>(lbase_g, t0, tl, th, a, pb, r, U, af);
        x3_grad<D, CT, NC, OFFG, OFFL_NEXT, DT + 1, HAS_G, VM, COLD>(lbase_g, t0, tl, th, pb, r, U, sm, wave_u, lane_off, a0, af);
    } else {
        if constexpr (HAS_G) x3_keep_pb<CT, NC>(pb);   // (and on into the next logits chain: x3_lstep, step 2)
        if constexpr (COLD) pipe_fence();
    }
}

// row-sum MFMAs POS .. NC CT - 1.  RESTART: the accumulator begins a new chunk (the first MFMA of every column tile takes C = 0)
template <int CT, int NC, int POS, bool COLD, bool RESTART = false>
__device__ __forceinline__ void x3_ones(f32x4 (&lsum)[CT], const bf16x8& ones, const bf16x8 (&pb)[NC][CT]) {
    if constexpr (POS < NC * CT) {
        if constexpr (RESTART && POS < CT) mfma_a0<COLD>(lsum[POS % CT], ones, pb[POS / CT][POS % CT]);
        else mfma_a<COLD>(lsum[POS % CT], ones, pb[POS / CT][POS % CT]);
        x3_ones<CT, NC, POS + 1, COLD, RESTART>(lsum, ones, pb);
    }
}

// one slot: L(t) || split of subtile t-1's numerators, then G(t-1) || exponentials of subtile t
template <int D, int CT, int NC, int OFFL, int OFFG, int OFFL_NEXT, bool HAS_G, int VM, bool COLD, bool RESTART = false, class RG>
__device__ __forceinline__ void x3_slot(const unsigned lbase_l, const unsigned lbase_g, const FastLane& L,
                                        PCVAE_X3_XARGS(D, CT, NC),
                                        bf16x8 (&af)[X3Geo<D, CT, NC>::NI], RG& r, f32x4 (&U)[X3Geo<D, CT, NC>::NDT][CT],
                                        f32x4 (&lsum)[CT], const X3Seam& sm, const int wave_u, const int (&lane_off)[2 * NC]) {
    if constexpr (COLD) pipe_fence();
    x3_logits<D, CT, NC, OFFL, 0, HAS_G, COLD>(lbase_l, L.a0, af, x, r);
    if constexpr (COLD && HAS_G) {      // fenced slots: nothing overlapped - the split of the previous subtile's numerators now
        pipe_fence();
        x3_all_lops<D, CT, NC>(r);
        asm volatile("s_nop 1" ::: "memory");
    }
    x3_pack<CT, NC>(r.w, r.pb);
    if constexpr (HAS_G) {
        if constexpr (COLD || !(X3_PROBE & 4)) x3_tr_prologue<D, CT, NC, OFFG, X3_TD>(lbase_g, L.t0, r.tl, r.th);
        x3_ones<CT, NC, 0, COLD, RESTART>(lsum, L.ones, r.pb);
    }
    if constexpr (COLD) {               // exponentials of this subtile, nothing overlapped
        pipe_fence();
        x3_all_gops<D, CT, NC>(r);
        asm volatile("s_nop 1" ::: "memory");
    }
    x3_grad<D, CT, NC, OFFG, OFFL_NEXT, 0, HAS_G, VM, COLD>(lbase_g, L.t0, r.tl, r.th, r.pb, r, U, sm, wave_u, lane_off, L.a0, af);
}

// ---- fenced gradient of one subtile (drain, ragged tail): nothing overlapped.  `istride`: bytes from image 0 to image 1 of the
// subtile (the ring: one chunk; the synchronously staged tail: its own image area)
template <int D, int CT, int NC, int DT = 0>
__device__ __forceinline__ void x3_cold_grad(const unsigned lbase_g, const unsigned istride, const int t0,
                                             s16x4 (&tl)[X3Geo<D, CT, NC>::NDTL], s16x4 (&th)[X3Geo<D, CT, NC>::NDTL],
                                             const bf16x8 (&pb)[NC][CT], f32x4 (&U)[X3Geo<D, CT, NC>::NDT][CT]) {
    using XG = X3Geo<D, CT, NC>;
    if constexpr (DT < XG::NDTL) {
        if constexpr (DT + 2 < XG::NDTL)
            tr_issue<XG::DL, 0, (DT + 2) % XG::NDTI>(lbase_g + ((DT + 2) / XG::NDTI) * istride, t0, tl[DT + 2], th[DT + 2]);
        lgkm_wait<2 * ((DT + 2 < XG::NDTL ? DT + 2 : XG::NDTL - 1) - DT)>();
        const s16x8 a16 = __builtin_shufflevector(tl[DT], th[DT], 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 a = __builtin_bit_cast(bf16x8, a16);
        constexpr int UD = (DT / XG::NDTI) * 8 + (DT % XG::NDTI) % 8, part = (DT % XG::NDTI) / 8;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int j = 0; j < NC - part; ++j) mfma_a<true>(U[UD][ct], a, pb[j][ct]);
        x3_cold_grad<D, CT, NC, DT + 1>(lbase_g, istride, t0, tl, th, pb, U);
    }
}
template <int D, int CT, int NC>
__device__ __forceinline__ void x3_cold_gradient(const unsigned lbase_g, const unsigned istride, const FastLane& L,
                                                 const bf16x8 (&pb)[NC][CT],
                                                 f32x4 (&U)[X3Geo<D, CT, NC>::NDT][CT], f32x4 (&lsum)[CT]) {
    using XG = X3Geo<D, CT, NC>;
    s16x4 tl[XG::NDTL], th[XG::NDTL];
    pipe_fence();
    tr_issue<XG::DL, 0, 0>(lbase_g, L.t0, tl[0], th[0]);
    tr_issue<XG::DL, 0, 1>(lbase_g, L.t0, tl[1], th[1]);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int j = 0; j < NC; ++j) mfma_a<true>(lsum[ct], L.ones, pb[j][ct]);
    x3_cold_grad<D, CT, NC>(lbase_g, istride, L.t0, tl, th, pb, U);
    pipe_fence();
}

// one subtile on its own (ragged tail): logits, bound check, numerators, gradient - nothing overlapped, every MFMA fenced
template <int D, int CT, int NC, int I = 0>
__device__ __forceinline__ void x3_cold_logits(const unsigned lbase, const unsigned istride, const int a0,
                                               bf16x8 (&af)[X3Geo<D, CT, NC>::NI], PCVAE_X3_XARGS(D, CT, NC), f32x4 (&acc)[2][CT],
                                               f32x4 (&accl)[2][CT]) {
    using XG = X3Geo<D, CT, NC>;
    if constexpr (I < XG::NI) {
        constexpr int img = I / XG::NIL, i = I % XG::NIL;
        pipe_a_issue<XG::DL, 0, i>(lbase + img * istride, a0, af[I]);
        lgkm_wait<0>();
        constexpr int s = i >> 1, rt = i & 1, xs = img * XG::KSH + s % XG::KSH, part = s / XG::KSH;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            if constexpr (XG::SPLIT) {   // c0 c0 -> acc, every other component pair -> accl
                if constexpr (s == 0 && img == 0) mfma_v0_ab<true>(acc[rt][ct], af[I], x[0][ct][xs]);
                else if constexpr (part == 0) mfma_v_ab<true>(acc[rt][ct], af[I], x[0][ct][xs]);
                else mfma_v_ab<true>(accl[rt][ct], af[I], x[0][ct][xs]);
#pragma unroll
                for (int j = 1; j < NC - part; ++j) {
                    if (s == 0 && img == 0 && j == 1) mfma_v0_ab<true>(accl[rt][ct], af[I], x[j][ct][xs]);
                    else mfma_v_ab<true>(accl[rt][ct], af[I], x[j][ct][xs]);
                }
            } else {
                if constexpr (s == 0 && img == 0) mfma_v0<true>(acc[rt][ct], af[I], x[0][ct][xs]);
                else mfma_v<true>(acc[rt][ct], af[I], x[0][ct][xs]);
#pragma unroll
                for (int j = 1; j < NC - part; ++j) mfma_v_ab<true>(acc[rt][ct], af[I], x[j][ct][xs]);
            }
        }
        x3_cold_logits<D, CT, NC, I + 1>(lbase, istride, a0, af, x, acc, accl);
    }
}
__device__ __forceinline__ unsigned x3_pack_rne(float a, float b) {   // two fp32 -> packed bf16 pair (RNE), a in the low half
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int D, int CT, int NC>
__device__ __forceinline__ void x3_solo(const unsigned lbase, const unsigned istride, const int64_t n0, const int64_t N, const FastLane& L,
                                        PCVAE_X3_XARGS(D, CT, NC),
                                        f32x4 (&U)[X3Geo<D, CT, NC>::NDT][CT], f32x4 (&lsum)[CT]) {
    using XG = X3Geo<D, CT, NC>;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc[2][CT], accl[2][CT];
    bf16x8 af[XG::NI];
    pipe_fence();
    x3_cold_logits<D, CT, NC>(lbase, istride, L.a0, af, x, acc, accl);
    pipe_fence();
    if constexpr (XG::SPLIT) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[rt][ct] += accl[rt][ct];
    }
    bf16x8 pb[NC][CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        u32x4 w[NC];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = 2 * h;
                const bool ok0 = n0 + 16 * rt + 4 * L.g + i < N, ok1 = n0 + 16 * rt + 4 * L.g + i + 1 < N;
                float e0 = ok0 ? __builtin_amdgcn_exp2f(acc[rt][ct][i]) : 0.f;
                float e1 = ok1 ? __builtin_amdgcn_exp2f(acc[rt][ct][i + 1]) : 0.f;
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    const unsigned c = x3_pack_rne(e0, e1);
                    w[j][2 * rt + h] = c;
                    e0 -= __uint_as_float(c << 16);
                    e1 -= __uint_as_float(c & 0xffff0000u);
                }
            }
#pragma unroll
        for (int j = 0; j < NC; ++j) pb[j][ct] = __builtin_bit_cast(bf16x8, w[j]);
    }
    asm volatile("s_nop 1" ::: "memory");
    x3_cold_gradient<D, CT, NC>(lbase, istride, L, pb, U, lsum);
}

// rx row fragments: x_0 = RNE bf16(rx * log2 e), x_1 = RNE bf16(rx * log2 e - x_0), ..., laid out like the bf16 kernels' B operand;
// fragment img * 4 + s multiplies k-step s of every part of image img (dims 128 img ..)
template <int D, int NC>
__device__ __forceinline__ void x3_load_x(const float* __restrict__ rx, const int64_t row, const int g, bf16x8 (&x)[NC][D / 32]) {
#pragma unroll
    for (int f = 0; f < D / 32; ++f) {
        const int img = f >> 2, s = f & 3;
        const int c0 = img * 128 + 8 * fchunk<256>(s, g);   // part-0 chunk of k-step s (the k-steps s + 4 i of the other parts multiply the same columns)
        const float4 v0 = *reinterpret_cast<const float4*>(rx + row * D + c0);
        const float4 v1 = *reinterpret_cast<const float4*>(rx + row * D + c0 + 4);
        float v[8] = {v0.x * kLog2e, v0.y * kLog2e, v0.z * kLog2e, v0.w * kLog2e,
                      v1.x * kLog2e, v1.y * kLog2e, v1.z * kLog2e, v1.w * kLog2e};
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const __bf16 h = (__bf16)v[j];
                x[c][f][j] = h;
                v[j] -= (float)h;
            }
    }
}

template <int D, int CT, int NC>
__global__ void __launch_bounds__(256, 1) catalog_ce_x3_pipe_kernel(CatParamsB p) {
    using XG = X3Geo<D, CT, NC>;
    using GL = typename XG::GL;
    constexpr int CB = XG::CB, ROWS = XG::ROWS, TR = XG::TR, NB = XG::NB, PF = XG::PF, DL = XG::DL, NIMG = XG::NIMG, PPW = XG::PPW;
    constexpr int NX = NIMG * XG::KSH;                                   // rx fragments per column tile and component
    static_assert(PPW == 2 * NC, "a wave stages 2 NC pieces of a chunk");
    static_assert(XG::NDTI / 2 + (2 * NC - 1) / X3_DMA_PPS * X3_DMA_SPREAD < XG::NDTI, "the spread refill is out before the gradient chain ends");
    static_assert(XG::gfirst(XG::MG) == XG::GOPS && XG::gfirst(XG::MG - 1) == XG::GOPS, "every exponential has a gap");
    static_assert(XG::lfirst(XG::ML - 2) == XG::LOPS, "every split op has a gap, the last two gaps of L stay free");
    static_assert(XG::gfirst(XG::NRS + 1) == 0, "no exponential before the row-sum MFMAs are out: the accumulators are fresh");
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
    int T = (int)min((int64_t)(t_end - t_beg), (p.N - nbase) / XG::BNF);    // full 32-item subtiles = slots (NIMG ring chunks each)
    T = max(T, 0);
    // ring chunk q = (subtile q / NIMG, image q % NIMG); image i of the table starts i * N rows behind image 0
    auto img_of = [&](int q) { return p.E + (int64_t)(q % NIMG) * p.N * DL; };
    auto item_of = [&](int q) { return nbase + (int64_t)min(q / NIMG, T - 1) * XG::BNF; };   // beyond the end: the last subtile again

    const int64_t rw = (int64_t)rb * ROWS + wave * 16 * CT;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int lane_off[PPW];
    x3_lane_off<NC>(lane, wave, lane_off);
#pragma unroll
    for (int q = 0; q < PF + NIMG; ++q)   // a constant number of chunks in flight from here on
        if (T > 0) x3_stage<NC>(img_of(q), item_of(q), smem + q * CB, wave_u, lane_off);

    bf16x8 x[NC][CT][NX];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int64_t r = rw + 16 * ct + c;
        bf16x8 xr[NC][D / 32];
        x3_load_x<D, NC>(p.rx, r < p.R ? r : p.R - 1, g, xr);
#pragma unroll
        for (int j = 0; j < NC; ++j)
#pragma unroll
            for (int s = 0; s < NX; ++s) {
                x[j][ct][s] = xr[j][s];
                if (j > 0 || XG::SPLIT) asm volatile("" : "+a"(x[j][ct][s]));   // components 1.. (SPLIT: all) live in AGPRs (mfma_v_ab)
            }
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
    float ltot[CT];   // (CHUNK) the row sums of the finished chunks
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) ltot[ct] = 0.f;

    X3Regs<CT, XG::NDTL, NC> r;
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
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PF * PPW) : "memory");   // the chunks of subtile 0 landed
        x3_a_prologue<D, CT, NC, 0, X3_AD>(lds0, L.a0, af);
        {   // slot 0: nothing to drain yet
            X3Seam sm = seam_of(0);
            sm.next_lbase = lds_of(T > 1 ? 1 : 0);
            x3_slot<D, CT, NC, 0, 0, 0, false, 0, true>(lds0, lds0, L, x, af, r, U, lsum, sm, wave_u, lane_off);
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
            if constexpr (XG::CHUNK) {   // the chunk the last trip summed joins the total; this trip's first row-sum MFMA restarts lsum
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) ltot[ct] += lsum[ct][0];
            }
#define PCVAE_X3S(UU)                                                                                                     \
            {                                                                                                             \
                constexpr int TL = 1 + UU, TG = UU, TN = 2 + UU;                                                          \
                constexpr int OL = ((NIMG * TL) % NB) * CB, OG = ((NIMG * TG) % NB) * CB, ON = ((NIMG * TN) % NB) * CB;   \
                X3Seam s2 = seam_of(t + UU);                                                                              \
                s2.stage_lds[0] = lds0 + ((NIMG * TL + NIMG + PF) % NB) * CB;   /* t = 1 (mod TR): constants */           \
                if constexpr (NIMG > 1) s2.stage_lds[NIMG - 1] = lds0 + ((NIMG * TL + NIMG - 1 + NIMG + PF) % NB) * CB;   \
                x3_slot<D, CT, NC, OL, OG, ON, true, (PF - 1) * PPW, false, XG::CHUNK && UU == 0>(lds0, lds0, L, x, af, r, U, lsum, s2, wave_u, lane_off); \
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
            x3_slot<D, CT, NC, 0, 0, 0, true, 0, true>(lds_of(t), lds_of(t - 1), L, x, af, r, U, lsum, s2, wave_u, lane_off);
        }
        {   // drain: the split of the last subtile's numerators, then its gradient chain
            bf16x8 pb[NC][CT];
            pipe_fence();
            x3_all_lops<D, CT, NC>(r);
            x3_pack<CT, NC>(r.w, pb);
            asm volatile("s_nop 1" ::: "memory");
            x3_cold_gradient<D, CT, NC>(lds_of(T - 1), (unsigned)CB, L, pb, U, lsum);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive its wave
    // ---- tail: the ragged last subtile of the catalog (and ranges shorter than one chunk), staged synchronously: up to 4 subtiles
    // of image 0 at smem, of image 1 behind them
    constexpr unsigned TAIL_IMG = 128 * XG::RB;   // 64 / 96 KB
    static_assert(NIMG * TAIL_IMG <= (unsigned)(NB * CB), "the tail images fit the ring's LDS");
    for (int tt = t_beg + T; tt < t_end; tt += 4) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NIMG; ++i) x3_stage_tail<NC>(p.E + (int64_t)i * p.N * DL, p.N, (int64_t)tt * 32, smem + i * TAIL_IMG);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int nsub = min(4, t_end - tt);
        for (int st = 0; st < nsub; ++st) x3_solo<D, CT, NC>(lds0 + st * GL::ST, TAIL_IMG, (int64_t)(tt + st) * 32, p.N, L, x, U, lsum);
    }
    pipe_fence();
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const float l = XG::CHUNK ? ltot[ct] + lsum[ct][0] : lsum[ct][0];
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
    // ln L = e ln2 + ln m with L = m 2^e, m in [0.5, 1): log2f(L) * ln2 rounds log2 L at ITS magnitude (1 ulp of ~17 is 1.9e-6, and
    // v_log_f32 is a 1-ulp instruction), which was the largest single term of this arithmetic's lse error.  e * kLn2Hi is exact
    // (kLn2Hi has 9 trailing zero bits, |e| < 256); the rest is two fmaf, so the result carries one rounding of lse + ~1e-7.
    int le;
    const float lm = frexpf(L, &le);
    constexpr float kLn2Hi = 0.693145751953125f, kLn2Lo = 1.42860682e-06f;
    const float lse_r = fmaf((float)le, kLn2Hi, fmaf((float)le, kLn2Lo, logf(lm)));
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
