// global_pk.h -- the banded global alignment (ksw_global2's recurrence) with the band across the lanes in packed 16-bit halves.
// Used by k_cigar.hip (with the direction matrix, for the traceback) and by post_common.h's gen_cigar (score only: the
// alignments mem_patch_reg asks for when a wavefront works on one long read).  Included from post_common.h after SeqAcc.
#pragma once
#include "pk16.h"
#include "wave_ops.h"

// the direction matrix lives in LDS (small jobs) or in global memory (a slab / the pool): the code says which (AS_LDS / AS_GLOBAL)
DEV void z_put(uint8_t* p, uint8_t v, bool z_lds) { if (z_lds) *AS_LDS(uint8_t, p) = v; else *AS_GLOBAL(uint8_t, p) = v; }
DEV int z_get(const uint8_t* p, bool z_lds) { return z_lds ? *AS_LDS(const uint8_t, p) : *AS_GLOBAL(const uint8_t, p); }

// The diagonal form in packed 16-bit halves (pk16.h): pair p of NP holds chunk p in the low halves and chunk p + NP in the high
// halves of its registers (slot s = 64 c + lane owns column i - w + s of row i, as above), so one stream of packed instructions
// computes two chunks -- bands of up to 128 NP columns.  What makes 16 bits enough for any read length:
//  * values are kept relative to a base that moves with the row maximum: every 64 rows the maximum of the live H values is
//    brought back to 0 (the recurrence only ever compares and adds constants, so a common offset changes no decision; the score
//    gets the base back at the end).  Between two such points values move by at most 64 x (largest score or penalty).
//  * the same step checks that the live values span less than `range`; if they ever do not, the function gives up (ok = false,
//    nothing but the scratch matrix touched) and the caller runs the 32-bit form.  In a band every cell is within
//    (a + e_del + e_ins) 2w + o_del + o_ins of its row's maximum, so for bwa's options this does not happen.
//  * minus infinity is a sentinel below every live value (SENT): it only ever meets live values in a maximum or a comparison
//    (the cell left of the band's first, the E above the band's last), and is regenerated every row, never accumulated.
//  * upstream's direction bits come from the same maxima: "e > m" is max(m, e) != m, and so on (differences of two live values
//    fit 16 bits by the range check).
// The max-plus prefix runs on values with the sign bit flipped (unsigned order, identity 0 = what DPP leaves in lanes without a
// source); the carries between chunks are scalars.  Scores come from one byte permute per pair (the five scores of the row's
// target base as bytes of a scalar pair, the query codes kept as selectors).
#define GPK_SENT (-32000)
struct GpkFit { int range; };
DEV bool gpk_fit(const MemOpt& opt, int w, int n_pairs, GpkFit& F)
{
    int lo = 127, hi = -128;
#pragma unroll
    for (int a = 0; a < 25; ++a) { lo = opt.mat[a] < lo ? opt.mat[a] : lo; hi = opt.mat[a] > hi ? opt.mat[a] : hi; }
    if (opt.e_ins < 0 || opt.e_del < 0 || opt.o_ins < 0 || opt.o_del < 0 || hi < 0) return false;
    const int maxpen = lo < 0 ? -lo : 0, oe_d = opt.o_del + opt.e_del, oe_i = opt.o_ins + opt.e_ins, oe = oe_d > oe_i ? oe_d : oe_i;
    int P = maxpen > oe ? maxpen : oe; P = P > hi ? P : hi;
    if (oe > 2000 || P > 15 || opt.e_ins > 30 || opt.e_del > 30) return false;      // (64 P: how far a sentinel or a live value can move between two re-basings)
    const long long margin = 64ll * P + 2ll * oe + maxpen + 128ll * n_pairs * opt.e_ins + 64;
    const long long range = 29000 - margin;
    // the first rows: the boundary column -(o_del + e_del (i + 1)) against a maximum of at most hi (i + 1), the first row -(o_ins + e_ins j)
    if (range < 4000 || (long long)(hi + opt.e_del) * (w + 2) + opt.o_del > range || (long long)opt.o_ins + (long long)opt.e_ins * (w + 1) > range) return false;
    F.range = (int)range;
    return true;
}
#define GPK_DPP0(v, ctrl, rowmask) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rowmask), 0xf, (rowmask) == 0xf))
DEV uint32_t gpk_scan_maxu(uint32_t v)                          // inclusive prefix maximum of both unsigned halves over the 64 lanes
{
    v = pk_maxu(v, GPK_DPP0(v, DPP_ROW_SHR(1), 0xf));
    v = pk_maxu(v, GPK_DPP0(v, DPP_ROW_SHR(2), 0xf));
    v = pk_maxu(v, GPK_DPP0(v, DPP_ROW_SHR(4), 0xf));
    v = pk_maxu(v, GPK_DPP0(v, DPP_ROW_SHR(8), 0xf));
    v = pk_maxu(v, GPK_DPP0(v, DPP_ROW_BCAST15, 0xa));
    v = pk_maxu(v, GPK_DPP0(v, DPP_ROW_BCAST31, 0xc));
    return v;
}
// WITH_Z = false: score only (no direction bytes).  sq == nullptr: the query is read from global memory through A, the bases
// entering the band 64 rows at a time (one lane each), instead of from a copy staged in LDS.
template <int NP, bool WITH_Z = true>
static __device__ __forceinline__ int global_wave_diag_pk(const DevIndex& ix, const MemOpt& opt, const uint8_t* sq, int lane, const SeqAcc& A, int w, const GpkFit& fit,
                                          uint8_t* z, bool z_lds, int n_col, bool& ok)
{
    const int qlen = A.qlen, tlen = A.tlen;
    const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
    const uint32_t OED = pk_both(o_del + e_del), OEI = pk_both(o_ins + e_ins), ED = pk_both(e_del), EI = pk_both(e_ins);
    const uint32_t SENT2 = pk_both(GPK_SENT), FLIP = 0x80008000u, BIAS = pk_both(128), ONE = 0x00010001u;
    ok = true;
    // the five scores (+ 128) of each target base: bytes 0..3 of s1 for query codes 0..3, byte 0 of s0 for code 4
    uint32_t row_s1[5], row_s0[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        row_s1[t] = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) row_s1[t] |= (uint32_t)(uint8_t)(opt.mat[t * 5 + q] + 128) << (q << 3);
        row_s0[t] = (uint32_t)(uint8_t)(opt.mat[t * 5 + 4] + 128);
    }
    uint32_t hd[NP], e[NP], qs[NP], SE[NP], am[NP], hrow[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        uint32_t h2 = 0, q2 = 0;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int s = (p + hf * NP) * 64 + lane, j0 = s - w;
            const int hv = j0 == 0 ? 0 : (j0 > 0 && j0 <= w ? -(o_ins + e_ins * j0) : GPK_SENT);    // H(-1, j - 1): upstream's initial eh[j].h
            const int qv = j0 >= 0 && j0 < qlen ? (sq ? (int)sq[j0] : acc_q(A, j0)) : 4;
            h2 |= (uint32_t)(uint16_t)hv << (hf << 4);
            q2 |= (uint32_t)(qv | 0x0c00) << (hf << 4);
        }
        hd[p] = h2; qs[p] = q2; e[p] = SENT2; hrow[p] = SENT2; am[p] = 0;
        SE[p] = pk_pair((p * 64 + lane) * e_ins, ((p + NP) * 64 + lane) * e_ins);
    }
    int base = 0, tch = 4, qch = 4, lb_prev = -1, le_prev = -1;
    for (int i = 0; i < tlen; ++i) {
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        const int lb = beg - (i - w), le = end - (i - w);            // the band's slots in this row: lb <= s < le
        if (lb != lb_prev || le != le_prev) {                        // (constant over the middle of the matrix)
            lb_prev = lb; le_prev = le;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int s0 = p * 64 + lane, s1 = (p + NP) * 64 + lane;
                am[p] = (s0 >= lb && s0 < le ? 0xffffu : 0u) | (s1 >= lb && s1 < le ? 0xffff0000u : 0u);
            }
        }
        if ((i & 63) == 0) {
            { const int ii = i + lane; tch = ii < tlen ? acc_t(ix, A, ii) : 4; }
            if (!sq) { const int jj = i + 128 * NP - w + lane; qch = jj >= 0 && jj < qlen ? acc_q(A, jj) : 4; }
            // re-base: the maximum of the live H(i-1, .) becomes 0; give up if the live values span too much
            uint32_t mx = FLIP, nmn = FLIP;                           // (the minimum as the maximum of the complements: ~x = -x - 1)
#pragma unroll
            for (int p = 0; p < NP; ++p) { mx = pk_max(mx, (hd[p] & am[p]) | (FLIP & ~am[p])); nmn = pk_max(nmn, (~hd[p] & am[p]) | (FLIP & ~am[p])); }
            int rmax = pk_shalf(mx, 0) > pk_shalf(mx, 1) ? pk_shalf(mx, 0) : pk_shalf(mx, 1);
            int rnmn = pk_shalf(nmn, 0) > pk_shalf(nmn, 1) ? pk_shalf(nmn, 0) : pk_shalf(nmn, 1);
            rmax = wave_max(rmax);
            const int rmin = -wave_max(rnmn) - 1;
            if (rmax - rmin > fit.range) { ok = false; return 0; }
            const uint32_t D2 = pk_both(-rmax);
            base += rmax;
#pragma unroll
            for (int p = 0; p < NP; ++p) { hd[p] = pk_addss(hd[p], D2); e[p] = pk_max(pk_addss(e[p], D2), SENT2); }
        }
        const int tb = wave_readlane(tch, i & 63);
        const uint32_t s1r = tb == 0 ? row_s1[0] : tb == 1 ? row_s1[1] : tb == 2 ? row_s1[2] : tb == 3 ? row_s1[3] : row_s1[4];
        const uint32_t s0r = tb == 0 ? row_s0[0] : tb == 1 ? row_s0[1] : tb == 2 ? row_s0[2] : tb == 3 ? row_s0[3] : row_s0[4];
        const int inj = i + 128 * NP - w;                            // query position entering the last slot for the next row
        const int qin = sq ? (inj >= 0 && inj < qlen ? (int)sq[inj] : 4) : wave_readlane(qch, i & 63);
        uint8_t* zi = WITH_Z ? z + (int64_t)i * n_col : nullptr;
        // phase 1: M and the max-plus prefix of every chunk
        uint32_t m[NP], P[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            m[p] = pk_add(pk_sub(hd[p], BIAS), pk_perm(s0r, s1r, qs[p]));
            const uint32_t U = pk_add(pk_sub(m[p], OEI), SE[p]);     // tins + s e_ins
            P[p] = gpk_scan_maxu((U ^ FLIP) & am[p]);                 // (cells outside the band: the identity)
        }
        // what the chunks to the left contribute, per half: chunks 0 .. NP-1 are the low halves, NP .. 2 NP - 1 the high ones
        uint32_t carry[NP];
        {
            uint32_t run = 0, tot[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) { tot[p] = (uint32_t)wave_readlane((int)P[p], 63); carry[p] = run; const uint32_t t = tot[p] & 0xffffu; run = run > t ? run : t; }
#pragma unroll
            for (int p = 0; p < NP; ++p) { carry[p] |= run << 16; const uint32_t t = tot[p] >> 16; run = run > t ? run : t; }
        }
        // phase 2: F, H, E and the direction bits
        uint32_t e2s[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const uint32_t tins = pk_sub(m[p], OEI);
            const uint32_t pex = pk_maxu((uint32_t)__builtin_amdgcn_update_dpp(0, (int)P[p], DPP_WAVE_SHR1, 0xf, 0xf, false), carry[p]);
            const uint32_t g = pk_subss(pk_add(pex ^ FLIP, EI), SE[p]);                // best (tins + s' e_ins) to the left - (s - 1) e_ins; nothing to the left: saturates
            const uint32_t f = pk_max(g, SENT2);
            const uint32_t h1 = pk_max(m[p], e[p]);
            uint32_t d = pk_minu(pk_sub(h1, m[p]), ONE);                                // m >= e ? 0 : 1
            const uint32_t h2 = pk_max(h1, f);
            const uint32_t fw = pk_sra15(pk_sub(h1, h2));                               // 0xffff where f > max(m, e)
            d = (0x00020002u & fw) | (d & ~fw);                                         // h >= f ? d : 2
            const uint32_t t = pk_sub(m[p], OED);
            const uint32_t e3 = pk_max(pk_sub(e[p], ED), t);
            d |= pk_minu(pk_sub(e3, t), ONE) << 2;                                      // e - e_del > m - oe_del ? 1 << 2 : 0
            const uint32_t x = pk_max(pk_sub(f, EI), tins);
            d |= pk_minu(pk_sub(x, tins), ONE) << 5;                                    // f - e_ins > m - oe_ins ? 2 << 4 : 0
            if (WITH_Z && (am[p] & 0xffffu)) z_put(zi + (p * 64 + lane - lb), (uint8_t)d, z_lds);
            if (WITH_Z && (am[p] >> 16)) z_put(zi + ((p + NP) * 64 + lane - lb), (uint8_t)(d >> 16), z_lds);
            hd[p] = (h2 & am[p]) | (SENT2 & ~am[p]);                                     // the lane moves one column to the right along its diagonal
            hrow[p] = h2;
            e2s[p] = (e3 & am[p]) | (SENT2 & ~am[p]);
        }
        if (i < w) {                                                 // the slot whose next column is 0 starts from the boundary column H(i, -1)
            const int sx = w - 1 - i, cx = sx >> 6;
            const uint32_t v = (uint32_t)(uint16_t)(-(o_del + e_del * (i + 1)) - base);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if (cx == p && (sx & 63) == lane) hd[p] = (hd[p] & 0xffff0000u) | v;
                if (cx == p + NP && (sx & 63) == lane) hd[p] = (hd[p] & 0xffffu) | v << 16;
            }
        }
        const uint32_t q_first = (uint32_t)wave_readlane((int)qs[0], 0);      // (before pair 0 moves)
#pragma unroll
        for (int p = 0; p < NP; ++p) {                               // E and the query move one slot down; the last lane takes the first one of the next chunk
            uint32_t efill, qfill;
            if (p + 1 < NP) { efill = (uint32_t)wave_readlane((int)e2s[p + 1 < NP ? p + 1 : p], 0); qfill = (uint32_t)wave_readlane((int)qs[p + 1 < NP ? p + 1 : p], 0); }
            else {
                efill = (uint32_t)wave_readlane((int)e2s[0], 0) >> 16 | (uint32_t)(uint16_t)GPK_SENT << 16;
                qfill = q_first >> 16 | (uint32_t)(qin | 0x0c00) << 16;
            }
            e[p] = (uint32_t)dpp_shl1((int)e2s[p], (int)efill);
            qs[p] = (uint32_t)dpp_shl1((int)qs[p], (int)qfill);
        }
    }
    const int l_end = qlen - 1 - (tlen - 1 - w);                     // H(tlen-1, qlen-1); the caller checked that slot is in the band
    int score = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if ((l_end >> 6) == p) score = pk_shalf((uint32_t)wave_bcast((int)hrow[p], l_end & 63), 0);
        if ((l_end >> 6) == p + NP) score = pk_shalf((uint32_t)wave_bcast((int)hrow[p], l_end & 63), 1);
    }
    score += base;
    return score;
}

