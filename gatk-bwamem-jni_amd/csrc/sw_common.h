// sw_common.h -- the local Smith-Waterman of upstream ksw.c (ksw_align2 / ksw_u8 / ksw_i16), used by mate rescue
// (row a19) and by the long-read seed re-scoring (row a10).  Upstream is an SSE2 "striped" kernel; its few
// observable quirks (E taken before the lazy-F correction, row maxima taken before it, smallest-qe tie rule, the
// second-best bookkeeping) depend on the striped layout, so the layout is kept lane by lane with scalar arithmetic.
#pragma once
#include "dev_common.h"
#include "pk16.h"

// ------------------------------------------------------------------ striped local SW (ksw_align2)
#define KSW_XBYTE  0x10000
#define KSW_XSTOP  0x20000
#define KSW_XSUBO  0x40000
#define KSW_XSTART 0x80000


struct SwIn {
    const uint8_t* ms; int l_ms; int is_rev;     // query = ms or its reverse complement
    int qrev;                                     // > 0: the first qrev query bases are read reversed (second pass)
    int64_t t0; int trev;                         // target = reference from t0; the first trev bases reversed (second pass)
};
DEV int sw_q0(const SwIn& I, int i) { int c = *AS_GLOBAL(const uint8_t, I.ms + (I.is_rev ? I.l_ms - 1 - i : i)); return I.is_rev ? (c < 4 ? 3 - c : 4) : c; }   // (ms: a read of the tile, global memory)
DEV int sw_q(const SwIn& I, int i) { return sw_q0(I, i < I.qrev ? I.qrev - 1 - i : i); }
DEV int sw_t(const DevIndex& ix, const SwIn& I, int i) { return ref_base2(ix, I.t0 + (i < I.trev ? I.trev - 1 - i : i)); }

struct SwScratch { int32_t* H0; int32_t* H1; int32_t* E; int32_t* Hmax; uint64_t* b; int cap_h, cap_b; };

DEV int sat_u8(int x) { return x < 0 ? 0 : x > 255 ? 255 : x; }
DEV int sat_i16(int x) { return x < -32768 ? -32768 : x > 32767 ? 32767 : x; }
DEV int subs_u16(int a, int b) { int x = (int)(uint16_t)a - (int)(uint16_t)b; return x < 0 ? 0 : (int)(int16_t)(uint16_t)x; }

static __device__ __attribute__((noinline)) KswR sw_core(const DevIndex& ix, const MemOpt& opt, const SwIn& I, int size, int qlen, int tlen, int xtra, SwScratch& W, int& err)
{
    const int p = 8 * (3 - size), slen = (qlen + p - 1) / p, u8 = size == 1;
    const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    KswR r; r.score = 0; r.te = r.qe = r.score2 = r.te2 = r.tb = r.qb = -1;
    if (slen * p > W.cap_h) { err |= ERR_SCRATCH; return r; }
    int lo = 127, hi = 0;
    for (int a = 0; a < 25; ++a) { if (opt.mat[a] < lo) lo = opt.mat[a]; if (opt.mat[a] > hi) hi = opt.mat[a]; }
    const int shift = (256 - (lo & 0xff)) & 0xff, qmax = hi;
    int n_b = 0, te = -1, gmax = 0;
    const int minsc = (xtra & KSW_XSUBO) ? xtra & 0xffff : 0x10000;
    const int endsc = (xtra & KSW_XSTOP) ? xtra & 0xffff : 0x10000;
    int32_t *H0 = W.H0, *H1 = W.H1, *E = W.E, *Hmax = W.Hmax;
    for (int i = 0; i < slen * p; ++i) { E[i] = 0; H0[i] = 0; Hmax[i] = 0; }
    int h[16], f[16], mx[16];
    for (int i = 0; i < tlen; ++i) {
        const int tb = sw_t(ix, I, i);
        int imax, done = 0;
        for (int l = 0; l < p; ++l) { f[l] = 0; mx[l] = 0; }
        h[0] = 0;
        for (int l = 1; l < p; ++l) h[l] = H0[(slen - 1) * p + l - 1];
        for (int j = 0; j < slen; ++j) {
            for (int l = 0; l < p; ++l) {
                int pos = j + l * slen;
                int sc = (pos >= qlen ? 0 : opt.mat[tb * 5 + sw_q(I, pos)]) + (u8 ? shift : 0);
                int hh, ee = E[j * p + l], t;
                if (u8) { hh = sat_u8(h[l] + sc); hh = sat_u8(hh - shift); }
                else hh = sat_i16(h[l] + sc);
                hh = hh > ee ? hh : ee;
                hh = hh > f[l] ? hh : f[l];
                mx[l] = mx[l] > hh ? mx[l] : hh;
                H1[j * p + l] = hh;
                if (u8) { ee = sat_u8(ee - e_del); t = sat_u8(hh - oe_del); }
                else    { ee = subs_u16(ee, e_del); t = subs_u16(hh, oe_del); }
                ee = ee > t ? ee : t;
                E[j * p + l] = ee;
                if (u8) { f[l] = sat_u8(f[l] - e_ins); t = sat_u8(hh - oe_ins); }
                else    { f[l] = subs_u16(f[l], e_ins); t = subs_u16(hh, oe_ins); }
                f[l] = f[l] > t ? f[l] : t;
                h[l] = H0[j * p + l];
            }
        }
        for (int k = 0; k < 16 && !done; ++k) {          // lazy-F across segment boundaries
            for (int l = p - 1; l > 0; --l) f[l] = f[l - 1];
            f[0] = 0;
            for (int j = 0; j < slen; ++j) {
                int all = 1;
                for (int l = 0; l < p; ++l) {
                    int hh = H1[j * p + l];
                    hh = hh > f[l] ? hh : f[l];
                    H1[j * p + l] = hh;
                    if (u8) { hh = sat_u8(hh - oe_ins); f[l] = sat_u8(f[l] - e_ins); if (sat_u8(f[l] - hh) != 0) all = 0; }
                    else    { hh = subs_u16(hh, oe_ins); f[l] = subs_u16(f[l], e_ins); if (f[l] > hh) all = 0; }
                }
                if (all) { done = 1; break; }
            }
        }
        imax = mx[0];
        for (int l = 1; l < p; ++l) imax = imax > mx[l] ? imax : mx[l];
        if (imax >= minsc) {
            if (n_b == 0 || (int32_t)W.b[n_b - 1] + 1 != i) {
                if (n_b >= W.cap_b) { err |= ERR_SCRATCH; return r; }
                W.b[n_b++] = (uint64_t)imax << 32 | (uint32_t)i;
            } else if ((int)(W.b[n_b - 1] >> 32) < imax) W.b[n_b - 1] = (uint64_t)imax << 32 | (uint32_t)i;
        }
        if (imax > gmax) {
            gmax = imax; te = i;
            for (int j = 0; j < slen * p; ++j) Hmax[j] = H1[j];
            if (u8) { if (gmax + shift >= 255 || gmax >= endsc) break; }
            else if (gmax >= endsc) break;
        }
        int32_t* S = H1; H1 = H0; H0 = S;
    }
    r.score = u8 ? (gmax + shift < 255 ? gmax : 255) : gmax;
    r.te = te;
    if (!u8 || r.score != 255) {
        int max = -1, tmp, n = slen * p;
        for (int i = 0; i < n; ++i) {
            int v = Hmax[i];
            if (v > max) { max = v; r.qe = i / p + i % p * slen; }
            else if (v == max && (tmp = i / p + i % p * slen) < r.qe) r.qe = tmp;
        }
        if (n_b > 0) {
            int i = (r.score + qmax - 1) / qmax;
            int low = te - i, high = te + i;
            for (i = 0; i < n_b; ++i) {
                int e = (int32_t)W.b[i];
                if ((e < low || e > high) && (int)(W.b[i] >> 32) > r.score2) { r.score2 = (int)(W.b[i] >> 32); r.te2 = e; }
            }
        }
    }
    return r;
}

DEV KswR sw_align2(const DevIndex& ix, const MemOpt& opt, SwIn I, int qlen, int tlen, int xtra, SwScratch& W, int& err)
{
    const int size = (xtra & KSW_XBYTE) ? 1 : 2;
    I.qrev = 0; I.trev = 0;
    KswR r = sw_core(ix, opt, I, size, qlen, tlen, xtra, W, err);
    if ((xtra & KSW_XSTART) == 0 || ((xtra & KSW_XSUBO) && r.score < (xtra & 0xffff))) return r;
    I.qrev = r.qe + 1; I.trev = r.te + 1;             // upstream reverses both prefixes in place, then runs over the full target
    KswR rr = sw_core(ix, opt, I, size, r.qe + 1, tlen, KSW_XSTOP | r.score, W, err);
    if (r.score == rr.score) { r.tb = r.te - rr.te; r.qb = r.qe - rr.qe; }
    return r;
}


// ------------------------------------------------------------------ the same kernel across the lanes of a wavefront
// Mate rescue needs ksw_align2 over a ~600-base window for a few per cent of the pairs; the scalar form above, with its
// arrays in global memory, takes ~0.4 s of dependent round trips per call.  In the wave form upstream's SSE2 lane k (16
// bytes or 8 shorts per vector) is lane k of a 16-lane group, the segment loop and the lazy-F loop are the sequential
// dimension exactly as upstream has them, the H/E/Hmax stripes and the query profile live in registers (NSEG segments)
// and only the row-maxima list is in LDS.
struct SwLds { uint64_t* b; int cap_b; };               // [64 / GW][cap_b] row-maxima lists, one per group of GW lanes

// Several alignments at once, one per group of GW lanes of the wavefront: GW = 16 for byte mode (upstream's 16-byte vector is
// exactly one group: four alignments per wave), GW = 8 for 16-bit mode (eight shorts per vector: eight alignments per wave).  A wave that owns several pairs in need of rescue would otherwise run their
// alignments one after another while every other wave has finished.  All arguments are per lane but uniform within a group;
// `on` says whether the group has an alignment at all; `size` (byte or 16-bit mode) is the same for the whole wave.
template <int NSEG, int GW>
static __device__ __attribute__((noinline)) KswR sw_core_wave4(const DevIndex& ix, const MemOpt& opt, const SwIn& I, bool on, int size, int qlen, int tlen, int xtra, const SwLds& W, int lane, int& err_out)
{
    int err = 0;                                                // (behind the reference it would be a flat access in the row loop)
    static_assert(GW == 16 || GW == 8, "group width");
    const int p = 8 * (3 - size), u8 = size == 1;              // (GW = 8 is for 16-bit mode only: p == GW)
    const int g = lane / GW, sl = lane % GW;
    const unsigned long long gmask = GW == 16 ? 0xffffull : 0xffull;
    const int slen = on ? (qlen + p - 1) / p : 0;
    const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    KswR r; r.score = 0; r.te = r.qe = r.score2 = r.te2 = r.tb = r.qb = -1;
    int lo = 127, hi = 0;
    for (int a = 0; a < 25; ++a) { if (opt.mat[a] < lo) lo = opt.mat[a]; if (opt.mat[a] > hi) hi = opt.mat[a]; }
    const int shift = (256 - (lo & 0xff)) & 0xff, qmax = hi;
    int n_b = 0, te = -1, gmax = 0, last_i = -2, last_sc = 0;
    const int minsc = (xtra & KSW_XSUBO) ? xtra & 0xffff : 0x10000;
    const int endsc = (xtra & KSW_XSTOP) ? xtra & 0xffff : 0x10000;
    const bool la = on && sl < p;                           // this lane is one of upstream's vector lanes of a live group
    const int l = sl < p ? sl : 0;
    uint64_t* const bl = W.b + (size_t)g * W.cap_b;          // the group's row-maxima list (LDS: said so at every access, see AS_LDS)
    const int cap_b = W.cap_b, trev = I.trev;
    const int64_t t0 = I.t0, l_pac = ix.l_pac;
    const uint8_t* const pac = ix.pac;
    const ScoreTab ST = score_tab(opt);
    int H0[NSEG], H1[NSEG], E[NSEG], Hmax[NSEG], sn[NSEG];
    uint32_t sp[NSEG];
    bool pad[NSEG];
#pragma unroll
    for (int j = 0; j < NSEG; ++j) {
        const int pos = j + l * slen;
        H0[j] = H1[j] = E[j] = Hmax[j] = 0;
        pad[j] = !(on && j < slen && pos < qlen);
        score_lane(ST, pad[j] ? 4 : sw_q(I, pos), sp[j], sn[j]);
    }
    bool stop = !on;
    PacCache pc; pc.w = -1; pc.v = 0;
    for (int i = 0; ; ++i) {
        const bool run = !stop && i < tlen;
        if (__ballot(run) == 0ull) break;
        const int tb = run ? ref_base2_cp(pac, l_pac, pc, t0 + (i < trev ? trev - 1 - i : i)) : 0;
        int f = 0, mx = 0;
        int hlast = 0;
#pragma unroll
        for (int j = 0; j < NSEG; ++j) if (j == slen - 1) hlast = H0[j];
        int h = __shfl_up(la ? hlast : 0, 1);
        if (sl == 0) h = 0;
#pragma unroll
        for (int j = 0; j < NSEG; ++j) {
            if (run && j < slen) {
                const int sc = (pad[j] ? 0 : score_at(sp[j], sn[j], tb)) + (u8 ? shift : 0);
                int hh, ee = E[j], t;
                if (u8) { hh = sat_u8(h + sc); hh = sat_u8(hh - shift); }
                else hh = sat_i16(h + sc);
                hh = hh > ee ? hh : ee;
                hh = hh > f ? hh : f;
                mx = mx > hh ? mx : hh;
                H1[j] = hh;
                if (u8) { ee = sat_u8(ee - e_del); t = sat_u8(hh - oe_del); }
                else    { ee = subs_u16(ee, e_del); t = subs_u16(hh, oe_del); }
                E[j] = ee > t ? ee : t;
                if (u8) { f = sat_u8(f - e_ins); t = sat_u8(hh - oe_ins); }
                else    { f = subs_u16(f, e_ins); t = subs_u16(hh, oe_ins); }
                f = f > t ? f : t;
                h = H0[j];
            }
        }
        bool done = !run;
        for (int k = 0; k < 16; ++k) {                      // lazy-F across segment boundaries, each group until its own fixed point
            if (__ballot(!done) == 0ull) break;
            int fs = __shfl_up(la ? f : 0, 1);
            if (sl == 0) fs = 0;
            if (!done) f = fs;
#pragma unroll
            for (int j = 0; j < NSEG; ++j) {
                const bool act = !done && j < slen;
                bool more = false;
                if (act) {
                    int hh = H1[j];
                    hh = hh > f ? hh : f;
                    H1[j] = hh;
                    if (u8) { hh = sat_u8(hh - oe_ins); f = sat_u8(f - e_ins); more = sat_u8(f - hh) != 0; }
                    else    { hh = subs_u16(hh, oe_ins); f = subs_u16(f, e_ins); more = f > hh; }
                }
                const unsigned long long bal = __ballot(la && act && more);
                if (act && ((bal >> (g * GW)) & gmask) == 0ull) done = true;
            }
        }
        int imax = la && run ? mx : 0;
        for (int o = GW / 2; o > 0; o >>= 1) { const int u = __shfl_xor(imax, o); imax = imax > u ? imax : u; }
        imax = __shfl(imax, g * GW);
        // the list of row maxima: a row next to the last listed one only replaces it when it is higher.  The last entry is kept
        // in registers by every lane (the list itself is only read after the loop), so the rows need no barrier
        if (run && imax >= minsc) {
            if (n_b == 0 || last_i + 1 != i) {
                if (n_b >= cap_b) { err |= ERR_SCRATCH; stop = true; }
                else { if (sl == 0) *AS_LDS(uint64_t, bl + n_b) = (uint64_t)imax << 32 | (uint32_t)i; ++n_b; last_i = i; last_sc = imax; }
            } else if (last_sc < imax) { if (sl == 0) *AS_LDS(uint64_t, bl + (n_b - 1)) = (uint64_t)imax << 32 | (uint32_t)i; last_i = i; last_sc = imax; }
        }
        if (run && !stop && imax > gmax) {
            gmax = imax; te = i;
#pragma unroll
            for (int j = 0; j < NSEG; ++j) Hmax[j] = H1[j];
            if (u8) { if (gmax + shift >= 255 || gmax >= endsc) stop = true; }
            else if (gmax >= endsc) stop = true;
        }
        if (run) {
#pragma unroll
            for (int j = 0; j < NSEG; ++j) { const int t = H0[j]; H0[j] = H1[j]; H1[j] = t; }
        }
    }
    __syncthreads();
    r.score = u8 ? (gmax + shift < 255 ? gmax : 255) : gmax;
    r.te = te;
    {
        int best = -1, bq = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < NSEG; ++j) {
            if (la && j < slen) {
                const int v = Hmax[j], pos = j + l * slen;
                if (v > best || (v == best && pos < bq)) { best = v; bq = pos; }
            }
        }
        for (int o = GW / 2; o > 0; o >>= 1) {
            const int ub = __shfl_xor(best, o), uq = __shfl_xor(bq, o);
            if (ub > best || (ub == best && uq < bq)) { best = ub; bq = uq; }
        }
        const int qe = __shfl(bq, g * GW);
        if (on && (!u8 || r.score != 255)) {
            r.qe = qe;
            if (n_b > 0) {
                int i = (r.score + qmax - 1) / qmax;
                int low = te - i, high = te + i;
                for (i = 0; i < n_b; ++i) {
                    const uint64_t be = *AS_LDS(const uint64_t, bl + i);
                    int e = (int32_t)be;
                    if ((e < low || e > high) && (int)(be >> 32) > r.score2) { r.score2 = (int)(be >> 32); r.te2 = e; }
                }
            }
        }
    }
    __syncthreads();
    if (err) err_out |= err;
    return r;
}

// ksw_align2 for 64 / GW alignments (see sw_core_wave4); arguments per lane, uniform within a group of GW lanes
template <int NSEG, int GW>
DEV KswR sw_align2_wave4(const DevIndex& ix, const MemOpt& opt, SwIn I, bool on, int size, int qlen, int tlen, int xtra, const SwLds& W, int lane, int& err)
{
    I.qrev = 0; I.trev = 0;
    KswR r = sw_core_wave4<NSEG, GW>(ix, opt, I, on, size, qlen, tlen, xtra, W, lane, err);
    const bool again = on && !((xtra & KSW_XSTART) == 0 || ((xtra & KSW_XSUBO) && r.score < (xtra & 0xffff)));
    if (__ballot(again) == 0ull) return r;
    I.qrev = r.qe + 1; I.trev = r.te + 1;
    KswR rr = sw_core_wave4<NSEG, GW>(ix, opt, I, again, size, again ? r.qe + 1 : 0, tlen, KSW_XSTOP | r.score, W, lane, err);
    if (again && r.score == rr.score) { r.tb = r.te - rr.te; r.qb = r.qe - rr.qe; }
    return r;
}


// ------------------------------------------------------------------ two alignments per lane
// The kernel above keeps one 8- or 16-bit value per 32-bit register.  Here every register carries two alignments, one per
// 16-bit half (pk16.h): a group of GW lanes (16 in byte mode, 8 in 16-bit mode: upstream's vector width) runs alignments 2g (low
// halves) and 2g + 1 (high halves) -- eight or sixteen per wavefront -- and one stream of packed instructions computes both.
// Byte mode: unsigned saturation at 255 is a packed minimum, saturation at 0 the clamp bit of the packed subtraction.  16-bit
// mode: the signed saturation of H + score cannot trigger for the queries these kernels take (at most 256 bases x the largest
// byte score), so it is a plain packed add.  Either way the scores of both alignments against their own target bases come out of
// one byte permute (the lane keeps a biased score for the four target bases as the four bytes of a word per segment and
// alignment).  Everything the two alignments do not share (target length, stop, the row-maxima list, the best row) is kept
// per half; segments beyond an alignment's own are masked, and F passes them unchanged.  The lazy-F loop runs whole passes until
// no half of the wave asks for more: a pass past an alignment's own fixed point changes nothing (at the fixed point F is below
// H - o - e everywhere, which is what the pass would take the maximum with).
struct SwPair { SwIn I[2]; bool on[2]; int qlen[2], tlen[2], xtra[2]; };

template <int NSEG, bool U8>
static __device__ __attribute__((noinline)) void sw_core_packed(const DevIndex& ix, const MemOpt& opt, const SwPair& P, const SwLds& W, int lane, int& err_out, KswR R[2])
{
    int err = 0;
    constexpr int GW = U8 ? 16 : 8;
    const int g = lane / GW, sl = lane % GW;
    const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
    int lo = 127, hi = 0;
    for (int a = 0; a < 25; ++a) { if (opt.mat[a] < lo) lo = opt.mat[a]; if (opt.mat[a] > hi) hi = opt.mat[a]; }
    const int shift = (256 - (lo & 0xff)) & 0xff, qmax = hi;
    const int bias = U8 ? shift : 128;                           // what is added to a score to make it an unsigned byte
    const uint32_t BIAS = pk_both(bias), C255 = pk_both(255), OED = pk_both(o_del + e_del), ED = pk_both(e_del), OEI = pk_both(o_ins + e_ins), EI = pk_both(e_ins);
    int slen[2], n_b[2] = { 0, 0 }, te[2] = { -1, -1 }, gmax[2] = { 0, 0 }, minsc[2], endsc[2], last_i[2] = { -2, -2 }, last_sc[2] = { 0, 0 };
    int tlen[2], trev[2], qlen[2];                               // (P and W are read once: they sit behind references the row loop's stores might alias)
    int64_t t0[2];
    const int cap_b = W.cap_b;
    const int64_t l_pac = ix.l_pac;
    const uint8_t* const pac = ix.pac;
    bool stop[2], on[2];
    SwIn Il[2];
    uint64_t* bl[2];
    PacCache pc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        slen[h] = P.on[h] ? (P.qlen[h] + GW - 1) / GW : 0;
        minsc[h] = (P.xtra[h] & KSW_XSUBO) ? P.xtra[h] & 0xffff : 0x10000;
        endsc[h] = (P.xtra[h] & KSW_XSTOP) ? P.xtra[h] & 0xffff : 0x10000;
        stop[h] = !P.on[h]; on[h] = P.on[h];
        tlen[h] = P.tlen[h]; qlen[h] = P.qlen[h]; Il[h] = P.I[h]; trev[h] = Il[h].trev; t0[h] = Il[h].t0;
        bl[h] = W.b + (size_t)(g * 2 + h) * W.cap_b;             // LDS: said so at every access (AS_LDS)
        pc[h].w = -1; pc[h].v = 0;
        R[h].score = 0; R[h].te = R[h].qe = R[h].score2 = R[h].te2 = R[h].tb = R[h].qb = -1;
    }
    const ScoreTab ST = score_tab(opt);
    uint32_t H0[NSEG], H1[NSEG], E[NSEG], Hmax[NSEG], SA[NSEG], SB[NSEG], sm[NSEG];
#pragma unroll
    for (int j = 0; j < NSEG; ++j) {
        H0[j] = H1[j] = E[j] = Hmax[j] = 0;
        uint32_t sw[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pos = j + sl * slen[h];
            const bool pad = !(on[h] && j < slen[h] && pos < qlen[h]);
            uint32_t sp; int sn;
            score_lane(ST, pad ? 4 : sw_q(Il[h], pos), sp, sn);
            // score + bias per target base as unsigned bytes; a pad position scores 0 against everything
            uint32_t w = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) w |= (uint32_t)(((pad ? 0 : (int)(int8_t)(sp >> (b << 3))) + bias) & 0xff) << (b << 3);
            sw[h] = w;
        }
        SA[j] = sw[0]; SB[j] = sw[1];
        sm[j] = (j < slen[0] ? 0xffffu : 0u) | (j < slen[1] ? 0xffff0000u : 0u);
    }
    for (int i = 0; ; ++i) {
        bool run[2];
        uint32_t tb[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            run[h] = !stop[h] && i < tlen[h];
            tb[h] = run[h] ? (uint32_t)ref_base2_cp(pac, l_pac, pc[h], t0[h] + (i < trev[h] ? trev[h] - 1 - i : i)) : 0u;
        }
        if (__ballot(run[0] || run[1]) == 0ull) break;
        const uint32_t runm = (run[0] ? 0xffffu : 0u) | (run[1] ? 0xffff0000u : 0u);
        uint32_t f = 0, mx = 0, hlast = 0;
#pragma unroll
        for (int j = 0; j < NSEG; ++j) {
            if (j == slen[0] - 1) hlast = (hlast & 0xffff0000u) | (H0[j] & 0xffffu);
            if (j == slen[1] - 1) hlast = (hlast & 0xffffu) | (H0[j] & 0xffff0000u);
        }
        uint32_t h = (uint32_t)__shfl_up((int)hlast, 1);
        if (sl == 0) h = 0;
#pragma unroll
        for (int j = 0; j < NSEG; ++j) {                    // (segments beyond an alignment's own are masked, not skipped: no branches in the row)
            const uint32_t sc = pk_bytes2(SA[j], SB[j], tb[0], tb[1]);
            uint32_t hh;
            if (U8) { hh = pk_minu(pk_add(h, sc), C255); hh = pk_subs(hh, BIAS); }
            else hh = pk_sub(pk_add(h, sc), BIAS);
            hh = pk_max(hh, E[j]);
            hh = pk_max(hh, f) & sm[j];
            mx = pk_max(mx, hh);
            H1[j] = hh;
            const uint32_t t = pk_subs(hh, OED);
            E[j] = pk_max(pk_subs(E[j], ED), t);
            const uint32_t t2 = pk_subs(hh, OEI);
            f = (pk_max(pk_subs(f, EI), t2) & sm[j]) | (f & ~sm[j]);       // (F leaves an alignment's last segment unchanged by the masked ones behind it)
            h = H0[j];
        }
        for (int k = 0; k < 16; ++k) {                      // lazy-F across segment boundaries, whole passes
            uint32_t fs = (uint32_t)__shfl_up((int)f, 1);
            if (sl == 0) fs = 0;
            f = fs;
            uint32_t more = 0;
#pragma unroll
            for (int j = 0; j < NSEG; ++j) {
                uint32_t hh = pk_max(H1[j], f);
                H1[j] = hh;
                hh = pk_subs(hh, OEI);
                f = (pk_subs(f, EI) & sm[j]) | (f & ~sm[j]);
                more |= pk_subs(f, hh) & sm[j];
            }
            if (__ballot((more & runm) != 0) == 0ull) break;
        }
        uint32_t imax = mx;
        for (int o = GW / 2; o > 0; o >>= 1) imax = pk_max(imax, (uint32_t)__shfl_xor((int)imax, o));
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int im = pk_half(imax, hf);
            if (run[hf] && im >= minsc[hf]) {                   // (the last listed row lives in registers: see sw_core_wave4)
                if (n_b[hf] == 0 || last_i[hf] + 1 != i) {
                    if (n_b[hf] >= cap_b) { err |= ERR_SCRATCH; stop[hf] = true; }
                    else { if (sl == 0) *AS_LDS(uint64_t, bl[hf] + n_b[hf]) = (uint64_t)im << 32 | (uint32_t)i; ++n_b[hf]; last_i[hf] = i; last_sc[hf] = im; }
                } else if (last_sc[hf] < im) { if (sl == 0) *AS_LDS(uint64_t, bl[hf] + (n_b[hf] - 1)) = (uint64_t)im << 32 | (uint32_t)i; last_i[hf] = i; last_sc[hf] = im; }
            }
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int im = pk_half(imax, hf);
            if (run[hf] && !stop[hf] && im > gmax[hf]) {
                gmax[hf] = im; te[hf] = i;
                const uint32_t hm = hf ? 0xffff0000u : 0xffffu;
#pragma unroll
                for (int j = 0; j < NSEG; ++j) Hmax[j] = (Hmax[j] & ~hm) | (H1[j] & hm);
                if (U8) { if (gmax[hf] + shift >= 255 || gmax[hf] >= endsc[hf]) stop[hf] = true; }
                else if (gmax[hf] >= endsc[hf]) stop[hf] = true;
            }
        }
#pragma unroll
        for (int j = 0; j < NSEG; ++j) { const uint32_t t = H0[j]; H0[j] = H1[j]; H1[j] = t; }
    }
    __syncthreads();
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        KswR r; r.qe = r.score2 = r.te2 = r.tb = r.qb = -1;    // (a local: R sits behind a pointer, and the list walk below updates score2 / te2 per entry)
        r.score = U8 ? (gmax[hf] + shift < 255 ? gmax[hf] : 255) : gmax[hf];
        r.te = te[hf];
        int best = -1, bq = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < NSEG; ++j) {
            if (j < slen[hf]) {
                const int v = pk_half(Hmax[j], hf), pos = j + sl * slen[hf];
                if (v > best || (v == best && pos < bq)) { best = v; bq = pos; }
            }
        }
        for (int o = GW / 2; o > 0; o >>= 1) {
            const int ub = __shfl_xor(best, o), uq = __shfl_xor(bq, o);
            if (ub > best || (ub == best && uq < bq)) { best = ub; bq = uq; }
        }
        if (on[hf] && (!U8 || r.score != 255)) {
            r.qe = bq;
            if (n_b[hf] > 0) {
                int i = (r.score + qmax - 1) / qmax;
                const int low = te[hf] - i, high = te[hf] + i;
                for (i = 0; i < n_b[hf]; ++i) {
                    const uint64_t be = *AS_LDS(const uint64_t, bl[hf] + i);
                    const int e = (int32_t)be;
                    if ((e < low || e > high) && (int)(be >> 32) > r.score2) { r.score2 = (int)(be >> 32); r.te2 = e; }
                }
            }
        }
        R[hf] = r;
    }
    __syncthreads();
    if (err) err_out |= err;
}

// ksw_align2 for the sixteen (byte mode: eight) alignments of a wavefront, two per group of GW lanes; P per lane, uniform within a group
template <int NSEG, bool U8>
DEV void sw_align2_packed(const DevIndex& ix, const MemOpt& opt, SwPair P, const SwLds& W, int lane, int& err, KswR R[2])
{
#pragma unroll
    for (int h = 0; h < 2; ++h) { P.I[h].qrev = 0; P.I[h].trev = 0; }
    sw_core_packed<NSEG, U8>(ix, opt, P, W, lane, err, R);
    SwPair Q = P;
    bool again[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        again[h] = P.on[h] && !((P.xtra[h] & KSW_XSTART) == 0 || ((P.xtra[h] & KSW_XSUBO) && R[h].score < (P.xtra[h] & 0xffff)));
        Q.on[h] = again[h];
        Q.I[h].qrev = R[h].qe + 1; Q.I[h].trev = R[h].te + 1;
        Q.qlen[h] = again[h] ? R[h].qe + 1 : 0;
        Q.xtra[h] = KSW_XSTOP | R[h].score;
    }
    if (__ballot(again[0] || again[1]) == 0ull) return;
    KswR RR[2];
    sw_core_packed<NSEG, U8>(ix, opt, Q, W, lane, err, RR);
#pragma unroll
    for (int h = 0; h < 2; ++h)
        if (again[h] && R[h].score == RR[h].score) { R[h].tb = R[h].te - RR[h].te; R[h].qb = R[h].qe - RR[h].qe; }
}
