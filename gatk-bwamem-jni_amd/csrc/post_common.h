// post_common.h -- device functions shared by the single-end and paired-end finalisation
// kernels: global alignment + CIGAR/NM/MD, MAPQ, primary marking, XA, record writer.
// Behavioural contract: SURVEY.md rows a14-a18 (see k_post.hip for the upstream names).
#pragma once
#include "dev_common.h"
#include "wave_ops.h"

#define INT_MAX_ 2147483647
#define MINUS_INF (-0x40000000)

struct PostScratch {
    int32_t* eh;  int eh_cap;        // (h,e) pairs
    uint8_t* z;   int64_t z_cap;     // traceback matrix
    uint32_t* cig; int cig_cap;      // CIGAR ops (slot 0 kept free for a leading clip)
    char* md;     int md_cap;
    int err;
};

DEV PostScratch post_scratch_for(const TileView& tv, int r)
{
    PostScratch S;
    uint8_t* base = tv.post_scratch + (size_t)r * tv.post_scratch_per_read;
    int L = tv.max_len;
    S.eh_cap = 2 * (L + 2);
    S.cig_cap = 4 * L + 16;
    S.md_cap = 8 * L + 32;
    S.eh = (int32_t*)base;                       base += (size_t)S.eh_cap * 4;
    S.cig = (uint32_t*)base;                     base += (size_t)S.cig_cap * 4;
    S.md = (char*)base;                          base += (size_t)S.md_cap;
    S.z = base;
    S.z_cap = tv.post_scratch_per_read - ((int64_t)S.eh_cap * 4 + (int64_t)S.cig_cap * 4 + S.md_cap);
    S.err = 0;
    return S;
}

struct OutBuf {
    uint8_t* p; int cap, len; bool ovf;
    __device__ void put32(int32_t v) { if (len + 4 > cap) { ovf = true; return; } *(int32_t*)(p + len) = v; len += 4; }
    __device__ void putc(char c) { if (len + 1 > cap) { ovf = true; return; } p[len++] = (uint8_t)c; }
    __device__ void pad4() { while (len & 3) putc(0); }
    // decimal digits through a register (16 BCD nibbles), not a private array: dynamically indexed private arrays live in scratch memory
    __device__ void put_digits(unsigned long x, int min_digits) {
        unsigned long bcd = 0; int l = 0;
        if (x >> 32) do { bcd |= (x % 10) << (l << 2); ++l; x /= 10; } while (x);
        else { unsigned y = (unsigned)x; do { bcd |= (unsigned long)(y % 10u) << (l << 2); ++l; y /= 10u; } while (y); }
        l = l > min_digits ? l : min_digits;
        while (l > 0) { --l; putc((char)('0' + (int)(bcd >> (l << 2) & 15ul))); }
    }
    __device__ void putl(long v) {
        const unsigned long x = v < 0 ? 0ul - (unsigned long)v : (unsigned long)v, e16 = 10000000000000000ul;
        if (v < 0) putc('-');
        if (x >= e16) { put_digits(x / e16, 1); put_digits(x % e16, 16); }
        else put_digits(x, 1);
    }
};

struct MdBuf {
    char* s; int cap, l; bool ovf;
    __device__ void putc(char c) { if (l + 1 >= cap) { ovf = true; return; } s[l++] = c; }
    __device__ void putw(int v) {
        unsigned x = v < 0 ? 0u - (unsigned)v : (unsigned)v;
        unsigned long bcd = 0; int n = 0;
        do { bcd |= (unsigned long)(x % 10u) << (n << 2); ++n; x /= 10u; } while (x);
        if (v < 0) putc('-');
        while (n > 0) { --n; putc((char)('0' + (int)(bcd >> (n << 2) & 15ul))); }
    }
};

// one alignment record (upstream mem_aln_t); cigar / md point into PostScratch
struct AlnRec {
    int64_t pos;
    int rid, flag, is_rev, is_alt, mapq, NM, n_cigar;
    const uint32_t* cigar;
    const char* md; int l_md;
    int score, sub, alt_sc;
    int ref_len;                 // sum of M and D lengths (jnibwa.c:30-41 cigarRefLen)
};

struct MateInfo { int rid; int64_t pos; int is_rev; int ref_len; };

struct SeqAcc {                  // query / target accessors with optional reversal (no copies)
    const uint8_t* q; int qlen; int rev;
    int64_t t0; int tlen;
};
DEV int acc_q(const SeqAcc& A, int j) { return A.q[A.rev ? A.qlen - 1 - j : j]; }
DEV int acc_t(const DevIndex& ix, const SeqAcc& A, int i) { return ref_base2(ix, A.t0 + (A.rev ? A.tlen - 1 - i : i)); }
#include "global_pk.h"

// The same accessors for the per-base loops of the one-lane-per-read kernels, through a one-word cache each: 16
// reference bases or 4 query bases per global load instead of one (those loops are bound by the latency of their
// dependent loads, not by arithmetic).
struct SeqCache { int64_t pw; uint32_t pv; const uint32_t* qp; uint32_t qv; };
DEV void seq_cache_init(SeqCache& c) { c.pw = -1; c.pv = 0; c.qp = 0; c.qv = 0; }
DEV int pac_base_c(const DevIndex& ix, SeqCache& c, int64_t l)
{
    const int64_t w = l >> 4;
    if (w != c.pw) { c.pw = w; c.pv = ((const uint32_t*)ix.pac)[w]; }
    const int k = (int)(l & 15);
    return (int)(c.pv >> (((k >> 2) << 3) + 6 - ((k & 3) << 1)) & 3u);
}
DEV int acc_t_c(const DevIndex& ix, const SeqAcc& A, SeqCache& c, int i)
{
    const int64_t p = A.t0 + (A.rev ? A.tlen - 1 - i : i);
    return p < ix.l_pac ? pac_base_c(ix, c, p) : 3 - pac_base_c(ix, c, (ix.l_pac << 1) - 1 - p);
}
DEV int acc_q_c(const SeqAcc& A, SeqCache& c, int j)
{
    const uint8_t* a = A.q + (A.rev ? A.qlen - 1 - j : j);
    const uint32_t* wp = (const uint32_t*)(a - ((uintptr_t)a & 3));      // pointer arithmetic, not an integer round trip: keeps the address space (global loads, not flat)
    if (wp != c.qp) { c.qp = wp; c.qv = *wp; }
    return (int)(c.qv >> (((uintptr_t)a & 3) << 3) & 0xffu);
}

// ksw_global2: banded global alignment, optional traceback into S.cig[1..]
DEV int global_dp(const DevIndex& ix, const MemOpt& opt, PostScratch& S, const SeqAcc& A, int w, bool want_cigar, int* n_cigar_)
{
    const int qlen = A.qlen, tlen = A.tlen;
    const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    int i, j, k, score, n_col;
    if (n_cigar_) *n_cigar_ = 0;
    n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    if (2 * (qlen + 1) > S.eh_cap || (want_cigar && (int64_t)n_col * tlen > S.z_cap)) { S.err |= ERR_SCRATCH; return 0; }
    int32_t* eh = S.eh;
    uint8_t* z = S.z;
    eh[0] = 0; eh[1] = MINUS_INF;
    for (j = 1; j <= qlen && j <= w; ++j) { eh[2 * j] = -(o_ins + e_ins * j); eh[2 * j + 1] = MINUS_INF; }
    for (; j <= qlen; ++j) eh[2 * j] = eh[2 * j + 1] = MINUS_INF;
    for (i = 0; i < tlen; ++i) {
        int32_t f = MINUS_INF, h1, beg, end, t;
        const int tb = acc_t(ix, A, i);
        uint8_t* zi = z + (int64_t)i * n_col;
        beg = i > w ? i - w : 0;
        end = i + w + 1 < qlen ? i + w + 1 : qlen;
        h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : MINUS_INF;
        for (j = beg; j < end; ++j) {
            int32_t h, M = eh[2 * j], e = eh[2 * j + 1];
            uint8_t d;
            eh[2 * j] = h1;
            M += opt.mat[tb * 5 + acc_q(A, j)];
            d = M >= e ? 0 : 1;
            h = M >= e ? M : e;
            d = h >= f ? d : 2;
            h = h >= f ? h : f;
            h1 = h;
            t = M - oe_del;
            e -= e_del;
            d |= e > t ? 1 << 2 : 0;
            e  = e > t ? e : t;
            eh[2 * j + 1] = e;
            t = M - oe_ins;
            f -= e_ins;
            d |= f > t ? 2 << 4 : 0;
            f  = f > t ? f : t;
            if (want_cigar) zi[j - beg] = d;
        }
        eh[2 * end] = h1; eh[2 * end + 1] = MINUS_INF;
    }
    score = eh[2 * qlen];
    if (want_cigar) {   // backtrack; ops are produced end-to-start, then reversed in place
        int n = 0, which = 0;
        uint32_t* cigar = S.cig + 1;
        const int cap = S.cig_cap - 2;
        i = tlen - 1; k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
        while (i >= 0 && k >= 0) {
            int op;
            which = z[(int64_t)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
            if (which == 0) { op = 0; --i; --k; }
            else if (which == 1) { op = 2; --i; }
            else { op = 1; --k; }
            if (n == 0 || op != (int)(cigar[n - 1] & 0xf)) { if (n >= cap) { S.err |= ERR_CIGAR_CAP; return score; } cigar[n++] = 1u << 4 | (uint32_t)op; }
            else cigar[n - 1] += 1u << 4;
        }
        if (i >= 0) { if (n == 0 || 2 != (int)(cigar[n - 1] & 0xf)) { if (n >= cap) { S.err |= ERR_CIGAR_CAP; return score; } cigar[n++] = (uint32_t)(i + 1) << 4 | 2; } else cigar[n - 1] += (uint32_t)(i + 1) << 4; }
        if (k >= 0) { if (n == 0 || 1 != (int)(cigar[n - 1] & 0xf)) { if (n >= cap) { S.err |= ERR_CIGAR_CAP; return score; } cigar[n++] = (uint32_t)(k + 1) << 4 | 1; } else cigar[n - 1] += (uint32_t)(k + 1) << 4; }
        for (i = 0; i < n >> 1; ++i) { uint32_t tmp = cigar[i]; cigar[i] = cigar[n - 1 - i]; cigar[n - 1 - i] = tmp; }
        *n_cigar_ = n;
    }
    return score;
}

// The same alignment, score only, by a whole wavefront: for callers in which all 64 lanes of a one-wave workgroup run this
// code in lockstep on the same read (k_post1_wave, long reads), where a lane-serial DP over thousands of rows would leave
// 63 lanes idle.  Rows are rings in LDS around the band (cf. ExtLds in k_extend.hip); row i writes index i + w + 1 before
// row i + 1 reads it, so only indices 0 .. w + 1 are initialised here.  Same recurrence and tie rules as global_dp.
struct WaveDp { int32_t* eh_h; int32_t* eh_e; int32_t* tmpM; int rm; int lane; bool no_pk; };   // no_pk: the rows-in-LDS form only (BWAMEM_HIP_DEBUGK bit 0x10000: tests)
DEV int global_score_wave(const DevIndex& ix, const MemOpt& opt, const WaveDp& L, const SeqAcc& A, int w)
{
    const int lane = L.lane, RM = L.rm;
    const int qlen = A.qlen, tlen = A.tlen;
    const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int NEG = -2000000000;
    const ScoreTab ST = score_tab(opt);
    __syncthreads();
    for (int j = lane; j <= qlen && j <= w + 1; j += WAVE) {
        L.eh_h[j & RM] = j == 0 ? 0 : (j <= w ? -(o_ins + e_ins * j) : MINUS_INF);
        L.eh_e[j & RM] = MINUS_INF;
    }
    int tch = 4;
    __syncthreads();
    for (int i = 0; i < tlen; ++i) {
        if ((i & 63) == 0) { int ii = i + lane; tch = ii < tlen ? acc_t(ix, A, ii) : 4; }
        const int tb = wave_bcast(tch, i & 63);
        const int ms0 = score_at(ST.p[0], ST.n[0], tb), ms1 = score_at(ST.p[1], ST.n[1], tb), ms2 = score_at(ST.p[2], ST.n[2], tb), ms3 = score_at(ST.p[3], ST.n[3], tb), ms4 = score_at(ST.p[4], ST.n[4], tb);
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        const int h1i = beg == 0 ? -(o_del + e_del * (i + 1)) : MINUS_INF;
        for (int c = beg; c < end; c += WAVE) {                     // M(i,j) from row i-1, before any cell of the row is overwritten
            const int j = c + lane;
            if (j < end) {
                const int qc = acc_q(A, j);
                const int sc = qc == 0 ? ms0 : qc == 1 ? ms1 : qc == 2 ? ms2 : qc == 3 ? ms3 : ms4;
                L.tmpM[j & RM] = L.eh_h[j & RM] + sc;
            }
        }
        __syncthreads();
        int fcarry = MINUS_INF;
        for (int c = beg; c < end; c += WAVE) {                     // F by scan, H, E
            const int j = c + lane;
            const bool act = j < end;
            const int m = act ? L.tmpM[j & RM] : 0;
            const int e = act ? L.eh_e[j & RM] : 0;
            const int tins = m - oe_ins;
            const int U = act ? tins + j * e_ins : NEG;
            const int P = wave_prefix_max(U, lane);
            const int Pex = __shfl_up(P, 1);
            int f = fcarry - (j - c) * e_ins;
            if (lane > 0) { const int g = Pex - (j - 1) * e_ins; f = f > g ? f : g; }
            int h = m >= e ? m : e;
            h = h >= f ? h : f;
            const int t = m - oe_del;
            int e2 = e - e_del;
            e2 = e2 > t ? e2 : t;
            const int Plast = wave_bcast(P, 63);
            { const int f1 = fcarry - WAVE * e_ins, f2 = Plast - (c + 63) * e_ins; fcarry = f1 > f2 ? f1 : f2; }
            if (act) { L.eh_e[j & RM] = e2; L.eh_h[(j + 1) & RM] = h; }
        }
        if (lane == 0) {
            if (end > beg) L.eh_h[beg & RM] = h1i;
            else L.eh_h[end & RM] = h1i;
            L.eh_e[end & RM] = MINUS_INF;
        }
        __syncthreads();
    }
    const int score = L.eh_h[qlen & RM];
    __syncthreads();
    return score;
}

// NM and MD of the CIGAR in S.cig[1..1+n_cig) (the tail of upstream bwa_gen_cigar2): MD lands NUL-terminated in S.md
DEV void cigar_nm_md(const DevIndex& ix, PostScratch& S, const SeqAcc& A, int n_cig, bool fwd_strand, int* NM, int* l_md)
{
    const uint32_t* cigar = S.cig + 1;
    int k, x, y, u, n_mm = 0, n_gap = 0;
    const char* int2base = fwd_strand ? "ACGTN" : "TGCAN";
    MdBuf md; md.s = S.md; md.cap = S.md_cap; md.l = 0; md.ovf = false;
    SeqCache sc; seq_cache_init(sc);
    for (k = 0, x = y = u = 0; k < n_cig; ++k) {
        int op = cigar[k] & 0xf, len = (int)(cigar[k] >> 4);
        if (op == 0) {
            for (int i = 0; i < len; ++i) {
                int tb = acc_t_c(ix, A, sc, y + i);
                if (acc_q_c(A, sc, x + i) != tb) { md.putw(u); md.putc(int2base[tb]); ++n_mm; u = 0; }
                else ++u;
            }
            x += len; y += len;
        } else if (op == 2) {
            if (k > 0 && k < n_cig - 1) {            // not for a leading / trailing D
                md.putw(u); md.putc('^');
                for (int i = 0; i < len; ++i) md.putc(int2base[acc_t_c(ix, A, sc, y + i)]);
                u = 0; n_gap += len;
            }
            y += len;
        } else if (op == 1) { x += len; n_gap += len; }
    }
    md.putw(u);
    md.s[md.l] = 0;
    if (md.ovf) S.err |= ERR_CIGAR_CAP;
    *NM = n_mm + n_gap;
    if (l_md) *l_md = md.l;
}

// bwa_gen_cigar2: returns false when upstream would return a NULL cigar.  CIGAR lands in
// S.cig[1..1+n), MD in S.md (NUL-terminated, length *l_md) when want_cigar.
DEV bool gen_cigar2(const DevIndex& ix, const MemOpt& opt, PostScratch& S, int w_, int l_query, const uint8_t* query,
                    int64_t rb, int64_t re, int* score, bool want_cigar, int* n_cigar, int* NM, int* l_md = 0, const WaveDp* wd = nullptr)
{
    const int64_t l_pac = ix.l_pac;
    if (n_cigar) *n_cigar = 0;
    if (NM) *NM = -1;
    if (l_md) *l_md = 0;
    if (want_cigar) S.md[0] = 0;
    if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return false;
    if (re > l_pac << 1 || rb < 0) return false;       // upstream: clipped fetch length != re - rb
    SeqAcc A; A.q = query; A.qlen = l_query; A.rev = rb >= l_pac; A.t0 = rb; A.tlen = (int)(re - rb);
    const int rlen = A.tlen;
    int n_cig = 0;
    if (l_query == re - rb && w_ == 0) {               // no gap: no DP
        int sc = 0;
        const ScoreTab ST = score_tab(opt);
        SeqCache cc; seq_cache_init(cc);
        for (int i = 0; i < l_query; ++i) {
            uint32_t sp; int sn;
            score_lane(ST, acc_q_c(A, cc, i), sp, sn);
            sc += score_at(sp, sn, acc_t_c(ix, A, cc, i));
        }
        *score = sc;
        if (want_cigar) { S.cig[1] = (uint32_t)l_query << 4; n_cig = 1; }
    } else {
        int w, max_gap, max_ins, max_del, min_w, d;
        max_ins = div_plus(((l_query + 1) >> 1) * opt.mat[0] - opt.o_ins, opt.e_ins, 1);
        max_del = div_plus(((l_query + 1) >> 1) * opt.mat[0] - opt.o_del, opt.e_del, 1);
        max_gap = max_ins > max_del ? max_ins : max_del;
        max_gap = max_gap > 1 ? max_gap : 1;
        d = rlen - l_query; d = d < 0 ? -d : d;
        w = (max_gap + d + 1) >> 1;
        w = w < w_ ? w : w_;
        min_w = d + 3;
        w = w > min_w ? w : min_w;
        // a wavefront running this in lockstep aligns across its lanes when the band fits its rings (score only)
        bool done = false;
        if (wd && !want_cigar && !wd->no_pk) {                     // the packed diagonal form first (global_pk.h): no rows in LDS, no barriers
            const int nch = (2 * w + 1 + 63) >> 6, l_end = l_query - 1 - (rlen - 1 - w);
            GpkFit fit;
            if (l_end >= 0 && l_end <= 2 * w && nch <= 14 && gpk_fit(opt, w, (nch + 1) >> 1, fit)) {
                int sc;
                if (nch <= 2) sc = global_wave_diag_pk<1, false>(ix, opt, nullptr, wd->lane, A, w, fit, nullptr, false, 0, done);
                else if (nch <= 4) sc = global_wave_diag_pk<2, false>(ix, opt, nullptr, wd->lane, A, w, fit, nullptr, false, 0, done);
                else if (nch <= 8) sc = global_wave_diag_pk<4, false>(ix, opt, nullptr, wd->lane, A, w, fit, nullptr, false, 0, done);
                else sc = global_wave_diag_pk<7, false>(ix, opt, nullptr, wd->lane, A, w, fit, nullptr, false, 0, done);
                if (done) *score = sc;
            }
        }
        if (done) { }
        else if (wd && !want_cigar && (2 * w + 4 <= wd->rm + 1 || l_query + 2 <= wd->rm + 1)) *score = global_score_wave(ix, opt, *wd, A, w);
        else *score = global_dp(ix, opt, S, A, w, want_cigar, &n_cig);
    }
    if (want_cigar) {
        if (n_cigar) *n_cigar = n_cig;
        if (NM) cigar_nm_md(ix, S, A, n_cig, rb < l_pac, NM, l_md);
    }
    return true;
}

DEV int approx_mapq_se(const DevIndex& ix, const MemOpt& opt, PostScratch& S, const AlnReg& a)
{
    int mapq, l, sub = a.sub ? a.sub : opt.min_seed_len * opt.a;
    double identity;
    sub = a.csub > sub ? a.csub : sub;
    if (sub >= a.score) return 0;
    l = a.qe - a.qb > a.re - a.rb ? a.qe - a.qb : (int)(a.re - a.rb);
    identity = 1. - (double)(l * opt.a - a.score) / (opt.a + opt.b) / l;
    if (l < 0 || l >= ix.log_tab_n || a.seedcov < 0 || a.seedcov >= ix.log_tab_n || a.sub_n + 1 >= ix.log_tab_n) { S.err |= ERR_SCRATCH; return 0; }
    if (a.score == 0) {
        mapq = 0;
    } else if (opt.mapQ_coef_len > 0) {
        double tmp;
        tmp = l < opt.mapQ_coef_len ? 1. : opt.mapQ_coef_fac / ix.log_tab[l];
        tmp *= identity * identity;
        mapq = (int)(6.02 * (a.score - sub) / opt.a * tmp * tmp + .499);
    } else {
        mapq = (int)(30.0 * (1. - (double)sub / a.score) * ix.log_tab[a.seedcov] + .499);
        mapq = identity < 0.95 ? (int)(mapq * identity * identity + .499) : mapq;
    }
    if (a.sub_n > 0) mapq -= (int)(4.343 * ix.log_tab[a.sub_n + 1] + .499);
    if (mapq > 60) mapq = 60;
    if (mapq < 0) mapq = 0;
    mapq = (int)(mapq * (1. - a.frac_rep) + .499);
    return mapq;
}

// CIGARs computed ahead by the wave-parallel global-alignment kernel (k_cigar.hip) for the regions that need DP
struct DpJob { int32_t read, reg; };
struct DpOut { int32_t score, n_cigar; };
struct JobView { const DpOut* out; const uint32_t* cig; int cig_cap; };

DEV int infer_bw(int l1, int l2, int score, int a, int q, int r)
{
    int w, d;
    if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
    w = div_plus((l1 < l2 ? l1 : l2) * a - score - q, r, 2);
    d = l1 - l2; d = d < 0 ? -d : d;
    if (w < d) w = d;
    return w;
}

// first band of mem_reg2aln's retry loop
DEV int first_w2(const MemOpt& opt, const AlnReg& ar)
{
    int tmp = infer_bw(ar.qe - ar.qb, (int)(ar.re - ar.rb), ar.truesc, opt.a, opt.o_del, opt.e_del);
    int w2  = infer_bw(ar.qe - ar.qb, (int)(ar.re - ar.rb), ar.truesc, opt.a, opt.o_ins, opt.e_ins);
    w2 = w2 > tmp ? w2 : tmp;
    if (w2 > opt.w) w2 = w2 < ar.w ? w2 : ar.w;
    return w2 < opt.w << 2 ? w2 : opt.w << 2;
}
// does mem_reg2aln run a banded global alignment for this region (false: the ungapped shortcut of bwa_gen_cigar2)?
DEV bool region_needs_dp(const MemOpt& opt, const AlnReg& ar)
{
    if (ar.rb < 0 || ar.re < 0) return false;
    return !(ar.qe - ar.qb == ar.re - ar.rb && first_w2(opt, ar) == 0);
}

// mem_reg2aln; ar == 0 gives the unmapped record.  jv != 0: regions flagged in pad_ take their CIGAR from the job pool.
// light: only position, strand, contig and reference span are wanted (the mate fields of the other read's record), so
// the per-base NM / MD / score walks are skipped; NM, MD and mapq of the result are then not meaningful
DEV AlnRec reg2aln(const DevIndex& ix, const MemOpt& opt, PostScratch& S, int l_query, const uint8_t* query, const AlnReg* ar, const JobView* jv = 0, bool light = false)
{
    AlnRec a;
    a.pos = 0; a.rid = 0; a.flag = 0; a.is_rev = 0; a.is_alt = 0; a.mapq = 0; a.NM = 0; a.n_cigar = 0;
    a.cigar = S.cig + 1; a.md = S.md; a.l_md = 0; a.score = a.sub = a.alt_sc = 0; a.ref_len = 0;
    S.md[0] = 0;
    if (ar == 0 || ar->rb < 0 || ar->re < 0) { a.rid = -1; a.pos = -1; a.flag |= 0x4; return a; }
    int i, w2, tmp, NM = -1, score = 0, is_rev, last_sc = -(1 << 30), n_cigar = 0, l_md = 0;
    const int qb = ar->qb, qe = ar->qe;
    const int64_t rb = ar->rb, re = ar->re;
    a.mapq = ar->secondary < 0 ? approx_mapq_se(ix, opt, S, *ar) : 0;
    if (ar->secondary >= 0) a.flag |= 0x100;
    tmp = infer_bw(qe - qb, (int)(re - rb), ar->truesc, opt.a, opt.o_del, opt.e_del);
    w2  = infer_bw(qe - qb, (int)(re - rb), ar->truesc, opt.a, opt.o_ins, opt.e_ins);
    w2 = w2 > tmp ? w2 : tmp;
    if (w2 > opt.w) w2 = w2 < ar->w ? w2 : ar->w;
    if (jv && ar->pad_ > 0) {                               // CIGAR was produced by k_gcigar (same retry loop, wave-parallel)
        const int job = ar->pad_ - 1;
        n_cigar = jv->out[job].n_cigar;
        if (n_cigar + 2 > S.cig_cap) { S.err |= ERR_CIGAR_CAP; n_cigar = 0; }
        const uint32_t* src = jv->cig + (size_t)job * jv->cig_cap;
        for (i = 0; i < n_cigar; ++i) S.cig[1 + i] = src[i];
        SeqAcc A; A.q = query + qb; A.qlen = qe - qb; A.rev = rb >= ix.l_pac; A.t0 = rb; A.tlen = (int)(re - rb);
        if (!light) cigar_nm_md(ix, S, A, n_cigar, rb < ix.l_pac, &NM, &l_md);
    } else if (light && !region_needs_dp(opt, *ar) && !(qe - qb <= 0 || rb >= re || (rb < ix.l_pac && re > ix.l_pac) || re > ix.l_pac << 1)) {
        S.cig[1] = (uint32_t)(qe - qb) << 4; n_cigar = 1;   // bwa_gen_cigar2's no-gap case: one M run
    } else {
        i = 0;
        do {
            w2 = w2 < opt.w << 2 ? w2 : opt.w << 2;
            gen_cigar2(ix, opt, S, w2, qe - qb, query + qb, rb, re, &score, true, &n_cigar, &NM, &l_md);
            if (score == last_sc || w2 == opt.w << 2) break;
            last_sc = score;
            w2 <<= 1;
        } while (++i < 3 && score < ar->truesc - opt.a);
    }
    a.NM = NM & 0x3fffff;                                   // NM:22 bit-field upstream
    int64_t pos = bns_depos(ix, rb < ix.l_pac ? rb : re - 1, is_rev);
    a.is_rev = is_rev;
    uint32_t* cig = S.cig + 1;
    if (n_cigar > 0) {                                      // squeeze out a leading or trailing deletion
        if ((cig[0] & 0xf) == 2) { pos += cig[0] >> 4; --n_cigar; ++cig; }
        else if ((cig[n_cigar - 1] & 0xf) == 2) --n_cigar;
    }
    if (qb != 0 || qe != l_query) {                         // soft clips (op 3 upstream)
        int clip5 = is_rev ? l_query - qe : qb;
        int clip3 = is_rev ? qb : l_query - qe;
        if (clip5) { --cig; cig[0] = (uint32_t)clip5 << 4 | 3; ++n_cigar; }
        if (clip3) { if (n_cigar + (int)(cig - S.cig) >= S.cig_cap) { S.err |= ERR_CIGAR_CAP; } else cig[n_cigar++] = (uint32_t)clip3 << 4 | 3; }
    }
    a.cigar = cig; a.n_cigar = n_cigar; a.md = S.md; a.l_md = l_md;
    for (i = 0; i < n_cigar; ++i) { int op = cig[i] & 0xf; if (op == 0 || op == 2) a.ref_len += (int)(cig[i] >> 4); }
    a.rid = bns_pos2rid(ix, pos);
    a.pos = pos - ix.ann_offset[a.rid];
    a.score = ar->score; a.sub = ar->sub > ar->csub ? ar->sub : ar->csub;
    a.is_alt = ar->is_alt; a.alt_sc = ar->alt_sc;
    return a;
}

// ------------------------------------------------------------------ sort comparators
struct RegReLt { __device__ bool operator()(const AlnReg& a, const AlnReg& b) const { return a.re < b.re; } };
struct RegSLt  { __device__ bool operator()(const AlnReg& a, const AlnReg& b) const {
    return a.score > b.score || (a.score == b.score && (a.rb < b.rb || (a.rb == b.rb && a.qb < b.qb))); } };
struct RegHLt  { __device__ bool operator()(const AlnReg& a, const AlnReg& b) const {
    return a.score > b.score || (a.score == b.score && (a.is_alt < b.is_alt || (a.is_alt == b.is_alt && a.hash < b.hash))); } };
struct RegHLt2 { __device__ bool operator()(const AlnReg& a, const AlnReg& b) const {
    return a.is_alt < b.is_alt || (a.is_alt == b.is_alt && (a.score > b.score || (a.score == b.score && a.hash < b.hash))); } };

// Sorting regions by key records.  A read in a repeat family carries hundreds of 96-byte regions and upstream sorts them four
// times (introsort, unstable: the permutation of tied regions is part of the result).  ks_introsort decides by comparisons
// only, so sorting 24-byte records {key, index} whose order is isomorphic to the comparator's gives the same permutation, and
// the regions then move once each (cycle by cycle) instead of once per swap.  Human-like genome: k_post1 565 -> XXX ms,
// k_final_prep's primary marking 346 -> XXX ms per 10 M reads.
struct SortKey { uint64_t k0, k1; uint32_t k2; int32_t idx; };
struct SortKeyLt { __device__ bool operator()(const SortKey& a, const SortKey& b) const {
    return a.k0 < b.k0 || (a.k0 == b.k0 && (a.k1 < b.k1 || (a.k1 == b.k1 && a.k2 < b.k2))); } };
DEV uint32_t key_asc(int32_t v) { return (uint32_t)v ^ 0x80000000u; }           // signed order -> unsigned order
DEV uint32_t key_desc(int32_t v) { return ~((uint32_t)v ^ 0x80000000u); }
DEV uint64_t key_asc64(int64_t v) { return (uint64_t)v ^ 0x8000000000000000ull; }
DEV SortKey sort_key(const AlnReg& r, RegReLt) { SortKey k; k.k0 = key_asc64(r.re); k.k1 = 0; k.k2 = 0; k.idx = 0; return k; }
DEV SortKey sort_key(const AlnReg& r, RegSLt)  { SortKey k; k.k0 = key_desc(r.score); k.k1 = key_asc64(r.rb); k.k2 = key_asc(r.qb); k.idx = 0; return k; }
DEV SortKey sort_key(const AlnReg& r, RegHLt)  { SortKey k; k.k0 = (uint64_t)key_desc(r.score) << 32 | key_asc(r.is_alt); k.k1 = r.hash; k.k2 = 0; k.idx = 0; return k; }
DEV SortKey sort_key(const AlnReg& r, RegHLt2) { SortKey k; k.k0 = (uint64_t)key_asc(r.is_alt) << 32 | key_desc(r.score); k.k1 = r.hash; k.k2 = 0; k.idx = 0; return k; }
#define SORT_BY_KEY_MIN 12
// room for a read's key records: its share of the chaining B-tree's node pool (40 bytes per seed + 480; free once k_chain is done)
DEV SortKey* sort_keys_for(const TileView& tv, int r) { return (SortKey*)(tv.bt_nodes + (tv.seed_off[r] / 4 + 3 * (int64_t)r) * BT_NODE_INTS); }

template <typename LT>
DEV void sort_regs(int n, AlnReg* a, SortKey* keys, LT lt)
{
    if (!keys || n < SORT_BY_KEY_MIN) { ks_introsort((size_t)n, a, lt); return; }
    for (int i = 0; i < n; ++i) { SortKey k = sort_key(a[i], lt); k.idx = i; keys[i] = k; }
    ks_introsort((size_t)n, keys, SortKeyLt());
    for (int i = 0; i < n; ++i) {                      // a[i] <- a[keys[i].idx], one cycle of the permutation at a time
        int src = keys[i].idx;
        if (src == i) continue;
        const AlnReg first = a[i];
        int j = i;
        while (src != i) { a[j] = a[src]; keys[j].idx = j; j = src; src = keys[j].idx; }
        a[j] = first; keys[j].idx = j;
    }
}

// mem_patch_reg: can two colinear regions be merged into one global alignment?
DEV int patch_reg(const DevIndex& ix, const MemOpt& opt, PostScratch& S, const uint8_t* query, const AlnReg& a, const AlnReg& b, int* _w, const WaveDp* wd = nullptr)
{
    if (a.rb < ix.l_pac && b.rb >= ix.l_pac) return 0;
    if (a.qb >= b.qb || a.qe >= b.qe || a.re >= b.re) return 0;
    int w = (int)((a.re - b.rb) - (a.qe - b.qb));
    w = w > 0 ? w : -w;
    double r = (double)(a.re - b.rb) / (b.re - a.rb) - (double)(a.qe - b.qb) / (b.qe - a.qb);
    r = r > 0. ? r : -r;
    if (a.re < b.rb || a.qe < b.qb) {
        if (w > opt.w << 1 || r >= 0.05f) return 0;
    } else if (w > opt.w << 2 || r >= 0.05f * 2) return 0;
    w += a.w + b.w;
    w = w < opt.w << 2 ? w : opt.w << 2;
    int score = 0;
    gen_cigar2(ix, opt, S, w, b.qe - a.qb, query + a.qb, a.rb, b.re, &score, false, 0, 0, 0, wd);
    int q_s = (int)((double)(b.qe - a.qb) / ((b.qe - b.qb) + (a.qe - a.qb)) * (b.score + a.score) + .499);
    int r_s = (int)((double)(b.re - a.rb) / ((b.re - b.rb) + (a.re - a.rb)) * (b.score + a.score) + .499);
    if ((double)score / (q_s > r_s ? q_s : r_s) < 0.90f) return 0;
    *_w = w;
    return score;
}

// mem_sort_dedup_patch; query == 0 disables patching (the mate-rescue caller)
DEV int sort_dedup_patch(const DevIndex& ix, const MemOpt& opt, PostScratch& S, const uint8_t* query, int n, AlnReg* a, int dbg = 0, SortKey* keys = nullptr)
{
    int m, i, j;
    if (n <= 1) return n;
    if (dbg & 0xff) printf("[k] sdp: n=%d before sort1\n", n);
    sort_regs(n, a, keys, RegReLt());
    for (i = 0; i < n; ++i) a[i].n_comp = 1;
    for (i = 1; i < n; ++i) {
        AlnReg* p = &a[i];
        if (p->rid != a[i - 1].rid || p->rb >= a[i - 1].re + opt.max_chain_gap) continue;
        for (j = i - 1; j >= 0 && p->rid == a[j].rid && p->rb < a[j].re + opt.max_chain_gap; --j) {
            AlnReg* q = &a[j];
            int64_t orr, oq, mr, mq;
            int score, w;
            if (q->qe == q->qb) continue;
            orr = q->re - p->rb;
            oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
            mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
            mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
            if ((float)orr > opt.mask_level_redun * (float)mr && (float)oq > opt.mask_level_redun * (float)mq) {
                if (p->score < q->score) { p->qe = p->qb; break; }
                else q->qe = q->qb;
            } else if (query && q->rb < p->rb && (score = patch_reg(ix, opt, S, query, *q, *p, &w)) > 0) {
                p->n_comp += q->n_comp + 1;
                p->seedcov = p->seedcov > q->seedcov ? p->seedcov : q->seedcov;
                p->sub = p->sub > q->sub ? p->sub : q->sub;
                p->csub = p->csub > q->csub ? p->csub : q->csub;
                p->qb = q->qb; p->rb = q->rb;
                p->truesc = p->score = score;
                p->w = w;
                q->qb = q->qe;
            }
        }
    }
    for (i = 0, m = 0; i < n; ++i)
        if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
    n = m;
    sort_regs(n, a, keys, RegSLt());
    for (i = 1; i < n; ++i)
        if (a[i].score == a[i - 1].score && a[i].rb == a[i - 1].rb && a[i].qb == a[i - 1].qb)
            a[i].qe = a[i].qb;
    for (i = 1, m = 1; i < n; ++i)
        if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
    return m;
}


// The redundancy test of mem_sort_dedup_patch's inner loop for q before p in end order (the loop's window rule included)
DEV bool sdp_redundant(const MemOpt& opt, const AlnReg& q, const AlnReg& p)
{
    if (p.rid != q.rid || p.rb >= q.re + opt.max_chain_gap) return false;
    const int64_t orr = q.re - p.rb;
    const int64_t oq = q.qb < p.qb ? q.qe - p.qb : p.qe - q.qb;
    const int64_t mr = q.re - q.rb < p.re - p.rb ? q.re - q.rb : p.re - p.rb;
    const int64_t mq = q.qe - q.qb < p.qe - p.qb ? q.qe - q.qb : p.qe - p.qb;
    return (float)orr > opt.mask_level_redun * (float)mr && (float)oq > opt.mask_level_redun * (float)mq;
}
DEV bool sdp_redundant_with(const MemOpt& opt, const AlnReg& x, const AlnReg& b)      // x from the list against the new region b
{
    return x.re < b.re ? sdp_redundant(opt, x, b) : sdp_redundant(opt, b, x);
}

// mem_matesw calls mem_sort_dedup_patch(query = 0) on the mate's whole hit list after every rescue attempt: sort by end,
// drop one region of every redundant pair met in that order, sort by (score desc, rb, qb), drop exact duplicates.  A pair in
// a repeat family does that a hundred times on hundreds of regions.  What a call *returns* is cheap to say once the list
// has been through one call (`settled`):
//  * its output is strictly ordered by (score desc, rb, qb) -- two regions with equal (rb, qb) are redundant, and a region
//    that survives the loop has been tested against every survivor in its window, so no two survivors are redundant (the
//    test is a property of the pair: for regions with equal ends it comes out the same either way round) -- i.e. the
//    list between calls is a *set* in canonical order, whatever the unstable sorts did with ties;
//  * a call on a settled list to which nothing was added changes nothing;
//  * a call on a settled list plus one new region b only ever fires on pairs that contain b: in end order b first walks
//    down (it removes every redundant region until one with a higher score removes b), then the regions above b meet it
//    (a lower-scoring one is removed, the first other one removes b).  Let C be the regions redundant with b.  If b
//    outscores all of C, C goes and b stays; if all of C outscore b, b goes; both whatever the order.  Otherwise the
//    order matters and it is the end order, well defined when the ends in C and b are distinct; if they are not (the tie
//    order of upstream's introsort decides) this function declines and the caller runs the procedure itself.
// Returns false when it declines (nothing modified).
DEV bool matesw_insert(const MemOpt& opt, const AlnReg& b, int& n_ma, AlnReg* ma)
{
    if (b.qe <= b.qb || b.re <= b.rb) return false;
    int n_c = 0, hi = -INT_MAX_, lo = INT_MAX_;
    bool tie_b = false;
    for (int k = 0; k < n_ma; ++k) {
        const AlnReg& x = ma[k];
        if (x.rid != b.rid) continue;
        if (sdp_redundant_with(opt, x, b)) {
            ++n_c; hi = hi > x.score ? hi : x.score; lo = lo < x.score ? lo : x.score;
            tie_b |= x.re == b.re;
        }
    }
    bool b_alive = true;
    int n_kill = 0;
    if (n_c == 0) {
    } else if (b.score > hi) {
        for (int k = 0; k < n_ma; ++k)
            if (ma[k].rid == b.rid && sdp_redundant_with(opt, ma[k], b)) { ma[k].qe = ma[k].qb; ++n_kill; }
    } else if (b.score < lo) {
        b_alive = false;
    } else {
        if (tie_b || n_c > 16) return false;
        for (int k = 0; k < n_ma; ++k) {                        // two members of C with the same end: their order is the sort's
            if (ma[k].rid != b.rid || !sdp_redundant_with(opt, ma[k], b)) continue;
            for (int l = k + 1; l < n_ma; ++l)
                if (ma[l].re == ma[k].re && ma[l].rid == b.rid && sdp_redundant_with(opt, ma[l], b)) return false;
        }
        int64_t cur = b.re;
        for (;;) {                                              // b's own walk down the end order
            int best = -1;
            for (int k = 0; k < n_ma; ++k) {
                const AlnReg& x = ma[k];
                if (x.rid != b.rid || x.qe <= x.qb || x.re >= cur || !sdp_redundant(opt, x, b)) continue;
                if (best < 0 || x.re > ma[best].re) best = k;
            }
            if (best < 0) break;
            if (b.score < ma[best].score) { b_alive = false; break; }
            cur = ma[best].re; ma[best].qe = ma[best].qb; ++n_kill;
        }
        cur = b.re;
        while (b_alive) {                                       // the regions above b, nearest end first
            int best = -1;
            for (int k = 0; k < n_ma; ++k) {
                const AlnReg& x = ma[k];
                if (x.rid != b.rid || x.qe <= x.qb || x.re <= cur || !sdp_redundant(opt, b, x)) continue;
                if (best < 0 || x.re < ma[best].re) best = k;
            }
            if (best < 0) break;
            if (ma[best].score < b.score) { cur = ma[best].re; ma[best].qe = ma[best].qb; ++n_kill; }
            else b_alive = false;
        }
    }
    int m = n_ma;
    if (n_kill) {
        m = 0;
        for (int k = 0; k < n_ma; ++k)
            if (ma[k].qe > ma[k].qb) { if (m != k) ma[m] = ma[k]; ++m; }
    }
    if (b_alive) {
        int pos = 0;                                            // canonical place: behind every region that sorts before b
        while (pos < m && (ma[pos].score > b.score || (ma[pos].score == b.score && (ma[pos].rb < b.rb || (ma[pos].rb == b.rb && ma[pos].qb <= b.qb))))) ++pos;
        for (int k = m; k > pos; --k) ma[k] = ma[k - 1];
        ma[pos] = b; ma[pos].n_comp = 1;
        ++m;
    }
    n_ma = m;
    return true;
}


// ---- the same for a pair with hundreds of regions, by the whole wavefront (k_pe.hip: k_pe_matesw_wave).  Called by all 64
// lanes of a one-wave workgroup with uniform arguments; the list is in global memory, so a barrier stands between a lane's
// stores and another lane's loads of the same record.
DEV bool reg_sorts_before(const AlnReg& x, const AlnReg& b)       // x stays in front of b in (score desc, rb, qb) order
{
    return x.score > b.score || (x.score == b.score && (x.rb < b.rb || (x.rb == b.rb && x.qb <= b.qb)));
}
// open a slot at pos in a[0 .. m) (records move up by one, 64 at a time from the top) and put b there
DEV void regs_insert_wave(AlnReg* a, int m, int pos, const AlnReg& b, int lane)
{
    for (int top = m; top > pos; top -= 64) {
        const int lo = top - 64 > pos ? top - 64 : pos, k = lo + lane;
        AlnReg v;
        if (k < top) v = a[k];
        __syncthreads();
        if (k < top) a[k + 1] = v;
    }
    __syncthreads();
    if (lane == 0) a[pos] = b;
    __syncthreads();
}
// drop the records whose bit is set in the lanes' marks (record 64 c + lane <-> bit c of that lane's mask); returns the new count
DEV int regs_remove_wave(AlnReg* a, int n, uint64_t marks, int lane)
{
    int m = 0;
    for (int base = 0, c = 0; base < n; base += 64, ++c) {
        const int k = base + lane;
        const bool keep = k < n && !(marks >> c & 1);
        const uint64_t bal = __ballot(keep);
        const int cnt = __popcll(bal);
        if (m != base || cnt != (n - base < 64 ? n - base : 64)) {          // (nothing above has gone yet: the records stay where they are)
            AlnReg v;
            if (keep) v = a[k];
            __syncthreads();
            if (keep) a[m + __popcll(bal & ((1ull << lane) - 1))] = v;
        }
        m += cnt;
    }
    __syncthreads();
    return m;
}
DEV bool matesw_insert_wave(const MemOpt& opt, const AlnReg& b, int& n_ma, AlnReg* ma, int lane)
{
    if (b.qe <= b.qb || b.re <= b.rb || n_ma > 4096) return false;
    uint64_t mine = 0;                                              // this lane's members of C, one bit per 64-record chunk
    int n_c = 0, hi = -INT_MAX_, lo = INT_MAX_, front_c = 0, front_nc = 0;
    for (int base = 0, c = 0; base < n_ma; base += 64, ++c) {
        const int k = base + lane;
        bool red = false, front = false;
        if (k < n_ma) {
            const AlnReg& x = ma[k];
            red = x.rid == b.rid && sdp_redundant_with(opt, x, b);
            front = reg_sorts_before(x, b);
            if (red) { mine |= 1ull << c; hi = hi > x.score ? hi : x.score; lo = lo < x.score ? lo : x.score; }
        }
        n_c += __popcll(__ballot(red));
        front_c += __popcll(__ballot(red && front));
        front_nc += __popcll(__ballot(!red && front));
    }
    if (n_c) { hi = wave_max(hi); lo = -wave_max(-lo); }
    if (n_c && !(b.score > hi) && !(b.score < lo)) {               // the order of C decides: rare, one lane walks it
        int ok = 0, m = n_ma;
        __syncthreads();
        if (lane == 0) ok = matesw_insert(opt, b, m, ma) ? 1 : 0;
        ok = __shfl(ok, 0); m = __shfl(m, 0);
        __syncthreads();
        if (ok) n_ma = m;
        return ok != 0;
    }
    if (n_c && b.score < lo) return true;                           // b goes, the list stays
    int m = n_ma, pos = front_nc + front_c;
    if (n_c) { m = regs_remove_wave(ma, n_ma, mine, lane); pos = front_nc; }
    AlnReg bb = b; bb.n_comp = 1;
    regs_insert_wave(ma, m, pos, bb, lane);
    n_ma = m + 1;
    return true;
}

// mem_sort_dedup_patch without a query, by the wavefront: the two sorts on lane 0 (upstream's introsort, whose tie order is
// part of the result), the overlap loop 64 regions of p's window per step -- p is not modified while it walks (no patching), so
// "every redundant region goes until one outscores p" is a ballot --, the compactions by prefix counts
DEV int sort_dedup_nq_wave(const MemOpt& opt, int n, AlnReg* a, SortKey* keys, int lane)
{
    if (n <= 1) return n;
    __syncthreads();
    if (lane == 0) sort_regs(n, a, keys, RegReLt());
    __syncthreads();
    for (int k = lane; k < n; k += 64) a[k].n_comp = 1;
    __syncthreads();
    for (int i = 1; i < n; ++i) {
        const AlnReg& p = a[i];
        const int64_t p_rb = p.rb, p_re = p.re; const int p_qb = p.qb, p_qe = p.qe, p_rid = p.rid, p_score = p.score;
        bool p_dead = false, any = false;
        for (int j0 = i - 1; j0 >= 0; j0 -= 64) {
            const int j = j0 - lane;
            bool inwin = false, red = false, kills_p = false;
            if (j >= 0) {
                const AlnReg& q = a[j];
                const int64_t q_rb = q.rb, q_re = q.re; const int q_qb = q.qb, q_qe = q.qe;
                inwin = q.rid == p_rid && p_rb < q_re + opt.max_chain_gap;
                if (inwin && q_qe != q_qb) {
                    const int64_t orr = q_re - p_rb, oq = q_qb < p_qb ? q_qe - p_qb : p_qe - q_qb;
                    const int64_t mr = q_re - q_rb < p_re - p_rb ? q_re - q_rb : p_re - p_rb, mq = q_qe - q_qb < p_qe - p_qb ? q_qe - q_qb : p_qe - p_qb;
                    red = (float)orr > opt.mask_level_redun * (float)mr && (float)oq > opt.mask_level_redun * (float)mq;
                    kills_p = red && p_score < q.score;
                }
            }
            const uint64_t out = __ballot(!inwin);                  // (lanes below record 0 count as outside)
            uint64_t live = out ? ((1ull << (__ffsll((long long)out) - 1)) - 1) : ~0ull;   // the lanes the walk reaches
            const uint64_t kp = __ballot(kills_p) & live;
            if (kp) { live &= (1ull << (__ffsll((long long)kp) - 1)) - 1; p_dead = true; }
            if (red && (live >> lane & 1)) { a[j].qe = a[j].qb; }
            any |= (__ballot(red) & live) != 0;
            if (out || p_dead) break;
        }
        if (p_dead && lane == 0) a[i].qe = a[i].qb;
        if (any || p_dead) __syncthreads();
    }
    __syncthreads();
    uint64_t marks = 0;
    for (int base = 0, c = 0; base < n; base += 64, ++c) { const int k = base + lane; if (k < n && !(a[k].qe > a[k].qb)) marks |= 1ull << c; }
    int m = n;
    if (n <= 4096) m = regs_remove_wave(a, n, marks, lane);
    else { if (lane == 0) { m = 0; for (int k = 0; k < n; ++k) if (a[k].qe > a[k].qb) { if (m != k) a[m] = a[k]; ++m; } } m = __shfl(m, 0); __syncthreads(); }
    n = m;
    if (lane == 0) {
        sort_regs(n, a, keys, RegSLt());
        for (int k = 1; k < n; ++k)
            if (a[k].score == a[k - 1].score && a[k].rb == a[k - 1].rb && a[k].qb == a[k - 1].qb) a[k].qe = a[k].qb;
        int k; for (k = 1, m = n < 1 ? n : 1; k < n; ++k)
            if (a[k].qe > a[k].qb) { if (m != k) a[m++] = a[k]; else ++m; }
    }
    m = __shfl(m, 0);
    __syncthreads();
    return m;
}

// The same procedure run by a whole wavefront on one read (k_post1<true>, long reads): every lane follows the control flow
// (all decisions are read from memory that only lane 0 writes, with a barrier on either side of each write), lane 0 does the
// updates, and the patch alignments run across the lanes.
DEV int sort_dedup_patch_wave(const DevIndex& ix, const MemOpt& opt, PostScratch& S, const uint8_t* query, int n, AlnReg* a, const WaveDp& wd, SortKey* keys = nullptr)
{
    const bool w0 = wd.lane == 0;
    int m = n, i, j;
    if (n <= 1) return n;
    __syncthreads();
    if (w0) { sort_regs(n, a, keys, RegReLt()); for (i = 0; i < n; ++i) a[i].n_comp = 1; }
    __syncthreads();
    for (i = 1; i < n; ++i) {
        AlnReg* p = &a[i];
        if (p->rid != a[i - 1].rid || p->rb >= a[i - 1].re + opt.max_chain_gap) continue;
        for (j = i - 1; j >= 0 && p->rid == a[j].rid && p->rb < a[j].re + opt.max_chain_gap; --j) {
            AlnReg* q = &a[j];
            int64_t orr, oq, mr, mq;
            int score, w;
            if (q->qe == q->qb) continue;
            orr = q->re - p->rb;
            oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
            mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
            mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
            if ((float)orr > opt.mask_level_redun * (float)mr && (float)oq > opt.mask_level_redun * (float)mq) {
                const bool drop_p = p->score < q->score;
                __syncthreads();
                if (w0) { if (drop_p) p->qe = p->qb; else q->qe = q->qb; }
                __syncthreads();
                if (drop_p) break;
            } else if (query && q->rb < p->rb && (score = patch_reg(ix, opt, S, query, *q, *p, &w, &wd)) > 0) {
                __syncthreads();
                if (w0) {
                    p->n_comp += q->n_comp + 1;
                    p->seedcov = p->seedcov > q->seedcov ? p->seedcov : q->seedcov;
                    p->sub = p->sub > q->sub ? p->sub : q->sub;
                    p->csub = p->csub > q->csub ? p->csub : q->csub;
                    p->qb = q->qb; p->rb = q->rb;
                    p->truesc = p->score = score;
                    p->w = w;
                    q->qb = q->qe;
                }
                __syncthreads();
            }
        }
    }
    __syncthreads();
    if (w0) {
        for (i = 0, m = 0; i < n; ++i)
            if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
        n = m;
        sort_regs(n, a, keys, RegSLt());
        for (i = 1; i < n; ++i)
            if (a[i].score == a[i - 1].score && a[i].rb == a[i - 1].rb && a[i].qb == a[i - 1].qb)
                a[i].qe = a[i].qb;
        for (i = 1, m = n < 1 ? n : 1; i < n; ++i)
            if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
    }
    m = __shfl(m, 0);
    __syncthreads();
    return m;
}

// ------------------------------------------------------------------ primary marking (a14)
DEV void mark_primary_core(const MemOpt& opt, int n, AlnReg* a, int32_t* z)
{
    int i, k, nz = 0, tmp;
    tmp = opt.a + opt.b;
    tmp = opt.o_del + opt.e_del > tmp ? opt.o_del + opt.e_del : tmp;
    tmp = opt.o_ins + opt.e_ins > tmp ? opt.o_ins + opt.e_ins : tmp;
    z[nz++] = 0;
    for (i = 1; i < n; ++i) {
        for (k = 0; k < nz; ++k) {
            int j = z[k];
            int b_max = a[j].qb > a[i].qb ? a[j].qb : a[i].qb;
            int e_min = a[j].qe < a[i].qe ? a[j].qe : a[i].qe;
            if (e_min > b_max) {
                int min_l = a[i].qe - a[i].qb < a[j].qe - a[j].qb ? a[i].qe - a[i].qb : a[j].qe - a[j].qb;
                if ((float)(e_min - b_max) >= (float)min_l * opt.mask_level) {
                    if (a[j].sub == 0) a[j].sub = a[i].score;
                    if (a[j].score - a[i].score <= tmp && (a[j].is_alt || !a[i].is_alt)) ++a[j].sub_n;
                    break;
                }
            }
        }
        if (k == nz) z[nz++] = i;
        else a[i].secondary = z[k];
    }
}

DEV int mark_primary_se(const MemOpt& opt, int n, AlnReg* a, int64_t id, int32_t* z, SortKey* keys = nullptr)
{
    int i, n_pri;
    if (n == 0) return 0;
    for (i = n_pri = 0; i < n; ++i) {
        a[i].sub = a[i].alt_sc = 0; a[i].secondary = a[i].secondary_all = -1; a[i].hash = hash_64((uint64_t)(id + i));
        if (!a[i].is_alt) ++n_pri;
    }
    sort_regs(n, a, keys, RegHLt());
    mark_primary_core(opt, n, a, z);
    for (i = 0; i < n; ++i) {
        AlnReg* p = &a[i];
        p->secondary_all = i;
        if (!p->is_alt && p->secondary >= 0 && a[p->secondary].is_alt) p->alt_sc = a[p->secondary].score;
    }
    if (n_pri >= 0 && n_pri < n) {
        if (n_pri > 0) sort_regs(n, a, keys, RegHLt2());
        for (i = 0; i < n; ++i) z[a[i].secondary_all] = i;
        for (i = 0; i < n; ++i) {
            if (a[i].secondary >= 0) {
                a[i].secondary_all = z[a[i].secondary];
                if (a[i].is_alt) a[i].secondary = INT_MAX_;
            } else a[i].secondary_all = -1;
        }
        if (n_pri > 0) {
            for (i = 0; i < n_pri; ++i) { a[i].sub = 0; a[i].secondary = -1; }
            mark_primary_core(opt, n_pri, a, z);
        }
    } else {
        for (i = 0; i < n; ++i) a[i].secondary_all = a[i].secondary;
    }
    return n_pri;
}

DEV void reorder_primary5(int T, int n, AlnReg* a)
{
    int k, n_pri = 0, left_st = INT_MAX_, left_k = -1;
    for (k = 0; k < n; ++k)
        if (a[k].secondary < 0 && !a[k].is_alt && a[k].score >= T) ++n_pri;
    if (n_pri <= 1) return;
    for (k = 0; k < n; ++k) {
        if (a[k].secondary >= 0 || a[k].is_alt || a[k].score < T) continue;
        if (a[k].qb < left_st) { left_st = a[k].qb; left_k = k; }
    }
    if (left_k == 0) return;
    AlnReg t = a[0]; a[0] = a[left_k]; a[left_k] = t;
    for (k = 1; k < n; ++k) {
        AlnReg* p = &a[k];
        if (p->secondary == 0) p->secondary = left_k;
        else if (p->secondary == left_k) p->secondary = 0;
        if (p->secondary_all == 0) p->secondary_all = left_k;
        else if (p->secondary_all == left_k) p->secondary_all = 0;
    }
}

// ------------------------------------------------------------------ XA (a17, mem_gen_alt)
DEV int get_pri_idx(double XA_drop_ratio, const AlnReg* a, int i)
{
    int k = a[i].secondary_all;
    if (k >= 0 && a[i].score >= a[k].score * XA_drop_ratio) return k;
    return -1;
}

// cnt[] / has_alt[] of mem_gen_alt; returns tot
DEV int xa_prepare(const MemOpt& opt, int n, const AlnReg* a, int32_t* cnt, int32_t* has_alt)
{
    int tot = 0;
    for (int i = 0; i < n; ++i) { cnt[i] = 0; has_alt[i] = 0; }
    for (int i = 0; i < n; ++i) {
        int r = get_pri_idx(opt.XA_drop_ratio, a, i);
        if (r >= 0) { ++cnt[r]; ++tot; if (a[i].is_alt) has_alt[r] = 1; }
    }
    return tot;
}

// append the XA string of primary k to ob; returns its length
DEV int xa_emit(const DevIndex& ix, const MemOpt& opt, PostScratch& S, OutBuf& ob, int l_query, const uint8_t* query,
                int n, const AlnReg* a, const int32_t* cnt, const int32_t* has_alt, int k, const JobView* jv = 0)
{
    int start = ob.len;
    if (cnt[k] == 0) return 0;
    if (cnt[k] > opt.max_XA_hits_alt || (!has_alt[k] && cnt[k] > opt.max_XA_hits)) return 0;
    for (int i = 0; i < n; ++i) {
        if (get_pri_idx(opt.XA_drop_ratio, a, i) != k) continue;
        AlnRec t = reg2aln(ix, opt, S, l_query, query, &a[i], jv);
        const char* nm = ix.names + ix.ann_name_off[t.rid];
        int nl = ix.ann_name_off[t.rid + 1] - ix.ann_name_off[t.rid] - 1;
        for (int j = 0; j < nl; ++j) ob.putc(nm[j]);
        ob.putc(','); ob.putc("+-"[t.is_rev]); ob.putl((long)(t.pos + 1));
        ob.putc(',');
        for (int j = 0; j < t.n_cigar; ++j) { ob.putl((long)(t.cigar[j] >> 4)); ob.putc("MIDSHN"[t.cigar[j] & 0xf]); }
        ob.putc(','); ob.putl(t.NM);
        ob.putc(';');
    }
    return ob.len - start;
}

// ------------------------------------------------------------------ record writer (a18)
// upstream mem_aln2sam's flag prologue followed by the reference's fmt_BAMish (jnibwa.c:43-97).
// xa_* describe how to produce the XA string of this record (k < 0: none).
DEV void aln2out(const DevIndex& ix, const MemOpt& opt, PostScratch& S, OutBuf& ob, int n_recs, int which, AlnRec p, const MateInfo* m_,
                 int l_query, const uint8_t* query, int n_regs, const AlnReg* regs, const int32_t* cnt, const int32_t* has_alt, int xa_k, const JobView* jv = 0)
{
    MateInfo mt; const bool has_m = m_ != 0;
    if (has_m) mt = *m_;
    p.flag |= has_m ? 0x1 : 0;
    p.flag |= p.rid < 0 ? 0x4 : 0;
    p.flag |= has_m && mt.rid < 0 ? 0x8 : 0;
    if (p.rid < 0 && has_m && mt.rid >= 0) { p.rid = mt.rid; p.pos = mt.pos; p.is_rev = mt.is_rev; p.n_cigar = 0; p.ref_len = 0; }
    if (has_m && mt.rid < 0 && p.rid >= 0) { mt.rid = p.rid; mt.pos = p.pos; mt.is_rev = p.is_rev; mt.ref_len = 0; }
    p.flag |= p.is_rev ? 0x10 : 0;
    p.flag |= has_m && mt.is_rev ? 0x20 : 0;
    if (!which) ob.put32(n_recs);
    int32_t flag_mapQ = p.flag;
    if (p.flag & 0x10000) flag_mapQ |= 0x100;
    flag_mapQ = (int32_t)((uint32_t)flag_mapQ << 16) | (p.mapq & 0xff);
    ob.put32(flag_mapQ);
    if (!(p.flag & 0x4)) {
        ob.put32(p.rid);
        ob.put32((int32_t)p.pos);
        ob.put32(p.NM);
        ob.put32(p.score);
        ob.put32(p.sub);
        ob.put32(p.n_cigar);
        for (int i = 0; i < p.n_cigar; ++i) {
            uint32_t lenOp = p.cigar[i];
            if ((lenOp & 0xf) > 2) ++lenOp;             // MIDSH -> BAM MIDNSH
            ob.put32((int32_t)lenOp);
        }
        int nMD = p.n_cigar ? p.l_md : 0;
        ob.put32(nMD);
        for (int i = 0; i < nMD; ++i) ob.putc(p.md[i]);
        ob.pad4();
        int at = ob.len;
        ob.put32(0);                                    // nXA, patched below
        if (xa_k >= 0 && cnt) {
            int nXA = xa_emit(ix, opt, S, ob, l_query, query, n_regs, regs, cnt, has_alt, xa_k, jv);
            ob.pad4();
            if (!ob.ovf) *(int32_t*)(ob.p + at) = nXA;
        }
    }
    if ((p.flag & 0x9) == 1) {
        ob.put32(mt.rid);
        ob.put32((int32_t)mt.pos);
        if ((p.flag & 0x4) || p.rid != mt.rid) ob.put32(0);
        else {                                          // jnibwa.c:82-95
            long p0 = (long)p.pos, m0 = (long)mt.pos;
            if (p.is_rev) p0 += p.ref_len - 1;
            if (mt.is_rev) m0 += mt.ref_len - 1;
            ob.put32((int32_t)(m0 - p0 + (p0 > m0 ? -1 : p0 < m0 ? 1 : 0)));
        }
    }
}

// mem_reg2sam's record selection (conditions use regions only)
DEV bool reg2sam_selects(const MemOpt& opt, const AlnReg* a, int k)
{
    const AlnReg* p = &a[k];
    if (p->score < opt.T) return false;
    if (p->secondary >= 0 && (p->is_alt || !(opt.flag & MEM_F_ALL))) return false;
    if (p->secondary >= 0 && p->secondary < INT_MAX_ && (float)p->score < (float)a[p->secondary].score * opt.drop_ratio) return false;
    return true;
}

// mem_reg2sam: select the records of one read and write them
DEV void reg2sam(const DevIndex& ix, const MemOpt& opt, PostScratch& S, OutBuf& ob, int l_query, const uint8_t* query,
                 int n, AlnReg* a, int32_t* zbuf, int extra_flag, const MateInfo* m, const JobView* jv = 0)
{
    int32_t *cnt = 0, *has_alt = 0;
    if (!(opt.flag & MEM_F_ALL) && n > 0) {
        cnt = zbuf; has_alt = zbuf + n;
        if (xa_prepare(opt, n, a, cnt, has_alt) == 0) cnt = has_alt = 0;
    }
    int n_aa = 0;
    for (int k = 0; k < n; ++k) n_aa += reg2sam_selects(opt, a, k);
    if (n_aa == 0) {
        AlnRec t = reg2aln(ix, opt, S, l_query, query, 0);
        t.flag |= extra_flag;
        aln2out(ix, opt, S, ob, 1, 0, t, m, l_query, query, n, a, 0, 0, -1);
        return;
    }
    int l = 0, mapq0 = 0;
    for (int k = 0; k < n; ++k) {
        if (!reg2sam_selects(opt, a, k)) continue;
        const AlnReg* p = &a[k];
        AlnRec q = reg2aln(ix, opt, S, l_query, query, p, jv);
        q.flag |= extra_flag;
        if (p->secondary >= 0) q.sub = -1;
        if (l && p->secondary < 0) q.flag |= (opt.flag & MEM_F_NO_MULTI) ? 0x10000 : 0x800;
        if (l && !p->is_alt && q.mapq > mapq0) q.mapq = mapq0;
        if (l == 0) mapq0 = q.mapq;
        // the record's own CIGAR/MD must be written before XA reuses the scratch: aln2out does that in order
        aln2out(ix, opt, S, ob, n_aa, l, q, m, l_query, query, n, a, cnt, has_alt, cnt ? k : -1, jv);
        ++l;
    }
}
