// k_extend.hip -- chain -> alignment regions by banded affine-gap extension, one wavefront
// per read.
//
// Replaces, for the reference call at jnibwa.c:214, upstream bwamem.c mem_chain2aln and
// ksw.c ksw_extend2.  The per-read control flow (which seed to
// extend, band retries, local vs. to-end choice) is sequential and wave-uniform; the DP row
// is the parallel dimension: lane l owns query column c + l of the current 64-column chunk.
// Within a row M(i,j) and E(i,j) depend only on row i-1, and F(i,j) is a max-plus prefix of
// M(i,.), so it is computed with a wavefront shuffle scan; H/E rows live in LDS with the
// in-place (eh[]) update order kept, because stale cells outside the live window are
// observable through the window-growth rule.  Integer scoring throughout: no MFMA.
#include "dev_common.h"
#include "wave_ops.h"
#include "pk16.h"
#include "kernels.h"

struct ExtRes { int score, qle, tle, gtle, gscore, max_off; };

// rm: index mask of the rows.  Only the band around the current row is ever live (columns i - w .. i + w + 1), so the rows of
// the general form are rings of a power-of-two size >= 2 w + 4 instead of arrays as long as the read: a 10 kb read needs 6 KB
// of LDS for its rows, not 120 KB, and a CU holds ten such wavefronts instead of one.  (Rows in global memory: rm = ~0.)
struct ExtLds { int32_t* eh_h; int32_t* eh_e; int32_t* tmpM; const uint8_t* query; int rm; };

// Early exit of the extension DP (both forms below).  Upstream's ksw_extend2 keeps computing rows until the target
// runs out, the row maximum is 0 or the z-drop test fires; for a query that has been matched to its end that means
// up to |query| further rows of slowly decaying deletion scores that change nothing.  After a row, every score a later
// row can still produce derives from the stored state {h[j] = H(i,j-1), e[j] = E(i+1,j)}, j >= beg, and from the
// first-column values still to be injected: moving down never gains, moving one column right gains at most mx.  Hence
// B = max_j (max(h[j] + mx, e[j]) + mx * (qlen-1-j)) bounds all later H.  When B <= max (the maximum can no longer be
// beaten: updates need m > max) and B < gscore (the to-end score and its row cannot change: updates need h1 >= gscore),
// the outputs are final and the loop stops -- same result, fewer rows.
static __device__ inline int ext_bound_term(int h, int e, int mx, int cols_right)
{
    int v = h > 0 ? h + mx : 0;                               // M of the next row is 0 for a dead predecessor
    v = v > e ? v : e;
    return v > 0 ? v + mx * cols_right : 0;
}

// ksw_extend2 with the query read as query[q0 + qstep*j] and the target as the doubled-strand
// reference base at t0 + tstep*i.
static __device__ ExtRes extend_wave(const DevIndex& ix, const MemOpt& opt, const ExtLds& L, int lane,
                                     int qlen, int q0, int qstep, int tlen, int64_t t0, int tstep,
                                     int w, int end_bonus, int zdrop, int h0, unsigned long long& n_cells)
{
    const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    int i, beg, end, max, max_i, max_j, max_ins, max_del, max_ie, gscore, max_off;
    if (h0 < 0) h0 = 0;
    const ScoreTab ST = score_tab(opt);
    const int RM = L.rm;
    const int mx = score_max(opt);
    {   // clip the band by the longest affordable gap
        max_ins = div_plus(qlen * mx + end_bonus - o_ins, e_ins, 1);
        max_ins = max_ins > 1 ? max_ins : 1;
        w = w < max_ins ? w : max_ins;
        max_del = div_plus(qlen * mx + end_bonus - o_del, e_del, 1);
        max_del = max_del > 1 ? max_del : 1;
        w = w < max_del ? w : max_del;
    }
    // first row: decay from h0 by insertion costs.  Index j of the row is H(-1, j-1); the ring holds indices 0 .. w + 1 now and
    // every row below adds the index the band is about to reach (no earlier row can have written it)
#define EXT_INIT_H(j) ((j) == 0 ? h0 : h0 > oe_ins && h0 - oe_ins - ((j) - 1) * e_ins > 0 ? h0 - oe_ins - ((j) - 1) * e_ins : 0)
    for (int j = lane; j <= qlen && j <= w + 1; j += WAVE) { L.eh_h[j & RM] = EXT_INIT_H(j); L.eh_e[j & RM] = 0; }
    max = h0; max_i = max_j = -1; max_ie = -1; gscore = -1; max_off = 0;
    beg = 0; end = qlen;
    int tch = 4;
    __syncthreads();
    for (i = 0; i < tlen; ++i) {
        if ((i & 63) == 0) {                               // 64 target bases, one per lane
            int ii = i + lane;
            tch = ii < tlen ? ref_base2(ix, t0 + (int64_t)tstep * ii) : 4;
        }
        if (lane == 0 && i + w + 1 <= qlen) { L.eh_h[(i + w + 1) & RM] = EXT_INIT_H(i + w + 1); L.eh_e[(i + w + 1) & RM] = 0; }
        const int tb = wave_bcast(tch, i & 63);
        const int ms0 = score_at(ST.p[0], ST.n[0], tb), ms1 = score_at(ST.p[1], ST.n[1], tb), ms2 = score_at(ST.p[2], ST.n[2], tb), ms3 = score_at(ST.p[3], ST.n[3], tb), ms4 = score_at(ST.p[4], ST.n[4], tb);
        int m = 0, mj = -1, h1, h1i, hlast = 0;
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        if (beg == 0) { h1i = h0 - (o_del + e_del * (i + 1)); if (h1i < 0) h1i = 0; }
        else h1i = 0;
        h1 = h1i;
        // phase A: M(i,j) from the previous row, before any cell of eh_h is overwritten
        for (int c = beg; c < end; c += WAVE) {
            int j = c + lane;
            if (j < end) {
                int Mp = L.eh_h[j & RM];
                int qc = L.query[q0 + qstep * j];
                int sc = qc == 0 ? ms0 : qc == 1 ? ms1 : qc == 2 ? ms2 : qc == 3 ? ms3 : ms4;
                L.tmpM[j & RM] = Mp ? Mp + sc : 0;
            }
        }
        __syncthreads();
        // phase B: F by prefix scan, H, E, row maximum; writes eh_h[j+1], eh_e[j]
        int fcarry = 0;
        for (int c = beg; c < end; c += WAVE) {
            int j = c + lane;
            bool act = j < end;
            int M = act ? L.tmpM[j & RM] : 0;
            int e = act ? L.eh_e[j & RM] : 0;
            int t = M - oe_ins; t = t > 0 ? t : 0;
            int U = act ? t + j * e_ins : NEG_INF_I32;
            int P = dpp_prefix_max(U, NEG_INF_I32);
            int Pex = dpp_shr1(P, NEG_INF_I32);
            int f = fcarry - (j - c) * e_ins;
            if (lane > 0) { int g = Pex - (j - 1) * e_ins; f = f > g ? f : g; }
            int h = M > e ? M : e;
            h = h > f ? h : f;
            if (!act) h = -1;
            int mc = wave_readlane(dpp_prefix_max(h, NEG_INF_I32), 63);
            unsigned long long bal = wave_ballot(act && h == mc);
            int mjc = c + 63 - __clzll(bal);
            if (mc >= m) { m = mc; mj = mjc; }
            int last = (end - 1 - c) < 63 ? (end - 1 - c) : 63;
            hlast = wave_readlane(h, last);
            int Plast = wave_readlane(P, 63);
            {   // F at column c+64 for the next chunk
                int f1 = fcarry - WAVE * e_ins, f2 = Plast - (c + 63) * e_ins;
                fcarry = f1 > f2 ? f1 : f2;
            }
            if (act) {
                int t2 = M - oe_del; t2 = t2 > 0 ? t2 : 0;
                int en = e - e_del; en = en > t2 ? en : t2;
                L.eh_e[j & RM] = en;
                L.eh_h[(j + 1) & RM] = h;
            }
        }
        if (end > beg) {                                   // eh[beg].h = first-column value, eh[end].h = H(i,end-1)
            n_cells += (unsigned long long)(end - beg);
            h1 = hlast;
            if (lane == 0) L.eh_h[beg & RM] = h1i;
        } else if (lane == 0) L.eh_h[end & RM] = h1i;
        if (lane == 0) L.eh_e[end & RM] = 0;
        {
            int jafter = end > beg ? end : beg;
            if (jafter == qlen) {
                max_ie = gscore > h1 ? max_ie : i;
                gscore = gscore > h1 ? gscore : h1;
            }
        }
        if (m == 0) break;
        if (m > max) {
            max = m; max_i = i; max_j = mj;
            int d = mj - i; d = d < 0 ? -d : d;
            max_off = max_off > d ? max_off : d;
        } else if (zdrop > 0) {
            if (i - max_i > mj - max_j) {
                if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break;
            } else {
                if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break;
            }
        }
        __syncthreads();
        // shrink the window to the non-zero span of the row just written
        {
            int nb = end;
            for (int c = beg; c < end; c += WAVE) {
                int j = c + lane;
                int nz = j < end && (L.eh_h[j & RM] != 0 || L.eh_e[j & RM] != 0);
                unsigned long long bal = wave_ballot(nz);
                if (bal) { nb = c + __ffsll((long long)bal) - 1; break; }
            }
            beg = nb;
            int jl = beg - 1;
            for (int hi = end; hi >= beg; hi -= WAVE) {
                int p = hi - lane;
                int nz = p >= beg && (L.eh_h[p & RM] != 0 || L.eh_e[p & RM] != 0);
                unsigned long long bal = wave_ballot(nz);
                if (bal) { jl = hi - (__ffsll((long long)bal) - 1); break; }
            }
            end = jl + 2 < qlen ? jl + 2 : qlen;
        }
        __syncthreads();
        if (gscore > 0 && m + mx * (qlen - 1 - mj) <= max) {   // see ext_bound(): the remaining rows cannot change the result
            int B = 0;
            const int lim = qlen < i + w + 3 ? qlen : i + w + 3;        // indices the ring holds (the next row's entering index included)
            if (lane == 0 && i + w + 2 <= qlen) { L.eh_h[(i + w + 2) & RM] = EXT_INIT_H(i + w + 2); L.eh_e[(i + w + 2) & RM] = 0; }
            __syncthreads();
            for (int c = beg; c < lim; c += WAVE) {
                const int j = c + lane;
                const int term = j < lim ? ext_bound_term(L.eh_h[j & RM], L.eh_e[j & RM], mx, qlen - 1 - j) : 0;
                const int bc = wave_max(term);
                B = B > bc ? B : bc;
            }
            // indices beyond hold the untouched first row, whose terms fall with j: the first of them is the largest
            if (lim < qlen) { const int t0 = ext_bound_term(EXT_INIT_H(lim), 0, mx, qlen - 1 - lim); B = B > t0 ? B : t0; }
            if (beg == 0) { const int hb = h0 - (o_del + e_del * (i + 2)); if (hb > 0 && hb + mx * qlen > B) B = hb + mx * qlen; }
            if (B <= max && B < gscore) break;
        }
    }
    __syncthreads();
    ExtRes r;
    r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
    return r;
}

// Register-resident form of the same DP for qlen + 1 <= 64 * C (every 150 bp read): lane l owns columns
// t*64 + l, t < C; the H/E rows never leave the VGPRs, cross-lane traffic is DPP (prefix maximum, shift by one
// lane), and the window bookkeeping is done on ballots.  No LDS, no barriers.  Same row-by-row decisions as
// extend_wave() (the general, LDS-backed form used for longer queries), hence the same results.
template <int C>
static __device__ ExtRes extend_wave_reg(const DevIndex& ix, const MemOpt& opt, const uint8_t* query, int lane,
                                         int qlen, int q0, int qstep, int tlen, int64_t t0, int tstep,
                                         int w, int end_bonus, int zdrop, int h0, unsigned long long& n_cells)
{
    const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    int i, beg, end, max, max_i, max_j, max_ins, max_del, max_ie, gscore, max_off;
    int ehh[C], ehe[C], je[C], jm1e[C];
    uint32_t scp[C];
    const ScoreTab ST = score_tab(opt);
    if (h0 < 0) h0 = 0;
#pragma unroll
    for (int t = 0; t < C; ++t) {                          // first row: decay from h0 by insertion costs
        const int j = t * WAVE + lane;
        int v = 0;
        if (j == 0) v = h0;
        else if (j <= qlen && h0 > oe_ins) {
            int vj = h0 - oe_ins - (j - 1) * e_ins;
            if (j == 1 || vj + e_ins > e_ins) v = vj;
        }
        ehh[t] = v; ehe[t] = 0;
        { int sn; score_lane(ST, j < qlen ? query[q0 + qstep * j] : 4, scp[t], sn); }
        je[t] = j * e_ins; jm1e[t] = (j - 1) * e_ins;
    }
    const int mx = score_max(opt);
    {
        max_ins = div_plus(qlen * mx + end_bonus - o_ins, e_ins, 1);
        max_ins = max_ins > 1 ? max_ins : 1;
        w = w < max_ins ? w : max_ins;
        max_del = div_plus(qlen * mx + end_bonus - o_del, e_del, 1);
        max_del = max_del > 1 ? max_del : 1;
        w = w < max_del ? w : max_del;
    }
    max = h0; max_i = max_j = -1; max_ie = -1; gscore = -1; max_off = 0;
    beg = 0; end = qlen;
    int tch = 4;
    for (i = 0; i < tlen; ++i) {
        if ((i & 63) == 0) { int ii = i + lane; tch = ii < tlen ? ref_base2(ix, t0 + (int64_t)tstep * ii) : 4; }
        const int tb = wave_readlane(tch, i & 63);
        int m, mj, h1, h1i;
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        if (beg == 0) { h1i = h0 - (o_del + e_del * (i + 1)); if (h1i < 0) h1i = 0; }
        else h1i = 0;
        h1 = h1i;
        const int pos1 = end > beg ? beg : end;            // the column whose eh.h becomes the first-column value
        int fcarry = NEG_INF_I32, prev_last = -1, best = -1;
#pragma unroll
        for (int t = 0; t < C; ++t) {
            const int c0 = t * WAVE, j = c0 + lane;
            const bool act = j >= beg && j < end;
            const int Mp = ehh[t], e = ehe[t];
            const int sc = (int)(int8_t)(scp[t] >> (tb << 3));   // tb is a reference base (0..3): rows past the target are never computed
            const int M = act && Mp ? Mp + sc : 0;
            int tt = M - oe_ins; tt = tt > 0 ? tt : 0;
            const int U = act ? tt + je[t] : NEG_INF_I32;
            const int P = dpp_prefix_max(U, NEG_INF_I32);
            const int Pex = dpp_shr1(P, NEG_INF_I32);
            // F(i,j) = max over earlier columns k of (tt_k - (j-1-k) e_ins): never negative; fcarry brings in the earlier chunks
            int f = Pex - jm1e[t];
            if (t > 0) { const int fc = fcarry - (j - c0) * e_ins; f = f > fc ? f : fc; }
            if (j == beg) f = 0;
            int h = M > e ? M : e;
            h = h > f ? h : f;
            if (!act) h = -1;
            // row maximum with the last column that attains it: one scan over (h, column) keys; later chunks win ties
            const int mk = wave_readlane(dpp_prefix_max((int)((uint32_t)h << 8) | j, -1), 63);    // h = -1 outside the live columns: those keys are negative
            best = mk >= best ? mk : best;
            if (t + 1 < C) {
                const int Plast = wave_readlane(P, 63);
                const int f1 = fcarry - WAVE * e_ins, f2 = Plast - (c0 + 63) * e_ins;
                fcarry = f1 > f2 ? f1 : f2;
            }
            int t2 = M - oe_del; t2 = t2 > 0 ? t2 : 0;
            int en = e - e_del; en = en > t2 ? en : t2;
            if (end > beg && ((end - 1) >> 6) == t) h1 = wave_readlane(h, (end - 1) & 63);
            // in-place row update: eh[pos1].h = h1i, eh[j+1].h = H(i,j) for the live columns, eh[j].e = E(i+1,j), eh[end].e = 0.
            // h is -1 outside the live columns, so "the column to the left was live" reads off the shifted value
            const int hsh = dpp_shr1(h, prev_last);
            if (t + 1 < C) prev_last = wave_readlane(h, 63);
            ehh[t] = j == pos1 ? h1i : hsh >= 0 ? hsh : ehh[t];
            ehe[t] = act ? en : j == end ? 0 : ehe[t];
        }
        m = best < 0 ? 0 : best >> 8;
        mj = best < 0 ? -1 : best & 255;
        if (end > beg) n_cells += (unsigned long long)(end - beg);
        {
            const int jafter = end > beg ? end : beg;
            if (jafter == qlen) {
                max_ie = gscore > h1 ? max_ie : i;
                gscore = gscore > h1 ? gscore : h1;
            }
        }
        if (m == 0) break;
        if (m > max) {
            max = m; max_i = i; max_j = mj;
            int d = mj - i; d = d < 0 ? -d : d;
            max_off = max_off > d ? max_off : d;
        } else if (zdrop > 0) {
            if (i - max_i > mj - max_j) {
                if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break;
            } else {
                if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break;
            }
        }
        {   // shrink the window to the non-zero span of the row just written
            int nb = end, jl = -2;
#pragma unroll
            for (int t = 0; t < C; ++t) {
                const int j = t * WAVE + lane;
                const unsigned long long nz = wave_ballot((ehh[t] != 0 || ehe[t] != 0) && j >= beg && j < end);
                if (nz && nb == end) nb = t * WAVE + __ffsll((long long)nz) - 1;
            }
            beg = nb;
#pragma unroll
            for (int t = C - 1; t >= 0; --t) {
                const int j = t * WAVE + lane;
                const unsigned long long nz = wave_ballot((ehh[t] != 0 || ehe[t] != 0) && j >= beg && j <= end);
                if (nz && jl == -2) jl = t * WAVE + 63 - __clzll((long long)nz);
            }
            if (jl == -2) jl = beg - 1;
            end = jl + 2 < qlen ? jl + 2 : qlen;
        }
        if (gscore > 0 && m + mx * (qlen - 1 - mj) <= max) {   // see ext_bound(): the remaining rows cannot change the result
            int B = 0;
#pragma unroll
            for (int t = 0; t < C; ++t) {
                const int j = t * WAVE + lane;
                const int term = j >= beg && j < qlen ? ext_bound_term(ehh[t], ehe[t], mx, qlen - 1 - j) : 0;
                const int bc = wave_readlane(dpp_prefix_max(term, 0), 63);
                B = B > bc ? B : bc;
            }
            if (beg == 0) { const int hb = h0 - (o_del + e_del * (i + 2)); if (hb > 0 && hb + mx * qlen > B) B = hb + mx * qlen; }
            if (B <= max && B < gscore) break;
        }
    }
    ExtRes r;
    r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
    return r;
}

// The register form for queries of 64 .. 126 bases -- two thirds of the DP rows of 150 bp reads -- with its two 64-column
// chunks packed into the halves of one register (pk16.h): lane l holds column l in the low half and column 64 + l in the high
// half of every row register, and one stream of packed 16-bit instructions computes both.  What cannot be packed stays
// 32-bit: the row maximum with its column (one scan over the larger of the two chunks' keys) and the cross-lane moves (DPP
// applies to 32-bit moves; the move carries both halves).  Two details differ from the 32-bit form, neither observable:
// the prefix maximum behind F uses 0 instead of -infinity as its identity (U >= 0 always; a phantom 0 only adds F
// candidates <= 0, and F <= 0 never decides anything because E >= 0 is in the same maximum), and dead columns take part in
// it with tt = 0 (same argument).  Usable when every score fits 16 bits: extend_pk2_ok().
static __device__ inline bool extend_pk2_ok(const MemOpt& opt, int qlen, int h0, int mx)
{
    return qlen + 1 > WAVE && qlen + 1 <= 2 * WAVE && opt.e_ins >= 0 && opt.e_del >= 0 && opt.o_ins >= 0 && opt.o_del >= 0 && mx > 0
        && (long long)(h0 > 0 ? h0 : 0) + (long long)qlen * mx + 130ll * opt.e_ins < 30000 && opt.o_ins + opt.e_ins < 30000 && opt.o_del + opt.e_del < 30000;
}
#define PK_DPP_ZERO(v, ctrl, rowmask) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rowmask), 0xf, (rowmask) == 0xf))
static __device__ ExtRes extend_wave_pk2(const DevIndex& ix, const MemOpt& opt, const uint8_t* query, int lane,
                                         int qlen, int q0, int qstep, int tlen, int64_t t0, int tstep,
                                         int w, int end_bonus, int zdrop, int h0, unsigned long long& n_cells)
{
    const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    int i, beg, end, max, max_i, max_j, max_ins, max_del, max_ie, gscore, max_off;
    const ScoreTab ST = score_tab(opt);
    if (h0 < 0) h0 = 0;
    uint32_t ehh, ehe = 0, scp_lo, scp_hi;
    {   // first row: decay from h0 by insertion costs
        int v[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int j = t * WAVE + lane;
            v[t] = 0;
            if (j == 0) v[t] = h0;
            else if (j <= qlen && h0 > oe_ins) { const int vj = h0 - oe_ins - (j - 1) * e_ins; if (j == 1 || vj > 0) v[t] = vj; }
        }
        ehh = pk_pair(v[0], v[1]);
        int sn;
        score_lane(ST, lane < qlen ? query[q0 + qstep * lane] : 4, scp_lo, sn);
        score_lane(ST, lane + WAVE < qlen ? query[q0 + qstep * (lane + WAVE)] : 4, scp_hi, sn);
    }
    const uint32_t je = pk_pair(lane * e_ins, (lane + WAVE) * e_ins);
    const uint32_t jm1e = pk_pair(lane == 0 ? 0 : (lane - 1) * e_ins, (lane + WAVE - 1) * e_ins);   // (column 0 has no F: with 0 here it gets F = 0)
    const uint32_t OEIB = pk_both(oe_ins + 128), OEDB = pk_both(oe_del + 128), ED = pk_both(e_del), ONE = pk_both(1), BIAS = pk_both(128);
    scp_lo ^= 0x80808080u; scp_hi ^= 0x80808080u;            // score + 128 as an unsigned byte
    const int jhi = lane + WAVE;
    const int mx = score_max(opt);
    {
        max_ins = div_plus(qlen * mx + end_bonus - o_ins, e_ins, 1);
        max_ins = max_ins > 1 ? max_ins : 1;
        w = w < max_ins ? w : max_ins;
        max_del = div_plus(qlen * mx + end_bonus - o_del, e_del, 1);
        max_del = max_del > 1 ? max_del : 1;
        w = w < max_del ? w : max_del;
    }
    max = h0; max_i = max_j = -1; max_ie = -1; gscore = -1; max_off = 0;
    beg = 0; end = qlen;
    int tch = 4;
    for (i = 0; i < tlen; ++i) {
        if ((i & 63) == 0) { int ii = i + lane; tch = ii < tlen ? ref_base2(ix, t0 + (int64_t)tstep * ii) : 4; }
        const int tb = wave_readlane(tch, i & 63);
        int m, mj, h1, h1i;
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        if (beg == 0) { h1i = h0 - (o_del + e_del * (i + 1)); if (h1i < 0) h1i = 0; }
        else h1i = 0;
        h1 = h1i;
        const int pos1 = end > beg ? beg : end;            // the column whose eh.h becomes the first-column value
        const bool a_lo = lane >= beg && lane < end, a_hi = jhi >= beg && jhi < end;
        const uint32_t am = (a_lo ? 0xffffu : 0u) | (a_hi ? 0xffff0000u : 0u);
        // M = live && Mp ? Mp + score : 0.  The scores sit in the lane as bytes biased by 128 (one byte permute fetches both columns'
        // scores against this row's base); "Mp == 0" is the sign of Mp - 1 (Mp >= 0).  Mb = M + 128 where the cell is alive, 0 where not:
        // the bias goes into the constants below, and a dead cell's -128 loses every maximum it enters just as its 0 would (E >= 0).
        const uint32_t Mp = ehh, e = ehe;
        const uint32_t Mb = pk_add(Mp, pk_bytes(scp_lo, scp_hi, tb)) & am & ~pk_sra15(pk_sub(Mp, ONE));
        const uint32_t M = pk_sub(Mb, BIAS);
        const uint32_t tt = pk_max(pk_sub(Mb, OEIB), 0u);
        uint32_t P = pk_add(tt, je);                                           // U, then its inclusive prefix maximum per half
        P = pk_max(P, PK_DPP_ZERO(P, DPP_ROW_SHR(1), 0xf));
        P = pk_max(P, PK_DPP_ZERO(P, DPP_ROW_SHR(2), 0xf));
        P = pk_max(P, PK_DPP_ZERO(P, DPP_ROW_SHR(4), 0xf));
        P = pk_max(P, PK_DPP_ZERO(P, DPP_ROW_SHR(8), 0xf));
        P = pk_max(P, PK_DPP_ZERO(P, DPP_ROW_BCAST15, 0xa));
        P = pk_max(P, PK_DPP_ZERO(P, DPP_ROW_BCAST31, 0xc));
        uint32_t Pex = PK_DPP_ZERO(P, DPP_WAVE_SHR1, 0xf);
        Pex = pk_max(Pex, (uint32_t)wave_readlane((int)P, 63) << 16);          // the second chunk continues the first one's scan
        const uint32_t f = pk_sub(Pex, jm1e);
        uint32_t h = pk_max(pk_max(M, e), f);
        h = (h & am) | ~am;                                                    // -1 outside the live columns
        // row maximum with the last column that attains it: one scan over (h, column) keys
        const int klo = (int)((uint32_t)(int)(int16_t)(h & 0xffffu) << 8) | lane, khi = (int)((uint32_t)((int)h >> 16) << 8) | jhi;
        const int best = wave_readlane(dpp_prefix_max(klo > khi ? klo : khi, -1), 63);
        const uint32_t t2 = pk_max(pk_sub(Mb, OEDB), 0u);
        const uint32_t en = pk_max(pk_sub(e, ED), t2);
        if (end > beg) { const int v = wave_readlane((int)h, (end - 1) & 63); h1 = ((end - 1) >> 6) ? v >> 16 : (int)(int16_t)(v & 0xffff); }
        // in-place row update: eh[pos1].h = h1i, eh[j+1].h = H(i,j) for the live columns, eh[j].e = E(i+1,j), eh[end].e = 0
        const uint32_t fill = (uint32_t)wave_readlane((int)h, 63) << 16 | 0xffffu;       // lane 0: nothing left of column 0, column 63 left of column 64
        const uint32_t hsh = (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)h, DPP_WAVE_SHR1, 0xf, 0xf, false);
        const uint32_t dead = pk_sra15(hsh);                                   // halves whose left neighbour was not live keep their value
        ehh = (ehh & dead) | (hsh & ~dead);
        {
            const uint32_t half = (pos1 >> 6) ? 0xffff0000u : 0xffffu;
            ehh = lane == (pos1 & 63) ? (ehh & ~half) | (pk_both(h1i) & half) : ehh;
        }
        ehe = (en & am) | (ehe & ~am);
        {
            const uint32_t half = (end >> 6) ? 0xffff0000u : 0xffffu;
            ehe = lane == (end & 63) ? ehe & ~half : ehe;
        }
        m = best < 0 ? 0 : best >> 8;
        mj = best < 0 ? -1 : best & 255;
        if (end > beg) n_cells += (unsigned long long)(end - beg);
        {
            const int jafter = end > beg ? end : beg;
            if (jafter == qlen) {
                max_ie = gscore > h1 ? max_ie : i;
                gscore = gscore > h1 ? gscore : h1;
            }
        }
        if (m == 0) break;
        if (m > max) {
            max = m; max_i = i; max_j = mj;
            int d = mj - i; d = d < 0 ? -d : d;
            max_off = max_off > d ? max_off : d;
        } else if (zdrop > 0) {
            if (i - max_i > mj - max_j) {
                if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break;
            } else {
                if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break;
            }
        }
        {   // shrink the window to the non-zero span of the row just written
            const uint32_t nz = ehh | ehe;
            const unsigned long long nzl = wave_ballot((nz & 0xffffu) != 0), nzh = wave_ballot((nz >> 16) != 0);
            const unsigned long long fl = nzl & wave_ballot(a_lo), fh = nzh & wave_ballot(a_hi);
            beg = fl ? __ffsll((long long)fl) - 1 : fh ? WAVE + __ffsll((long long)fh) - 1 : end;
            const unsigned long long gl = nzl & wave_ballot(lane >= beg && lane <= end), gh = nzh & wave_ballot(jhi >= beg && jhi <= end);
            const int jl = gh ? WAVE + 63 - __clzll((long long)gh) : gl ? 63 - __clzll((long long)gl) : beg - 1;
            end = jl + 2 < qlen ? jl + 2 : qlen;
        }
        if (gscore > 0 && m + mx * (qlen - 1 - mj) <= max) {   // see ext_bound(): the remaining rows cannot change the result
            int B = 0;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int j = t * WAVE + lane;
                const int hh = t ? (int)ehh >> 16 : (int)(int16_t)(ehh & 0xffffu), ee = t ? (int)ehe >> 16 : (int)(int16_t)(ehe & 0xffffu);
                const int term = j >= beg && j < qlen ? ext_bound_term(hh, ee, mx, qlen - 1 - j) : 0;
                const int bc = wave_readlane(dpp_prefix_max(term, 0), 63);
                B = B > bc ? B : bc;
            }
            if (beg == 0) { const int hb = h0 - (o_del + e_del * (i + 2)); if (hb > 0 && hb + mx * qlen > B) B = hb + mx * qlen; }
            if (B <= max && B < gscore) break;
        }
    }
    ExtRes r;
    r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
    return r;
}

// The packed form for queries of any length: the same halves-of-a-register arithmetic as extend_wave_pk2, with the H and E rows
// in LDS instead of registers.  Columns are grouped in aligned pairs of 64-column chunks (pair p = columns 128 p .. 128 p + 127; lane
// l owns 128 p + l in the low and 128 p + 64 + l in the high half of one packed word), and only the pairs the band touches are
// live: the rows are rings of NP pairs (the LDS the general form's three int rows use holds them with room to spare).  Because
// a lane keeps H(i, j - 1) for its *own* column j -- the shift by one column is a DPP move with carries between halves and
// pairs, as in the register forms -- a row is one pass over the pairs: no intermediate M row, one barrier less than the general
// form, half its arithmetic.  F's prefix maximum runs on column offsets relative to the row's first pair, so that its operands
// stay small whatever the query length.  Usable when the scores fit 16 bits and the ring holds the band: extend_pkl_ok().
static __device__ inline bool extend_pkl_ok(const MemOpt& opt, int ring_ints, int qlen, int w, int h0, int mx)
{
    if (ring_ints < 64 || ring_ints > (1 << 20)) return false;
    const int np = ring_ints >> 6;                                  // pairs the ring holds (a power of two)
    return opt.e_ins >= 0 && opt.e_del >= 0 && opt.o_ins >= 0 && opt.o_del >= 0 && mx > 0 && qlen < 65536
        && (long long)(h0 > 0 ? h0 : 0) + (long long)qlen * mx + (long long)(np * 128 + 130) * opt.e_ins < 30000 && opt.o_ins + opt.e_ins < 30000 && opt.o_del + opt.e_del < 30000
        && 2ll * w + 4 <= (long long)np * 128;        // the live columns (i - w .. i + w + 2) never meet a column 128 np further on in the ring
}
static __device__ ExtRes extend_wave_pkl(const DevIndex& ix, const MemOpt& opt, const ExtLds& L, int lane,
                                         int qlen, int q0, int qstep, int tlen, int64_t t0, int tstep,
                                         int w, int end_bonus, int zdrop, int h0, unsigned long long& n_cells)
{
    const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    int i, beg, end, max, max_i, max_j, max_ins, max_del, max_ie, gscore, max_off;
    if (h0 < 0) h0 = 0;
    uint32_t* const EH = (uint32_t*)L.eh_h;                        // [NP][64] packed H rows: H(i-1, j-1) for the lane's two columns
    uint32_t* const EE = (uint32_t*)L.eh_e;                        // [NP][64] packed E rows
    const int NPM = ((L.rm + 1) >> 6) - 1;                         // ring mask over pairs
    const ScoreTab ST = score_tab(opt);
    const int mx = score_max(opt);
    {
        max_ins = div_plus(qlen * mx + end_bonus - o_ins, e_ins, 1);
        max_ins = max_ins > 1 ? max_ins : 1;
        w = w < max_ins ? w : max_ins;
        max_del = div_plus(qlen * mx + end_bonus - o_del, e_del, 1);
        max_del = max_del > 1 ? max_del : 1;
        w = w < max_del ? w : max_del;
    }
#define PKL_INIT_H(j) ((j) == 0 ? h0 : h0 > oe_ins && h0 - oe_ins - ((j) - 1) * e_ins > 0 ? h0 - oe_ins - ((j) - 1) * e_ins : 0)
#define PKL_SLOT(j) (((((j) >> 7) & NPM) << 6) + ((j) & 63))
    // one column's H (and E = 0) into its half of its slot; called by all lanes, done by the owning one
#define PKL_SET_COL(j, hv, with_e) do { if (lane == ((j) & 63)) { const uint32_t hm_ = ((j) >> 6 & 1) ? 0xffff0000u : 0xffffu; const int s_ = PKL_SLOT(j); \
        EH[s_] = (EH[s_] & ~hm_) | (pk_both(hv) & hm_); if (with_e) EE[s_] = EE[s_] & ~hm_; } } while (0)
    // first row: columns 0 .. min(qlen, w + 1); every row below adds the column the band is about to reach
    for (int pp = 0; pp <= ((qlen < w + 1 ? qlen : w + 1) >> 7); ++pp) {
        const int jl = (pp << 7) + lane, jh = jl + WAVE;
        EH[((pp & NPM) << 6) + lane] = pk_pair(jl <= qlen && jl <= w + 1 ? PKL_INIT_H(jl) : 0, jh <= qlen && jh <= w + 1 ? PKL_INIT_H(jh) : 0);
        EE[((pp & NPM) << 6) + lane] = 0;
    }
    const uint32_t OEIB = pk_both(oe_ins + 128), OEDB = pk_both(oe_del + 128), ED = pk_both(e_del), ONE = pk_both(1), BIAS = pk_both(128);
    const uint32_t JE0 = pk_pair(lane * e_ins, (lane + WAVE) * e_ins), JM0 = pk_pair((lane - 1) * e_ins, (lane + WAVE - 1) * e_ins);
    max = h0; max_i = max_j = -1; max_ie = -1; gscore = -1; max_off = 0;
    beg = 0; end = qlen;
    int tch = 4;
    __syncthreads();
    for (i = 0; i < tlen; ++i) {
        if ((i & 63) == 0) { int ii = i + lane; tch = ii < tlen ? ref_base2(ix, t0 + (int64_t)tstep * ii) : 4; }
        const int tb = wave_readlane(tch, i & 63);
        // the column the band reaches in this row enters the ring (its slot may still hold a column 128 NP to the left)
        if (i + w + 1 <= qlen) { const int jn = i + w + 1; PKL_SET_COL(jn, PKL_INIT_H(jn), true); }
        int m, mj, h1, h1i;
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        if (beg == 0) { h1i = h0 - (o_del + e_del * (i + 1)); if (h1i < 0) h1i = 0; }
        else h1i = 0;
        h1 = h1i;
        // score of a query base against this row's target base, biased by 128: bytes 0..3 = bases A..T, an ambiguous base apart
        const uint32_t msw = (uint32_t)(uint8_t)(score_at(ST.p[0], ST.n[0], tb) + 128) | (uint32_t)(uint8_t)(score_at(ST.p[1], ST.n[1], tb) + 128) << 8
                           | (uint32_t)(uint8_t)(score_at(ST.p[2], ST.n[2], tb) + 128) << 16 | (uint32_t)(uint8_t)(score_at(ST.p[3], ST.n[3], tb) + 128) << 24;
        const uint32_t msn = (uint32_t)(uint8_t)(score_at(ST.p[4], ST.n[4], tb) + 128);
        // (every word of the rows is read and written by one lane only -- the owner of its two columns -- so the row needs no barrier)
        int kmax = -1;
        if (end > beg) {
            const int pb = beg >> 7, pe = (end - 1) >> 7;
            uint32_t carryU = 0, fillh = 0xffffu;                // F's running maximum and H of the column left of the pair, from the pairs before
            for (int pr = pb; pr <= pe; ++pr) {
                const int jlo = (pr << 7) + lane, jhi = jlo + WAVE, slot = ((pr & NPM) << 6) + lane;
                const bool a_lo = jlo >= beg && jlo < end, a_hi = jhi >= beg && jhi < end;
                const uint32_t am = (a_lo ? 0xffffu : 0u) | (a_hi ? 0xffff0000u : 0u);
                const int qlo = jlo < qlen ? L.query[q0 + qstep * jlo] : 4, qhi = jhi < qlen ? L.query[q0 + qstep * jhi] : 4;
                const uint32_t sc = (qlo < 4 ? msw >> (qlo << 3) & 0xffu : msn) | (qhi < 4 ? msw >> (qhi << 3) & 0xffu : msn) << 16;
                const uint32_t Mp = EH[slot], e = EE[slot];
                const uint32_t Mb = pk_add(Mp, sc) & am & ~pk_sra15(pk_sub(Mp, ONE));
                const uint32_t M = pk_sub(Mb, BIAS);
                const uint32_t tt = pk_max(pk_sub(Mb, OEIB), 0u);
                const uint32_t joff = pk_both(((pr - pb) << 7) * e_ins);
                uint32_t P = pk_add(tt, pk_add(JE0, joff));
                P = pk_max(P, PK_DPP_ZERO(P, DPP_ROW_SHR(1), 0xf));
                P = pk_max(P, PK_DPP_ZERO(P, DPP_ROW_SHR(2), 0xf));
                P = pk_max(P, PK_DPP_ZERO(P, DPP_ROW_SHR(4), 0xf));
                P = pk_max(P, PK_DPP_ZERO(P, DPP_ROW_SHR(8), 0xf));
                P = pk_max(P, PK_DPP_ZERO(P, DPP_ROW_BCAST15, 0xa));
                P = pk_max(P, PK_DPP_ZERO(P, DPP_ROW_BCAST31, 0xc));
                uint32_t Pex = PK_DPP_ZERO(P, DPP_WAVE_SHR1, 0xf);
                const uint32_t plast = (uint32_t)wave_readlane((int)P, 63);
                Pex = pk_max(Pex, pk_max(carryU, plast << 16));       // earlier pairs feed both halves, the first chunk feeds the second
                carryU = pk_max(carryU, pk_both((int)(plast >> 16)));
                carryU = pk_max(carryU, pk_both((int)(plast & 0xffffu)));
                // F = (maximum of U before the column) - (offset of the column before it); nothing precedes the row's very first offset
                uint32_t jm1 = pk_add(JM0, joff);
                if (pr == pb && lane == 0) jm1 &= 0xffff0000u;
                const uint32_t f = pk_sub(Pex, jm1);
                uint32_t h = pk_max(pk_max(M, e), f);
                h = (h & am) | ~am;                                    // -1 outside the live columns
                const int klo = (int)((uint32_t)(int)(int16_t)(h & 0xffffu) << 16) | jlo, khi = (int)(h & 0xffff0000u) | jhi;
                const int kk = klo > khi ? klo : khi;
                kmax = kmax > kk ? kmax : kk;
                const uint32_t t2 = pk_max(pk_sub(Mb, OEDB), 0u);
                const uint32_t en = pk_max(pk_sub(e, ED), t2);
                if (pr == pe) { const int v = wave_readlane((int)h, (end - 1) & 63); h1 = ((end - 1) >> 6 & 1) ? v >> 16 : (int)(int16_t)(v & 0xffff); }
                const uint32_t h63 = (uint32_t)wave_readlane((int)h, 63);
                const uint32_t fill = h63 << 16 | fillh;              // lane 0: column 63 of this pair left of column 64, the pair before left of column 0
                const uint32_t hsh = (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)h, DPP_WAVE_SHR1, 0xf, 0xf, false);
                fillh = h63 >> 16;
                const uint32_t dead = pk_sra15(hsh);                   // halves whose left neighbour was not live keep their value
                EH[slot] = (Mp & dead) | (hsh & ~dead);
                EE[slot] = (en & am) | (e & ~am);
            }
            n_cells += (unsigned long long)(end - beg);
        }
        // the row's single cells: eh[beg].h = first-column value, eh[end].h = H(i, end - 1) (first-column value when the window is empty), eh[end].e = 0
        if (end > beg) { PKL_SET_COL(beg, h1i, false); }
        { const int hv = end > beg ? h1 : h1i; PKL_SET_COL(end, hv, true); }
        const int best = wave_readlane(dpp_prefix_max(kmax, -1), 63);
        m = best < 0 ? 0 : best >> 16;
        mj = best < 0 ? -1 : best & 0xffff;
        {
            const int jafter = end > beg ? end : beg;
            if (jafter == qlen) {
                max_ie = gscore > h1 ? max_ie : i;
                gscore = gscore > h1 ? gscore : h1;
            }
        }
        if (m == 0) break;
        if (m > max) {
            max = m; max_i = i; max_j = mj;
            int d = mj - i; d = d < 0 ? -d : d;
            max_off = max_off > d ? max_off : d;
        } else if (zdrop > 0) {
            if (i - max_i > mj - max_j) {
                if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break;
            } else {
                if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break;
            }
        }
        {   // shrink the window to the non-zero span of the row just written
            int nb = end, jl = -2;
            for (int pr = beg >> 7; pr <= (end - 1) >> 7 && nb == end && end > beg; ++pr) {
                const int jlo = (pr << 7) + lane, jhi = jlo + WAVE, slot = ((pr & NPM) << 6) + lane;
                const uint32_t nz = EH[slot] | EE[slot];
                const unsigned long long fl = wave_ballot((nz & 0xffffu) != 0 && jlo >= beg && jlo < end), fh = wave_ballot((nz >> 16) != 0 && jhi >= beg && jhi < end);
                if (fl) nb = (pr << 7) + __ffsll((long long)fl) - 1;
                else if (fh) nb = (pr << 7) + WAVE + __ffsll((long long)fh) - 1;
            }
            beg = nb;
            for (int pr = end >> 7; pr >= beg >> 7 && jl == -2; --pr) {
                const int jlo = (pr << 7) + lane, jhi = jlo + WAVE, slot = ((pr & NPM) << 6) + lane;
                const uint32_t nz = EH[slot] | EE[slot];
                const unsigned long long gl = wave_ballot((nz & 0xffffu) != 0 && jlo >= beg && jlo <= end), gh = wave_ballot((nz >> 16) != 0 && jhi >= beg && jhi <= end);
                if (gh) jl = (pr << 7) + WAVE + 63 - __clzll((long long)gh);
                else if (gl) jl = (pr << 7) + 63 - __clzll((long long)gl);
            }
            if (jl == -2) jl = beg - 1;
            end = jl + 2 < qlen ? jl + 2 : qlen;
        }
        if (gscore > 0 && m + mx * (qlen - 1 - mj) <= max) {   // see ext_bound(): the remaining rows cannot change the result
            int B = 0;
            const int lim = qlen < i + w + 3 ? qlen : i + w + 3;        // columns the ring holds (the next row's entering column included)
            if (i + w + 2 <= qlen) { const int jn = i + w + 2; PKL_SET_COL(jn, PKL_INIT_H(jn), true); }
            for (int pr = beg >> 7; pr <= (lim - 1) >> 7 && lim > beg; ++pr) {
                const int slot = ((pr & NPM) << 6) + lane;
                const uint32_t hw = EH[slot], ew = EE[slot];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int j = (pr << 7) + t * WAVE + lane;
                    const int hh = t ? (int)hw >> 16 : (int)(int16_t)(hw & 0xffffu), ee = t ? (int)ew >> 16 : (int)(int16_t)(ew & 0xffffu);
                    const int term = j >= beg && j < lim ? ext_bound_term(hh, ee, mx, qlen - 1 - j) : 0;
                    const int bc = wave_readlane(dpp_prefix_max(term, 0), 63);
                    B = B > bc ? B : bc;
                }
            }
            // columns beyond hold the untouched first row, whose terms fall with j: the first of them is the largest
            if (lim < qlen) { const int t0_ = ext_bound_term(PKL_INIT_H(lim), 0, mx, qlen - 1 - lim); B = B > t0_ ? B : t0_; }
            if (beg == 0) { const int hb = h0 - (o_del + e_del * (i + 2)); if (hb > 0 && hb + mx * qlen > B) B = hb + mx * qlen; }
            if (B <= max && B < gscore) break;
        }
    }
#undef PKL_INIT_H
#undef PKL_SLOT
#undef PKL_SET_COL
    __syncthreads();
    ExtRes r;
    r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
    return r;
}

// The diagonal certificate: many extensions of well-placed reads run along the seed's diagonal with at most one mismatch,
// and then the banded DP is decided before it starts.  Let s_j = mat[t_j][q_j] be the scores on the diagonal, a = max(mat),
// D = sum_j (a - s_j) over the whole query (the "deficit") and g = min(o_del + e_del, o_ins + e_ins).  A path from the origin
// to any cell of row i that opens a gap scores at most h0 + a (i+1) - g (n diagonal steps score <= a each, n <= i+1), while
// the ungapped diagonal prefix scores P_i = h0 + a (i+1) - D(0..i) >= h0 + a (i+1) - D.  So if D < g (and D < h0, which keeps
// every P_i alive), every gapped cell of row i < qlen is strictly below P_i and every cell of a row >= qlen strictly below
// P_{qlen-1}: H(i,i) = P_i (its E and F are gapped), the row maximum is P_i at column i only, max_off stays 0, the last
// column's best is the diagonal's (gscore = P_{qlen-1} at row qlen - 1, which needs tlen >= qlen), later rows change nothing,
// and ksw_extend2's outputs are those of a prefix-sum scan: max = the first maximum of (h0, P_0, P_1, ...).  The band never
// matters (the diagonal is inside any band) and the window always contains column i (its predecessor is alive).  The z-drop
// rule can only fire on the diagonal itself (i - max_i == mj - max_j): if it would, the certificate is refused and the
// DP runs.  Independent of w, so the caller's band-doubling loop sees the same score twice and stops, as it would.
static __device__ inline int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
static __device__ inline int wave_prefix_sum(int v, int lane)
{
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) { const int u = __shfl_up(v, o); if (lane >= o) v += u; }
    return v;
}
static __device__ bool extend_diag(const DevIndex& ix, const MemOpt& opt, const uint8_t* query, int lane,
                                   int qlen, int q0, int qstep, int tlen, int64_t t0, int tstep, int zdrop, int h0, ExtRes& r)
{
    if (tlen < qlen || qlen <= 0 || h0 <= 0) return false;
    const int mx = score_max(opt);
    const int oe_del = opt.o_del + opt.e_del, oe_ins = opt.o_ins + opt.e_ins;
    const int g1 = oe_del < oe_ins ? oe_del : oe_ins;
    const ScoreTab ST = score_tab(opt);
    int run = h0, max = h0, max_i = -1, deficit = 0;
    for (int c = 0; c < qlen; c += WAVE) {
        const int j = c + lane;
        const bool act = j < qlen;
        int s = 0;
        if (act) {
            uint32_t p; int n;
            score_lane(ST, query[q0 + qstep * j], p, n);
            s = score_at(p, n, ref_base2(ix, t0 + (int64_t)tstep * j));
        }
        deficit += wave_sum(act ? mx - s : 0);
        if (deficit >= g1 || deficit >= h0) return false;
        const int P = run + wave_prefix_sum(s, lane);                 // P_j (lanes past the query repeat the last one)
        const int pm = wave_prefix_max(act ? P : NEG_INF_I32, lane);
        int before = __shfl_up(pm, 1);                                // the maximum before row j: h0, earlier chunks, earlier lanes
        if (lane == 0 || before < max) before = max;
        if (zdrop > 0 && wave_any(act && P <= before && before - P > zdrop)) return false;
        const int cm = wave_max(act ? P : NEG_INF_I32);
        if (cm > max) { max = cm; max_i = c + __ffsll((long long)wave_ballot(act && P == cm)) - 1; }
        run = wave_bcast(P, WAVE - 1);
    }
    r.score = max; r.qle = max_i + 1; r.tle = max_i + 1; r.gtle = qlen; r.gscore = run; r.max_off = 0;
    return true;
}

// picks the register-resident form when the query fits
// SHORT: the caller knows that every query fits the register forms (tiles of reads of at most 3 * 64 - 1 bases): the forms with
// rows in LDS are then not compiled into the kernel at all -- it is the kernel's hungriest path that sets its register count
template <bool SHORT = false>
static __device__ ExtRes extend_any(const DevIndex& ix, const MemOpt& opt, const ExtLds& L, int lane,
                                    int qlen, int q0, int qstep, int tlen, int64_t t0, int tstep,
                                    int w, int end_bonus, int zdrop, int h0, unsigned long long& n_cells, bool try_diag, bool pk2 = true)
{
    { ExtRes r; if (try_diag && extend_diag(ix, opt, L.query, lane, qlen, q0, qstep, tlen, t0, tstep, zdrop, h0, r)) return r; }
    if (qlen + 1 <= WAVE) return extend_wave_reg<1>(ix, opt, L.query, lane, qlen, q0, qstep, tlen, t0, tstep, w, end_bonus, zdrop, h0, n_cells);
    if (pk2 && extend_pk2_ok(opt, qlen, h0, score_max(opt))) return extend_wave_pk2(ix, opt, L.query, lane, qlen, q0, qstep, tlen, t0, tstep, w, end_bonus, zdrop, h0, n_cells);
    if (qlen + 1 <= 2 * WAVE) return extend_wave_reg<2>(ix, opt, L.query, lane, qlen, q0, qstep, tlen, t0, tstep, w, end_bonus, zdrop, h0, n_cells);
    if (SHORT || qlen + 1 <= 3 * WAVE) return extend_wave_reg<3>(ix, opt, L.query, lane, qlen, q0, qstep, tlen, t0, tstep, w, end_bonus, zdrop, h0, n_cells);
    if (pk2 && L.rm != 0x7fffffff && extend_pkl_ok(opt, L.rm + 1, qlen, w, h0, score_max(opt)))
        return extend_wave_pkl(ix, opt, L, lane, qlen, q0, qstep, tlen, t0, tstep, w, end_bonus, zdrop, h0, n_cells);
    return extend_wave(ix, opt, L, lane, qlen, q0, qstep, tlen, t0, tstep, w, end_bonus, zdrop, h0, n_cells);
}

struct U64Lt { __device__ bool operator()(uint64_t a, uint64_t b) const { return a < b; } };

// entries of the row rings of the general form: a power of two >= 2 w + 4 for the widest band tried (opt.w << (MAX_BAND_TRY - 1)),
// never more than the read needs
static __host__ __device__ inline int extend_ring(const MemOpt& opt, int max_len)
{
    long long need = 4ll * (opt.w > 0 ? opt.w : 0) + 8;
    if (need > (long long)max_len + 4) need = (long long)max_len + 4;
    int r = 64;
    while (r < need) r <<= 1;
    return r;
}

#define MAX_BAND_TRY 2

// minimum resident waves per SIMD the register allocator must leave room for (the DP rows are one long dependent
// chain per wave, so latency hiding comes from co-resident waves)
#ifndef K_EXTEND_MIN_WAVES
#define K_EXTEND_MIN_WAVES 8
#endif
// HBM = false: one workgroup per read, the rows of the general DP form in LDS (every read up to ~12 000 bases).
// HBM = true: reads whose rows do not fit a CU's LDS -- a bounded grid walks the reads and each workgroup keeps its rows in
// its own slice of tv.dp_rows (global memory; only the read itself stays in LDS).  Same code, same results, slower rows.
template <bool HBM, bool SHORT = false>
static __device__ __forceinline__ void extend_read(const DevIndex& ix, const MemOpt& opt, const TileView& tv, int32_t* smem, const int r, const int lane)
{
    const int64_t s0 = tv.seed_off[r];
    const int l_query = (int)(tv.seq_off[r + 1] - tv.seq_off[r] - 1);
    const int cap = tv.max_len + 2;
    ExtLds L;
    uint8_t* sq;
    if (HBM) {
        int32_t* rows = tv.dp_rows + (size_t)blockIdx.x * 3 * (size_t)cap;
        L.eh_h = rows; L.eh_e = rows + cap; L.tmpM = rows + 2 * cap; L.rm = 0x7fffffff;
        sq = (uint8_t*)smem;
    } else {                                                    // the read first: tiles of short reads (register form only) allocate nothing else
        sq = (uint8_t*)smem;
        int32_t* rows = smem + (((size_t)cap + 15) & ~(size_t)15) / 4;
        const int ring = extend_ring(opt, tv.max_len);
        L.eh_h = rows; L.eh_e = rows + ring; L.tmpM = rows + 2 * ring; L.rm = ring - 1;
    }
    L.query = sq;
    for (int j = lane; j < l_query; j += WAVE) sq[j] = tv.seq[tv.seq_off[r] + j];
    __syncthreads();

    const int n_chn = tv.n_chains[r];
    const Chain* chains = tv.chains + s0;
    AlnReg* regs = tv.regs + s0;
    uint64_t* srt = tv.srt + s0;
    const int64_t l_pac = ix.l_pac;
    int n_regs = 0;
    unsigned long long n_cells = 0;
    const bool try_diag = !(tv.debug & 0x100);               // BWAMEM_HIP_DEBUGK=256: every extension through the DP (measurements, tests)

    for (int ci = 0; ci < n_chn; ++ci) {
        const Chain c = chains[ci];
        const Seed* seeds = tv.cseeds + s0 + c.seed0;
        if (c.n == 0) continue;
        int64_t rmax0 = l_pac << 1, rmax1 = 0;
        for (int i = 0; i < c.n; ++i) {
            const Seed t = seeds[i];
            int64_t b = t.rbeg - (t.qbeg + cal_max_gap(opt, t.qbeg));
            int64_t e = t.rbeg + t.len + ((l_query - t.qbeg - t.len) + cal_max_gap(opt, l_query - t.qbeg - t.len));
            rmax0 = rmax0 < b ? rmax0 : b;
            rmax1 = rmax1 > e ? rmax1 : e;
        }
        rmax0 = rmax0 > 0 ? rmax0 : 0;
        rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
        if (rmax0 < l_pac && l_pac < rmax1) {
            if (seeds[0].rbeg < l_pac) rmax1 = l_pac;
            else rmax0 = l_pac;
        }
        { int rid; bns_clamp(ix, rmax0, seeds[0].rbeg, rmax1, rid); }

        if (lane == 0) {
            for (int i = 0; i < c.n; ++i) srt[i] = (uint64_t)(uint32_t)seeds[i].score << 32 | (uint32_t)i;
            ks_introsort((size_t)c.n, srt, U64Lt());
        }
        __syncthreads();

        for (int k = c.n - 1; k >= 0; --k) {
            const Seed s = seeds[(uint32_t)srt[k]];
            int i = n_regs;
            // already covered by an earlier region?  Upstream walks the read's regions in order and stops at the first that
            // contains the seed near its own diagonal; a read in a repeat family has hundreds of regions by now, so the lanes
            // test 64 of them at a time and the first hit is the lowest set bit
            for (int base = 0; base < n_regs; base += WAVE) {
                const int ii = base + lane;
                bool hit = false;
                if (ii < n_regs) {
                    const AlnReg* pp = regs + ii;
                    const int64_t prb = pp->rb, pre = pp->re;
                    const int pqb = pp->qb, pqe = pp->qe;
                    if (!(s.rbeg < prb || s.rbeg + s.len > pre || s.qbeg < pqb || s.qbeg + s.len > pqe) && !(s.len - pp->seedlen0 > .1 * l_query)) {
                        const int pw = pp->w;
                        int64_t rd; int qd, w, max_gap;
                        qd = s.qbeg - pqb; rd = s.rbeg - prb;
                        max_gap = cal_max_gap(opt, qd < rd ? qd : (int)rd);
                        w = max_gap < pw ? max_gap : pw;
                        hit = qd - rd < w && rd - qd < w;
                        if (!hit) {
                            qd = pqe - (s.qbeg + s.len); rd = pre - (s.rbeg + s.len);
                            max_gap = cal_max_gap(opt, qd < rd ? qd : (int)rd);
                            w = max_gap < pw ? max_gap : pw;
                            hit = qd - rd < w && rd - qd < w;
                        }
                    }
                }
                const unsigned long long bal = wave_ballot(hit);
                if (bal) { i = base + __ffsll((long long)bal) - 1; break; }
            }
            if (i < n_regs) {
                for (i = k + 1; i < c.n; ++i) {       // an overlapping off-diagonal seed forces extension
                    if (srt[i] == 0) continue;
                    const Seed t = seeds[(uint32_t)srt[i]];
                    if (t.len < s.len * .95) continue;
                    if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) break;
                    if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) break;
                }
                if (i == c.n) {
                    __syncthreads();
                    if (lane == 0) srt[k] = 0;
                    __syncthreads();
                    continue;
                }
            }

            AlnReg a;
            a.rb = a.re = 0; a.qb = a.qe = 0; a.sub = a.alt_sc = a.csub = a.sub_n = 0; a.seedcov = 0;
            a.secondary = a.secondary_all = 0; a.seedlen0 = 0; a.n_comp = 0; a.is_alt = 0; a.frac_rep = 0.f; a.pad_ = 0; a.hash = 0;
            int aw0 = opt.w, aw1 = opt.w;
            a.w = opt.w;
            a.score = a.truesc = -1;
            a.rid = c.rid;

            if (s.qbeg) {                              // left extension, both sequences reversed
                int64_t tmp = s.rbeg - rmax0;
                ExtRes e; e.score = e.qle = e.tle = e.gtle = e.gscore = e.max_off = 0;
                for (i = 0; i < MAX_BAND_TRY; ++i) {
                    int prev = a.score;
                    aw0 = opt.w << i;
                    e = extend_any<SHORT>(ix, opt, L, lane, s.qbeg, s.qbeg - 1, -1, (int)tmp, s.rbeg - 1, -1,
                                    aw0, opt.pen_clip5, opt.zdrop, s.len * opt.a, n_cells, try_diag);
                    a.score = e.score;
                    if (a.score == prev || e.max_off < (aw0 >> 1) + (aw0 >> 2)) break;
                }
                if (e.gscore <= 0 || e.gscore <= a.score - opt.pen_clip5) {
                    a.qb = s.qbeg - e.qle; a.rb = s.rbeg - e.tle;
                    a.truesc = a.score;
                } else {
                    a.qb = 0; a.rb = s.rbeg - e.gtle;
                    a.truesc = e.gscore;
                }
            } else { a.score = a.truesc = s.len * opt.a; a.qb = 0; a.rb = s.rbeg; }

            if (s.qbeg + s.len != l_query) {           // right extension
                int qe = s.qbeg + s.len, sc0 = a.score;
                int64_t re = s.rbeg + s.len - rmax0;
                ExtRes e; e.score = e.qle = e.tle = e.gtle = e.gscore = e.max_off = 0;
                for (i = 0; i < MAX_BAND_TRY; ++i) {
                    int prev = a.score;
                    aw1 = opt.w << i;
                    e = extend_any<SHORT>(ix, opt, L, lane, l_query - qe, qe, 1, (int)(rmax1 - rmax0 - re), rmax0 + re, 1,
                                    aw1, opt.pen_clip3, opt.zdrop, sc0, n_cells, try_diag);
                    a.score = e.score;
                    if (a.score == prev || e.max_off < (aw1 >> 1) + (aw1 >> 2)) break;
                }
                if (e.gscore <= 0 || e.gscore <= a.score - opt.pen_clip3) {
                    a.qe = qe + e.qle; a.re = rmax0 + re + e.tle;
                    a.truesc += a.score - sc0;
                } else {
                    a.qe = l_query; a.re = rmax0 + re + e.gtle;
                    a.truesc += e.gscore - sc0;
                }
            } else { a.qe = l_query; a.re = s.rbeg + s.len; }

            a.seedcov = 0;
            for (i = 0; i < c.n; ++i) {
                const Seed t = seeds[i];
                if (t.qbeg >= a.qb && t.qbeg + t.len <= a.qe && t.rbeg >= a.rb && t.rbeg + t.len <= a.re)
                    a.seedcov += t.len;
            }
            a.w = aw0 > aw1 ? aw0 : aw1;
            a.seedlen0 = s.len;
            a.frac_rep = c.frac_rep;
            __syncthreads();
            if (lane == 0) regs[n_regs] = a;
            ++n_regs;
            __syncthreads();
        }
    }
    if (lane == 0) {
        tv.n_regs[r] = n_regs;
        count_add(&tv.cnt->n_dp_cells, n_cells);
    }
}

template <bool HBM>
__global__ void __launch_bounds__(64, K_EXTEND_MIN_WAVES) k_extend(DevIndex ix, MemOpt opt, TileView tv)
{
    HIP_DYNAMIC_SHARED(int32_t, smem)
    if (!HBM) { extend_read<false>(ix, opt, tv, smem, tv.order ? tv.order[blockIdx.x] : (int)blockIdx.x, threadIdx.x); return; }
    for (int r = blockIdx.x; r < tv.n_reads; r += (int)gridDim.x) {
        extend_read<true>(ix, opt, tv, smem, r, threadIdx.x);
        __syncthreads();                                        // the next read reuses the rows and the staged query
    }
}

// Tiles of reads of at most 191 bases (every query takes a register form): the same kernel without the LDS forms
__global__ void __launch_bounds__(64, K_EXTEND_MIN_WAVES) k_extend_short(DevIndex ix, MemOpt opt, TileView tv)
{
    HIP_DYNAMIC_SHARED(int32_t, smem)
    extend_read<false, true>(ix, opt, tv, smem, tv.order ? tv.order[blockIdx.x] : (int)blockIdx.x, threadIdx.x);
}

// Tiles whose reads keep rows in LDS (beyond 191 bases): the rows and the staged read bound the resident waves (16 KB per
// 10 kb read: two or three waves per SIMD), so the register allocator gets that room -- 144 registers and no spills instead
// of 64 with 92 spilled (k_extend 2.55 -> 2.30 s per 200 k reads of 10 kb).
__global__ void __launch_bounds__(64, 2) k_extend_long(DevIndex ix, MemOpt opt, TileView tv)
{
    HIP_DYNAMIC_SHARED(int32_t, smem)
    extend_read<false>(ix, opt, tv, smem, tv.order ? tv.order[blockIdx.x] : (int)blockIdx.x, threadIdx.x);
}

// LDS of one k_extend workgroup for reads of up to max_len bases: the H, E and M rows of the general form + the read
size_t extend_lds_bytes(const MemOpt& opt, int max_len)
{
    size_t cap = (size_t)max_len + 2;
    return 3 * (size_t)extend_ring(opt, max_len) * sizeof(int32_t) + ((cap + 15) & ~(size_t)15);
}
void launch_extend(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv)
{
    if (tv.n_reads <= 0) return;
    if (tv.ext_hbm && tv.dp_rows) {                             // rows in global memory: tv.dp_rows_blocks slices of 3 x (max_len + 2) ints
        const size_t cap = (size_t)tv.max_len + 2;
        const int grid = tv.n_reads < tv.dp_rows_blocks ? tv.n_reads : tv.dp_rows_blocks;
        hipLaunchKernelGGL(k_extend<true>, dim3(grid), dim3(64), (cap + 15) & ~(size_t)15, st, ix, opt, tv);
        return;
    }
    // every query of a tile whose reads are at most 3 * 64 - 1 bases long takes the register form: no rows in LDS, only the
    // read -- which matters for overlap, because k_seed fills the CUs' LDS and a workgroup that asks for 2 KB finds no room
    if (tv.max_len + 1 <= 3 * WAVE) { hipLaunchKernelGGL(k_extend_short, dim3(tv.n_reads), dim3(64), ((size_t)tv.max_len + 2 + 15) & ~(size_t)15, st, ix, opt, tv); return; }
    const size_t shmem = extend_lds_bytes(opt, tv.max_len);
    if (shmem >= 8192) hipLaunchKernelGGL(k_extend_long, dim3(tv.n_reads), dim3(64), shmem, st, ix, opt, tv);
    else hipLaunchKernelGGL(k_extend<false>, dim3(tv.n_reads), dim3(64), shmem, st, ix, opt, tv);
}
