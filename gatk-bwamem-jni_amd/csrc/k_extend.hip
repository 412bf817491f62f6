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
#include "kernels.h"

struct ExtRes { int score, qle, tle, gtle, gscore, max_off; };

struct ExtLds { int32_t* eh_h; int32_t* eh_e; int32_t* tmpM; const uint8_t* query; };

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
    // first row: decay from h0 by insertion costs
    for (int j = lane; j <= qlen; j += WAVE) {
        int v = 0;
        if (j == 0) v = h0;
        else if (h0 > oe_ins) {
            int vj = h0 - oe_ins - (j - 1) * e_ins;       // value if the decay chain reaches column j
            if (j == 1 || vj + e_ins > e_ins) v = vj;     // previous cell > e_ins
        }
        L.eh_h[j] = v; L.eh_e[j] = 0;
    }
    const int mx = score_max(opt);
    {   // clip the band by the longest affordable gap
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
    __syncthreads();
    for (i = 0; i < tlen; ++i) {
        if ((i & 63) == 0) {                               // 64 target bases, one per lane
            int ii = i + lane;
            tch = ii < tlen ? ref_base2(ix, t0 + (int64_t)tstep * ii) : 4;
        }
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
                int Mp = L.eh_h[j];
                int qc = L.query[q0 + qstep * j];
                int sc = qc == 0 ? ms0 : qc == 1 ? ms1 : qc == 2 ? ms2 : qc == 3 ? ms3 : ms4;
                L.tmpM[j] = Mp ? Mp + sc : 0;
            }
        }
        __syncthreads();
        // phase B: F by prefix scan, H, E, row maximum; writes eh_h[j+1], eh_e[j]
        int fcarry = 0;
        for (int c = beg; c < end; c += WAVE) {
            int j = c + lane;
            bool act = j < end;
            int M = act ? L.tmpM[j] : 0;
            int e = act ? L.eh_e[j] : 0;
            int t = M - oe_ins; t = t > 0 ? t : 0;
            int U = act ? t + j * e_ins : NEG_INF_I32;
            int P = wave_prefix_max(U, lane);
            int Pex = __shfl_up(P, 1);
            int f = fcarry - (j - c) * e_ins;
            if (lane > 0) { int g = Pex - (j - 1) * e_ins; f = f > g ? f : g; }
            int h = M > e ? M : e;
            h = h > f ? h : f;
            if (!act) h = -1;
            int mc = wave_max(h);
            unsigned long long bal = wave_ballot(act && h == mc);
            int mjc = c + 63 - __clzll(bal);
            if (mc >= m) { m = mc; mj = mjc; }
            int last = (end - 1 - c) < 63 ? (end - 1 - c) : 63;
            hlast = wave_bcast(h, last);
            int Plast = wave_bcast(P, 63);
            {   // F at column c+64 for the next chunk
                int f1 = fcarry - WAVE * e_ins, f2 = Plast - (c + 63) * e_ins;
                fcarry = f1 > f2 ? f1 : f2;
            }
            if (act) {
                int t2 = M - oe_del; t2 = t2 > 0 ? t2 : 0;
                int en = e - e_del; en = en > t2 ? en : t2;
                L.eh_e[j] = en;
                L.eh_h[j + 1] = h;
            }
        }
        if (end > beg) {                                   // eh[beg].h = first-column value, eh[end].h = H(i,end-1)
            n_cells += (unsigned long long)(end - beg);
            h1 = hlast;
            if (lane == 0) L.eh_h[beg] = h1i;
        } else if (lane == 0) L.eh_h[end] = h1i;
        if (lane == 0) L.eh_e[end] = 0;
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
                int nz = j < end && (L.eh_h[j] != 0 || L.eh_e[j] != 0);
                unsigned long long bal = wave_ballot(nz);
                if (bal) { nb = c + __ffsll((long long)bal) - 1; break; }
            }
            beg = nb;
            int jl = beg - 1;
            for (int hi = end; hi >= beg; hi -= WAVE) {
                int p = hi - lane;
                int nz = p >= beg && (L.eh_h[p] != 0 || L.eh_e[p] != 0);
                unsigned long long bal = wave_ballot(nz);
                if (bal) { jl = hi - (__ffsll((long long)bal) - 1); break; }
            }
            end = jl + 2 < qlen ? jl + 2 : qlen;
        }
        __syncthreads();
        if (gscore > 0 && m + mx * (qlen - 1 - mj) <= max) {   // see ext_bound(): the remaining rows cannot change the result
            int B = 0;
            for (int c = beg; c < qlen; c += WAVE) {
                const int j = c + lane;
                const int term = j < qlen ? ext_bound_term(L.eh_h[j], L.eh_e[j], mx, qlen - 1 - j) : 0;
                const int bc = wave_max(term);
                B = B > bc ? B : bc;
            }
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
    int ehh[C], ehe[C], scn[C];
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
        score_lane(ST, j < qlen ? query[q0 + qstep * j] : 4, scp[t], scn[t]);
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
        int m = 0, mj = -1, h1, h1i;
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        if (beg == 0) { h1i = h0 - (o_del + e_del * (i + 1)); if (h1i < 0) h1i = 0; }
        else h1i = 0;
        h1 = h1i;
        int hnew[C], enew[C];
        int fcarry = NEG_INF_I32, prev_last = 0;
#pragma unroll
        for (int t = 0; t < C; ++t) {
            const int c0 = t * WAVE, j = c0 + lane;
            const bool act = j >= beg && j < end;
            const int Mp = ehh[t], e = ehe[t];
            const int sc = score_at(scp[t], scn[t], tb);
            const int M = act && Mp ? Mp + sc : 0;
            int tt = M - oe_ins; tt = tt > 0 ? tt : 0;
            const int U = act ? tt + j * e_ins : NEG_INF_I32;
            const int P = dpp_prefix_max(U, NEG_INF_I32);
            const int Pex = dpp_shr1(P, NEG_INF_I32);
            int f = fcarry - (j - c0) * e_ins;
            { int g = Pex - (j - 1) * e_ins; f = f > g ? f : g; }
            if (j == beg) f = 0;
            int h = M > e ? M : e;
            h = h > f ? h : f;
            if (!act) h = -1;
            const int mc = wave_readlane(dpp_prefix_max(h, -1), 63);
            const unsigned long long bal = wave_ballot(act && h == mc);
            if (bal && mc >= m) { m = mc; mj = c0 + 63 - __clzll((long long)bal); }
            {
                const int Plast = wave_readlane(P, 63);
                const int f1 = fcarry - WAVE * e_ins, f2 = Plast - (c0 + 63) * e_ins;
                fcarry = f1 > f2 ? f1 : f2;
            }
            int t2 = M - oe_del; t2 = t2 > 0 ? t2 : 0;
            int en = e - e_del; en = en > t2 ? en : t2;
            hnew[t] = h; enew[t] = en;
            // h of column end-1 (needed below) lives in this chunk?
            if (end > beg && ((end - 1) >> 6) == t) h1 = wave_readlane(h, (end - 1) & 63);
        }
#pragma unroll
        for (int t = 0; t < C; ++t) {                      // in-place row update: eh[beg].h = h1i, eh[j+1].h = H(i,j), eh[end].e = 0
            const int j = t * WAVE + lane;
            const int hsh = dpp_shr1(hnew[t], prev_last);
            prev_last = wave_readlane(hnew[t], 63);
            if (end > beg) {
                if (j == beg) ehh[t] = h1i;
                else if (j > beg && j <= end) ehh[t] = hsh;
                if (j >= beg && j < end) ehe[t] = enew[t];
                else if (j == end) ehe[t] = 0;
            } else if (j == end) { ehh[t] = h1i; ehe[t] = 0; }
        }
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

// picks the register-resident form when the query fits
static __device__ ExtRes extend_any(const DevIndex& ix, const MemOpt& opt, const ExtLds& L, int lane,
                                    int qlen, int q0, int qstep, int tlen, int64_t t0, int tstep,
                                    int w, int end_bonus, int zdrop, int h0, unsigned long long& n_cells)
{
    if (qlen + 1 <= WAVE) return extend_wave_reg<1>(ix, opt, L.query, lane, qlen, q0, qstep, tlen, t0, tstep, w, end_bonus, zdrop, h0, n_cells);
    if (qlen + 1 <= 2 * WAVE) return extend_wave_reg<2>(ix, opt, L.query, lane, qlen, q0, qstep, tlen, t0, tstep, w, end_bonus, zdrop, h0, n_cells);
    if (qlen + 1 <= 3 * WAVE) return extend_wave_reg<3>(ix, opt, L.query, lane, qlen, q0, qstep, tlen, t0, tstep, w, end_bonus, zdrop, h0, n_cells);
    return extend_wave(ix, opt, L, lane, qlen, q0, qstep, tlen, t0, tstep, w, end_bonus, zdrop, h0, n_cells);
}

struct U64Lt { __device__ bool operator()(uint64_t a, uint64_t b) const { return a < b; } };

#define MAX_BAND_TRY 2

// minimum resident waves per SIMD the register allocator must leave room for (the DP rows are one long dependent
// chain per wave, so latency hiding comes from co-resident waves)
#ifndef K_EXTEND_MIN_WAVES
#define K_EXTEND_MIN_WAVES 8
#endif
// read_list != 0: the workgroups take their reads from this list (left-overs of the round scheme below)
__global__ void __launch_bounds__(64, K_EXTEND_MIN_WAVES) k_extend(DevIndex ix, MemOpt opt, TileView tv, const int32_t* read_list)
{
    HIP_DYNAMIC_SHARED(int32_t, smem)
    const int r = read_list ? read_list[blockIdx.x] : (int)blockIdx.x, lane = threadIdx.x;
    const int64_t s0 = tv.seed_off[r];
    const int l_query = (int)(tv.seq_off[r + 1] - tv.seq_off[r] - 1);
    const int cap = tv.max_len + 2;
    ExtLds L;
    L.eh_h = smem; L.eh_e = smem + cap; L.tmpM = smem + 2 * cap;
    uint8_t* sq = (uint8_t*)(smem + 3 * cap);
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
            int i;
            for (i = 0; i < n_regs; ++i) {            // already covered by an earlier region?
                const AlnReg p = regs[i];
                int64_t rd; int qd, w, max_gap;
                if (s.rbeg < p.rb || s.rbeg + s.len > p.re || s.qbeg < p.qb || s.qbeg + s.len > p.qe) continue;
                if (s.len - p.seedlen0 > .1 * l_query) continue;
                qd = s.qbeg - p.qb; rd = s.rbeg - p.rb;
                max_gap = cal_max_gap(opt, qd < rd ? qd : (int)rd);
                w = max_gap < p.w ? max_gap : p.w;
                if (qd - rd < w && rd - qd < w) break;
                qd = p.qe - (s.qbeg + s.len); rd = p.re - (s.rbeg + s.len);
                max_gap = cal_max_gap(opt, qd < rd ? qd : (int)rd);
                w = max_gap < p.w ? max_gap : p.w;
                if (qd - rd < w && rd - qd < w) break;
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
                    e = extend_any(ix, opt, L, lane, s.qbeg, s.qbeg - 1, -1, (int)tmp, s.rbeg - 1, -1,
                                    aw0, opt.pen_clip5, opt.zdrop, s.len * opt.a, n_cells);
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
                    e = extend_any(ix, opt, L, lane, l_query - qe, qe, 1, (int)(rmax1 - rmax0 - re), rmax0 + re, 1,
                                    aw1, opt.pen_clip3, opt.zdrop, sc0, n_cells);
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


// ------------------------------------------------------------------------------------------------------------------
// Lane-per-read form (short reads).  The wave-per-read form above spends ~3 vector instructions per DP cell on scans,
// cross-lane moves and window bookkeeping that exist only because one row is spread over the lanes.  Here every lane
// runs the scalar recurrence of ksw_extend2 for its own read -- 64 reads per wavefront, ~25 vector instructions per
// 64 cells -- with the (H, E) row of each lane in LDS as one word per query column: h:14 | e:14 | query base:4, laid
// out [column][lane].  Scores are non-negative and bounded by read length x best match score, which the launcher
// checks against the 14 bits; longer reads and larger scores take the wave-per-read kernel.
struct RowTab { uint32_t p[5]; int n[5]; };            // per target base: scores against query bases 0..3 (bytes), and against N
DEV RowTab row_tab(const MemOpt& opt)
{
    RowTab T;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        T.p[t] = (uint32_t)(uint8_t)opt.mat[t * 5] | (uint32_t)(uint8_t)opt.mat[t * 5 + 1] << 8 | (uint32_t)(uint8_t)opt.mat[t * 5 + 2] << 16 | (uint32_t)(uint8_t)opt.mat[t * 5 + 3] << 24;
        T.n[t] = opt.mat[t * 5 + 4];
    }
    return T;
}

#define LH_MASK 0x3fffu
#define LROW(j) eh[(j) << 6]

static __device__ ExtRes extend_lane(const DevIndex& ix, const MemOpt& opt, const RowTab& RT, int mx, uint32_t* eh, const uint8_t* query,
                                     int qlen, int q0, int qstep, int tlen, int64_t t0, int tstep,
                                     int w, int end_bonus, int zdrop, int h0, unsigned long long& n_cells)
{
    const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    int i, j, beg, end, max, max_i, max_j, max_ins, max_del, max_ie, gscore, max_off;
    if (h0 < 0) h0 = 0;
    {   // first row: decay from h0 by insertion costs; the query base of each column rides along in the same word
        int prev = h0 > oe_ins ? h0 - oe_ins : 0;
        LROW(0) = (uint32_t)h0 | (uint32_t)(qlen > 0 ? query[q0] : 4) << 28;
        for (j = 1; j <= qlen; ++j) {
            const uint32_t qb = j < qlen ? query[q0 + qstep * j] : 4u;
            LROW(j) = (uint32_t)prev | qb << 28;
            prev = prev > e_ins ? prev - e_ins : 0;
        }
    }
    {   // clip the band by the longest affordable gap
        max_ins = div_plus(qlen * mx + end_bonus - o_ins, e_ins, 1);
        max_ins = max_ins > 1 ? max_ins : 1;
        w = w < max_ins ? w : max_ins;
        max_del = div_plus(qlen * mx + end_bonus - o_del, e_del, 1);
        max_del = max_del > 1 ? max_del : 1;
        w = w < max_del ? w : max_del;
    }
    max = h0; max_i = max_j = -1; max_ie = -1; gscore = -1; max_off = 0;
    beg = 0; end = qlen;
    PacCache pc; pc.w = -1; pc.v = 0;
    for (i = 0; i < tlen; ++i) {
        const int tb = ref_base2_c(ix, pc, t0 + (int64_t)tstep * i);
        const uint32_t R = tb == 0 ? RT.p[0] : tb == 1 ? RT.p[1] : tb == 2 ? RT.p[2] : RT.p[3];
        const int Rn = tb == 0 ? RT.n[0] : tb == 1 ? RT.n[1] : tb == 2 ? RT.n[2] : RT.n[3];
        int f = 0, h1, m = 0, mj = -1;
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
        else h1 = 0;
        for (j = beg; j < end; ++j) {
            const uint32_t wd = LROW(j);
            int M = (int)(wd & LH_MASK), e = (int)(wd >> 14 & LH_MASK);
            const uint32_t q = wd >> 28;
            const int sc = q < 4 ? (int)(int8_t)(R >> (q << 3)) : Rn;
            M = M ? M + sc : 0;
            int h = M > e ? M : e;
            h = h > f ? h : f;
            mj = m > h ? mj : j;
            m = m > h ? m : h;
            int t = M - oe_del; t = t > 0 ? t : 0;
            e -= e_del; e = e > t ? e : t;
            LROW(j) = (uint32_t)h1 | (uint32_t)e << 14 | q << 28;       // H(i,j-1) for the next row, E(i+1,j)
            h1 = h;
            t = M - oe_ins; t = t > 0 ? t : 0;
            f -= e_ins; f = f > t ? f : t;
        }
        LROW(end) = (LROW(end) & 0xf0000000u) | (uint32_t)h1;            // eh[end].h = h1, eh[end].e = 0
        if (end > beg) n_cells += (unsigned long long)(end - beg);
        if ((end > beg ? end : beg) == qlen) {
            max_ie = gscore > h1 ? max_ie : i;
            gscore = gscore > h1 ? gscore : h1;
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
        for (j = beg; j < end && (LROW(j) & 0x0fffffffu) == 0; ++j);      // shrink the window to the non-zero span
        beg = j;
        for (j = end; j >= beg && (LROW(j) & 0x0fffffffu) == 0; --j);
        end = j + 2 < qlen ? j + 2 : qlen;
        if (gscore > 0 && m + mx * (qlen - 1 - mj) <= max) {               // see ext_bound_term(): the remaining rows cannot change the result
            int B = 0;
            for (j = beg; j < qlen; ++j) {
                const uint32_t wd = LROW(j);
                const int term = ext_bound_term((int)(wd & LH_MASK), (int)(wd >> 14 & LH_MASK), mx, qlen - 1 - j);
                B = B > term ? B : term;
            }
            if (beg == 0) { const int hb = h0 - (o_del + e_del * (i + 2)); if (hb > 0 && hb + mx * qlen > B) B = hb + mx * qlen; }
            if (B <= max && B < gscore) break;
        }
    }
    ExtRes r;
    r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
    return r;
}

#undef LROW
#undef LH_MASK

// ------------------------------------------------------------------------------------------------------------------
// Extension in rounds (single-end tiles of short reads).  Running extend_lane() from inside the per-read control flow
// leaves a wave waiting on its slowest lane at every loop level.  Instead the control flow of mem_chain2aln is made
// resumable (k_ext_ctrl: one lane per read, state in global memory) and stops at every ksw_extend2 call, emitting a task;
// the tasks of a round are binned by query length and k_ext_dp runs one task per lane over a bin-sorted list, so the
// lanes of a wave do rows and columns of similar extent.  A read needs one round per extension call (typically
// left + right of one seed).  Reads still unfinished after EXT_ROUNDS rounds are redone from scratch by the
// wave-per-read kernel (same results; their partial regions are simply overwritten).
struct ExtTask { int64_t t0; int32_t qlen, q0, qstep, tlen, tstep, w, end_bonus, h0; };
struct ExtState {
    int64_t rmax0, rmax1, a_rb, a_re;
    int32_t phase, ci, k, n_regs, try_i, prev, sc0, aw0, aw1, a_score, a_truesc, a_qb, a_qe, pad_;
};
enum { XP_CHAIN = 0, XP_SEED = 1, XP_LEFT = 2, XP_RIGHT_START = 3, XP_RIGHT = 4, XP_STORE = 5, XP_DONE = 6 };
#define EXT_BINS 20                     // query-length bins of width 8
struct BinStart { int32_t s[EXT_BINS + 1]; };   // exclusive prefix of the bin counts of a round

// advance one read until it needs an extension (returns true, task filled) or is finished (returns false)
static __device__ bool ext_advance(const DevIndex& ix, const MemOpt& opt, const TileView& tv, int r, ExtState& S, const ExtRes& e, ExtTask& task)
{
    const int64_t s0 = tv.seed_off[r];
    const int l_query = (int)(tv.seq_off[r + 1] - tv.seq_off[r] - 1);
    const int n_chn = tv.n_chains[r];
    const Chain* chains = tv.chains + s0;
    AlnReg* regs = tv.regs + s0;
    uint64_t* srt = tv.srt + s0;
    const int64_t l_pac = ix.l_pac;
    for (;;) {
        if (S.phase == XP_CHAIN) {
            if (S.ci >= n_chn) { S.phase = XP_DONE; return false; }
            const Chain c = chains[S.ci];
            if (c.n == 0) { ++S.ci; continue; }
            const Seed* seeds = tv.cseeds + s0 + c.seed0;
            int64_t rmax0 = l_pac << 1, rmax1 = 0;
            for (int i = 0; i < c.n; ++i) {
                const Seed t = seeds[i];
                int64_t b = t.rbeg - (t.qbeg + cal_max_gap(opt, t.qbeg));
                int64_t en = t.rbeg + t.len + ((l_query - t.qbeg - t.len) + cal_max_gap(opt, l_query - t.qbeg - t.len));
                rmax0 = rmax0 < b ? rmax0 : b;
                rmax1 = rmax1 > en ? rmax1 : en;
            }
            rmax0 = rmax0 > 0 ? rmax0 : 0;
            rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
            if (rmax0 < l_pac && l_pac < rmax1) {
                if (seeds[0].rbeg < l_pac) rmax1 = l_pac;
                else rmax0 = l_pac;
            }
            { int rid; bns_clamp(ix, rmax0, seeds[0].rbeg, rmax1, rid); }
            S.rmax0 = rmax0; S.rmax1 = rmax1;
            for (int i = 0; i < c.n; ++i) srt[i] = (uint64_t)(uint32_t)seeds[i].score << 32 | (uint32_t)i;
            ks_introsort((size_t)c.n, srt, U64Lt());
            S.k = c.n - 1;
            S.phase = XP_SEED;
            continue;
        }
        const Chain c = chains[S.ci];
        const Seed* seeds = tv.cseeds + s0 + c.seed0;
        if (S.phase == XP_SEED) {
            if (S.k < 0) { ++S.ci; S.phase = XP_CHAIN; continue; }
            const Seed s = seeds[(uint32_t)srt[S.k]];
            int i;
            for (i = 0; i < S.n_regs; ++i) {            // already covered by an earlier region?
                const AlnReg p = regs[i];
                int64_t rd; int qd, w, max_gap;
                if (s.rbeg < p.rb || s.rbeg + s.len > p.re || s.qbeg < p.qb || s.qbeg + s.len > p.qe) continue;
                if (s.len - p.seedlen0 > .1 * l_query) continue;
                qd = s.qbeg - p.qb; rd = s.rbeg - p.rb;
                max_gap = cal_max_gap(opt, qd < rd ? qd : (int)rd);
                w = max_gap < p.w ? max_gap : p.w;
                if (qd - rd < w && rd - qd < w) break;
                qd = p.qe - (s.qbeg + s.len); rd = p.re - (s.rbeg + s.len);
                max_gap = cal_max_gap(opt, qd < rd ? qd : (int)rd);
                w = max_gap < p.w ? max_gap : p.w;
                if (qd - rd < w && rd - qd < w) break;
            }
            if (i < S.n_regs) {
                for (i = S.k + 1; i < c.n; ++i) {       // an overlapping off-diagonal seed forces extension
                    if (srt[i] == 0) continue;
                    const Seed t = seeds[(uint32_t)srt[i]];
                    if (t.len < s.len * .95) continue;
                    if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) break;
                    if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) break;
                }
                if (i == c.n) { srt[S.k] = 0; --S.k; continue; }
            }
            S.aw0 = S.aw1 = opt.w;
            S.a_score = S.a_truesc = -1;
            S.a_qb = S.a_qe = 0; S.a_rb = S.a_re = 0;
            if (s.qbeg) {                                  // left extension, both sequences reversed
                S.try_i = 0; S.prev = S.a_score;
                task.qlen = s.qbeg; task.q0 = s.qbeg - 1; task.qstep = -1; task.tlen = (int)(s.rbeg - S.rmax0); task.t0 = s.rbeg - 1; task.tstep = -1;
                task.w = opt.w; task.end_bonus = opt.pen_clip5; task.h0 = s.len * opt.a;
                S.phase = XP_LEFT;
                return true;
            }
            S.a_score = S.a_truesc = s.len * opt.a; S.a_qb = 0; S.a_rb = s.rbeg;
            S.phase = XP_RIGHT_START;
            continue;
        }
        const Seed s = seeds[(uint32_t)srt[S.k]];
        if (S.phase == XP_LEFT) {                          // e = result of left try S.try_i
            S.aw0 = opt.w << S.try_i;
            S.a_score = e.score;
            if (!(S.a_score == S.prev || e.max_off < (S.aw0 >> 1) + (S.aw0 >> 2)) && S.try_i + 1 < MAX_BAND_TRY) {
                ++S.try_i; S.prev = S.a_score;
                task.qlen = s.qbeg; task.q0 = s.qbeg - 1; task.qstep = -1; task.tlen = (int)(s.rbeg - S.rmax0); task.t0 = s.rbeg - 1; task.tstep = -1;
                task.w = opt.w << S.try_i; task.end_bonus = opt.pen_clip5; task.h0 = s.len * opt.a;
                return true;
            }
            if (e.gscore <= 0 || e.gscore <= S.a_score - opt.pen_clip5) {
                S.a_qb = s.qbeg - e.qle; S.a_rb = s.rbeg - e.tle;
                S.a_truesc = S.a_score;
            } else {
                S.a_qb = 0; S.a_rb = s.rbeg - e.gtle;
                S.a_truesc = e.gscore;
            }
            S.phase = XP_RIGHT_START;
            continue;
        }
        if (S.phase == XP_RIGHT_START) {
            if (s.qbeg + s.len != l_query) {
                const int qe = s.qbeg + s.len;
                const int64_t re = s.rbeg + s.len - S.rmax0;
                S.sc0 = S.a_score; S.try_i = 0; S.prev = S.a_score;
                task.qlen = l_query - qe; task.q0 = qe; task.qstep = 1; task.tlen = (int)(S.rmax1 - S.rmax0 - re); task.t0 = S.rmax0 + re; task.tstep = 1;
                task.w = opt.w; task.end_bonus = opt.pen_clip3; task.h0 = S.sc0;
                S.phase = XP_RIGHT;
                return true;
            }
            S.a_qe = l_query; S.a_re = s.rbeg + s.len;
            S.phase = XP_STORE;
            continue;
        }
        if (S.phase == XP_RIGHT) {                         // e = result of right try S.try_i
            const int qe = s.qbeg + s.len;
            const int64_t re = s.rbeg + s.len - S.rmax0;
            S.aw1 = opt.w << S.try_i;
            S.a_score = e.score;
            if (!(S.a_score == S.prev || e.max_off < (S.aw1 >> 1) + (S.aw1 >> 2)) && S.try_i + 1 < MAX_BAND_TRY) {
                ++S.try_i; S.prev = S.a_score;
                task.qlen = l_query - qe; task.q0 = qe; task.qstep = 1; task.tlen = (int)(S.rmax1 - S.rmax0 - re); task.t0 = S.rmax0 + re; task.tstep = 1;
                task.w = opt.w << S.try_i; task.end_bonus = opt.pen_clip3; task.h0 = S.sc0;
                return true;
            }
            if (e.gscore <= 0 || e.gscore <= S.a_score - opt.pen_clip3) {
                S.a_qe = qe + e.qle; S.a_re = S.rmax0 + re + e.tle;
                S.a_truesc += S.a_score - S.sc0;
            } else {
                S.a_qe = l_query; S.a_re = S.rmax0 + re + e.gtle;
                S.a_truesc += e.gscore - S.sc0;
            }
            S.phase = XP_STORE;
            continue;
        }
        {   // XP_STORE
            AlnReg a;
            a.sub = a.alt_sc = a.csub = a.sub_n = 0;
            a.secondary = a.secondary_all = 0; a.n_comp = 0; a.is_alt = 0; a.pad_ = 0; a.hash = 0;
            a.rb = S.a_rb; a.re = S.a_re; a.qb = S.a_qb; a.qe = S.a_qe;
            a.score = S.a_score; a.truesc = S.a_truesc; a.rid = c.rid;
            a.seedcov = 0;
            for (int i = 0; i < c.n; ++i) {
                const Seed t = seeds[i];
                if (t.qbeg >= a.qb && t.qbeg + t.len <= a.qe && t.rbeg >= a.rb && t.rbeg + t.len <= a.re)
                    a.seedcov += t.len;
            }
            a.w = S.aw0 > S.aw1 ? S.aw0 : S.aw1;
            a.seedlen0 = s.len;
            a.frac_rep = c.frac_rep;
            regs[S.n_regs] = a;
            ++S.n_regs;
            --S.k;
            S.phase = XP_SEED;
        }
    }
}

// round step 1: consume the previous round's results, emit the next tasks into their query-length bins.
// final != 0 (after the last round): unfinished reads go to the left-over list instead.
__global__ void k_ext_ctrl(DevIndex ix, MemOpt opt, TileView tv, ExtState* states, const ExtRes* res, ExtTask* tasks, int32_t* bin_list, int32_t* bin_cnt,
                           int first, int final, int32_t* left_list, int32_t* left_cnt)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= tv.n_reads) return;
    ExtState S;
    if (first) {
        memset(&S, 0, sizeof S);
        S.phase = XP_CHAIN;
    } else {
        S = states[r];
        if (S.phase == XP_DONE) return;
    }
    ExtRes e; e.score = e.qle = e.tle = e.gtle = e.gscore = e.max_off = 0;
    if (!first) e = res[r];
    ExtTask task;
    const bool need = ext_advance(ix, opt, tv, r, S, e, task);
    states[r] = S;
    if (need && final) left_list[atomicAdd(left_cnt, 1)] = r;   // out of rounds: the wave-per-read kernel redoes this read
    else if (need) {
        tasks[r] = task;
        int b = task.qlen >> 3; b = b < EXT_BINS ? b : EXT_BINS - 1;
        bin_list[(size_t)b * tv.n_reads + atomicAdd(bin_cnt + b, 1)] = r;
    } else tv.n_regs[r] = S.n_regs;
}

// round step 2: one task per lane over the bins [bin0, bin1) laid end to end (bin_start = exclusive prefix of the bin counts)
__global__ void __launch_bounds__(64) k_ext_dp(DevIndex ix, MemOpt opt, TileView tv, const ExtTask* tasks, ExtRes* res, const int32_t* bin_list, BinStart bs,
                                               int bin0, int bin1)
{
    HIP_DYNAMIC_SHARED(uint32_t, lrow)
    const int lane = threadIdx.x;
    uint32_t* eh = lrow + lane;
    const int g = blockIdx.x * 64 + lane + bs.s[bin0];
    unsigned long long n_cells = 0;
    if (g < bs.s[bin1]) {
        int b = bin0;
        for (int k = bin0 + 1; k < bin1; ++k) if (g >= bs.s[k]) b = k;
        const int r = bin_list[(size_t)b * tv.n_reads + (g - bs.s[b])];
        const ExtTask t = tasks[r];
        const RowTab RT = row_tab(opt);
        const int mx = score_max(opt);
        res[r] = extend_lane(ix, opt, RT, mx, eh, tv.seq + tv.seq_off[r], t.qlen, t.q0, t.qstep, t.tlen, t.t0, t.tstep, t.w, t.end_bonus, opt.zdrop, t.h0, n_cells);
    }
    for (int o = 32; o > 0; o >>= 1) n_cells += __shfl_down(n_cells, o);
    if (lane == 0) count_add(&tv.cnt->n_dp_cells, n_cells);
}


void launch_extend(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv)
{
    if (tv.n_reads <= 0) return;
    size_t cap = (size_t)tv.max_len + 2;
    size_t shmem = 3 * cap * sizeof(int32_t) + ((cap + 15) & ~(size_t)15);
    hipLaunchKernelGGL(k_extend, dim3(tv.n_reads), dim3(64), shmem, st, ix, opt, tv, (const int32_t*)0);
}

// ---- extension in rounds: host side
#define EXT_ROUNDS 8
bool extend_rounds_supported(const MemOpt& opt, const TileView& tv)
{   // a lane's row must fit its share of LDS and every score 14 bits
    int mx = 0;
    for (int k = 0; k < 25; ++k) mx = mx > opt.mat[k] ? mx : opt.mat[k];
    const long long top = (long long)(tv.max_len + 1) * (mx > opt.a ? mx : opt.a) + 64;
    const char* e = getenv("BWAMEM_HIP_EXTEND_WAVE");
    return tv.max_len <= 250 && top < 16000 && opt.a > 0 && !(e && atoi(e) > 0);
}
size_t extend_rounds_bytes(int n_reads, int what)
{
    const size_t t = (size_t)n_reads;
    return what == 0 ? t * sizeof(ExtState) : what == 1 ? t * sizeof(ExtRes) : what == 2 ? t * sizeof(ExtTask) : what == 3 ? t * EXT_BINS * 4 : what == 4 ? 256 : t * 4;
}

hipError_t launch_extend_rounds(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, const ExtRoundBufs& B)
{
    if (tv.n_reads <= 0) return hipSuccess;
    ExtState* states = (ExtState*)B.states; ExtRes* res = (ExtRes*)B.res; ExtTask* tasks = (ExtTask*)B.tasks;
    int32_t* bin_cnt = B.counters;                         // [EXT_BINS] bin counts, [EXT_BINS] left-over count
    int32_t* left_cnt = B.counters + EXT_BINS;
    const dim3 cgrid((tv.n_reads + 127) / 128), cblock(128);
    int32_t h[EXT_BINS + 1];
    hipError_t err;
    bool pending = true;
    int n_rounds = EXT_ROUNDS;
    { const char* e = getenv("BWAMEM_HIP_EXT_ROUNDS"); if (e && atoi(e) > 0) n_rounds = atoi(e); }
    for (int round = 0; round < n_rounds && pending; ++round) {
        if ((err = hipMemsetAsync(bin_cnt, 0, (EXT_BINS + 1) * 4, st)) != hipSuccess) return err;
        hipLaunchKernelGGL(k_ext_ctrl, cgrid, cblock, 0, st, ix, opt, tv, states, res, tasks, B.bin_list, bin_cnt, round == 0 ? 1 : 0, 0, B.left_list, left_cnt);
        if ((err = hipMemcpyAsync(h, bin_cnt, sizeof h, hipMemcpyDeviceToHost, st)) != hipSuccess) return err;
        if ((err = hipStreamSynchronize(st)) != hipSuccess) return err;
        BinStart bs; bs.s[0] = 0;
        for (int b = 0; b < EXT_BINS; ++b) bs.s[b + 1] = bs.s[b] + h[b];
        pending = bs.s[EXT_BINS] > 0;
        static const int edges[4] = { 0, 6, 12, EXT_BINS };   // launches by query length: < 48, < 96, the rest (LDS per wave follows)
        for (int g = 0; g < 3; ++g) {
            const int n = bs.s[edges[g + 1]] - bs.s[edges[g]];
            if (n <= 0) continue;
            int max_q = edges[g + 1] * 8 - 1;
            if (g == 2 || max_q > tv.max_len) max_q = tv.max_len;
            const size_t lds = (size_t)(max_q + 2) * 64 * sizeof(uint32_t);
            hipLaunchKernelGGL(k_ext_dp, dim3((n + 63) / 64), dim3(64), lds, st, ix, opt, tv, tasks, res, B.bin_list, bs, edges[g], edges[g + 1]);
        }
    }
    if (pending) {                                          // reads that need more than EXT_ROUNDS extensions
        if ((err = hipMemsetAsync(bin_cnt, 0, (EXT_BINS + 1) * 4, st)) != hipSuccess) return err;
        hipLaunchKernelGGL(k_ext_ctrl, cgrid, cblock, 0, st, ix, opt, tv, states, res, tasks, B.bin_list, bin_cnt, 0, 1, B.left_list, left_cnt);
        if ((err = hipMemcpyAsync(h, bin_cnt, sizeof h, hipMemcpyDeviceToHost, st)) != hipSuccess) return err;
        if ((err = hipStreamSynchronize(st)) != hipSuccess) return err;
        const int n_left = h[EXT_BINS];
        if (n_left > 0) {
            size_t cap = (size_t)tv.max_len + 2;
            size_t shmem = 3 * cap * sizeof(int32_t) + ((cap + 15) & ~(size_t)15);
            hipLaunchKernelGGL(k_extend, dim3(n_left), dim3(64), shmem, st, ix, opt, tv, (const int32_t*)B.left_list);
        }
    }
    return hipGetLastError();
}
