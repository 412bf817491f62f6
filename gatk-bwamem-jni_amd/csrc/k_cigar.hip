// k_cigar.hip -- banded global alignment with traceback, one wavefront per region.
//
// Replaces, for the reference call at jnibwa.c:214, upstream ksw.c ksw_global2 as driven by the retry loop of
// bwamem.c mem_reg2aln / bwa.c bwa_gen_cigar2 for the single-end path.  Only a few per cent of
// short reads need this DP (those with indels), but when it ran inside the one-lane-per-read record kernel nearly
// every wave contained such a lane and waited for its latency-bound scalar DP.  Regions that need DP are therefore
// compacted into a job list by k_final_prep and solved here with the band across the lanes: within a row
// M(i,j) and E(i,j) depend on row i-1 only, and F(i,j) is a max-plus prefix of M(i,.), i.e. one wave shuffle scan
// (the same structure as k_extend).  Direction bits, tie rules and -inf arithmetic are upstream's, so the
// traceback (done by lane 0 from LDS) yields the identical CIGAR.
#include "dev_common.h"
#include "wave_ops.h"
#include "kernels.h"
#include "post_common.h"
#include "pk16.h"

#define NEG_SCAN (-2000000000)

// BWAMEM_HIP_DEBUGK bit 0x2000: shader clocks of the wave form's phases, summed over wavefronts into DevCounters::dbg[5..7]
// (DP rows, staging of traceback tiles, lane 0's walk); clk == 0: no clock is read
struct GClk { unsigned long long* c; long long t; };
DEV void gclk(GClk& K, int k) { if (K.c) { const long long t = clock64(); if (threadIdx.x == 0) atomicAdd(K.c + k, (unsigned long long)(t - K.t)); K.t = clock64(); } }

struct GLds { int32_t* eh_h; int32_t* eh_e; int32_t* tmpM; int rm; };   // rm: ring mask of the rows (only columns i - w .. i + w + 1 are live: see ExtLds in k_extend.hip); ~0 for rows in global memory

// backtrack through the direction bytes (lane 0); ops come out end-to-start and are reversed in place.  The run being
// built stays in registers: the CIGAR pool is global memory, and a read-modify-write per step would put a global
// round trip on every one of the ~tlen steps.
DEV void z_put(uint8_t* p, uint8_t v, bool z_lds) { if (z_lds) *AS_LDS(uint8_t, p) = v; else *AS_GLOBAL(uint8_t, p) = v; }
DEV int z_get(const uint8_t* p, bool z_lds) { return z_lds ? *AS_LDS(const uint8_t, p) : *AS_GLOBAL(const uint8_t, p); }

static __device__ __forceinline__ int traceback(const uint8_t* z, bool z_lds, int n_col, int w, int tlen, int qlen, int lane, uint32_t* cigar, int cig_cap, int& err, uint8_t* tile, GClk& K)
{
    gclk(K, 5);
    int n = 0;
    int which = 0, i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
    int cur_op = -1;
    uint32_t cur_len = 0;
    bool ovf = false;
#define TB_PUSH(OP, LEN) do { if ((OP) == cur_op) cur_len += (uint32_t)(LEN); else { \
        if (cur_len) { if (n >= cig_cap) ovf = true; else cigar[n++] = cur_len << 4 | (uint32_t)cur_op; } cur_op = (OP); cur_len = (uint32_t)(LEN); } } while (0)
    if (!tile) {                                                  // the matrix is in LDS: lane 0 walks it directly
        if (lane == 0)
            while (i >= 0 && k >= 0 && !ovf) {
                which = z_get(z + ((int64_t)i * n_col + (k - (i > w ? i - w : 0))), z_lds) >> (which << 1) & 3;
                if (which == 0) { TB_PUSH(0, 1); --i; --k; }
                else if (which == 1) { TB_PUSH(2, 1); --i; }
                else { TB_PUSH(1, 1); --k; }
            }
    } else {
        // The matrix is in global memory (long reads: megabytes per job), and a walk of one dependent global load per step
        // costs a microsecond a step.  The path moves up and/or left by one cell per step, so the 64 x 64 cells up-left of
        // the current one hold its next 64 steps at least: the wave stages that tile in LDS (64 coalesced byte loads in
        // flight), lane 0 walks inside it, and so on.
        for (;;) {
            const int i0 = wave_bcast(i, 0), k0 = wave_bcast(k, 0);
            if (i0 < 0 || k0 < 0 || wave_bcast(ovf ? 1 : 0, 0)) break;
            __syncthreads();
            // (every load unconditional -- cells outside the matrix read z[0] and are zeroed afterwards -- and sixteen rows per
            // batch, so a batch is sixteen loads in flight, not sixteen round trips: with one conditional load per iteration
            // this loop alone was 16 of the 23 ms of a 10 kb job)
            for (int t0 = 0; t0 < 64; t0 += 16) {
                uint8_t bt[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int r = i0 - (t0 + u), c = k0 - 63 + lane;
                    const int rb = r > w ? r - w : 0, re = r + w + 1 < qlen ? r + w + 1 : qlen;
                    const bool in = r >= 0 && c >= rb && c < re;
                    const uint8_t v = *AS_GLOBAL(const uint8_t, z + (in ? (int64_t)r * n_col + (c - rb) : 0));
                    bt[u] = in ? v : (uint8_t)0;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) *AS_LDS(uint8_t, tile + ((t0 + u) * 64 + lane)) = bt[u];
            }
            __syncthreads();
            gclk(K, 6);
            if (lane == 0)
                while (i >= 0 && k >= 0 && !ovf && i > i0 - 64 && k > k0 - 64) {
                    which = *AS_LDS(const uint8_t, tile + ((i0 - i) * 64 + (k - (k0 - 63)))) >> (which << 1) & 3;
                    if (which == 0) { TB_PUSH(0, 1); --i; --k; }
                    else if (which == 1) { TB_PUSH(2, 1); --i; }
                    else { TB_PUSH(1, 1); --k; }
                }
            gclk(K, 7);
        }
    }
    if (lane == 0) {
        if (!ovf && i >= 0) TB_PUSH(2, i + 1);
        if (!ovf && k >= 0) TB_PUSH(1, k + 1);
        if (!ovf) TB_PUSH(-1, 0);                                 // flush the last run
        if (ovf) { err |= ERR_CIGAR_CAP; n = 0; }
    }
#undef TB_PUSH
    n = wave_bcast(n, 0);
    // the reversal across the lanes: the pool is global memory, and one lane swapping the ~3 000 operations of a noisy 10 kb
    // read pair by pair is ~1 500 dependent round trips (a third of such a job's time when it was written that way)
    __syncthreads();
    for (int a = lane; a < n >> 1; a += WAVE) { const uint32_t lo = load_fresh(cigar + a), hi = load_fresh(cigar + (n - 1 - a)); cigar[a] = hi; cigar[n - 1 - a] = lo; }
    __syncthreads();
    gclk(K, 8);
    return n;
}

// one ksw_global2 call; the raw CIGAR (before clip / deletion squeezing) goes to cigar[0..*n_cigar)
static __device__ __forceinline__ int global_wave(const DevIndex& ix, const MemOpt& opt, const GLds& L, int lane, const SeqAcc& A, int w,
                                  uint8_t* z, bool z_lds, int n_col)
{
    const int RM = L.rm;
    const int qlen = A.qlen, tlen = A.tlen;
    const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const ScoreTab ST = score_tab(opt);
    for (int j = lane; j <= qlen && j <= w + 1; j += WAVE) {     // (row i writes index i + w + 1 itself before row i + 1 reads it)
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
        for (int c = beg; c < end; c += WAVE) {                     // phase A: M(i,j) from row i-1
            int j = c + lane;
            if (j < end) {
                int qc = acc_q(A, j);
                int sc = qc == 0 ? ms0 : qc == 1 ? ms1 : qc == 2 ? ms2 : qc == 3 ? ms3 : ms4;
                L.tmpM[j & RM] = L.eh_h[j & RM] + sc;
            }
        }
        __syncthreads();
        int fcarry = MINUS_INF, hlast = h1i;
        uint8_t* zi = z + (int64_t)i * n_col;
        for (int c = beg; c < end; c += WAVE) {                     // phase B: F by scan, H, E, direction bits
            int j = c + lane;
            bool act = j < end;
            int m = act ? L.tmpM[j & RM] : 0;
            int e = act ? L.eh_e[j & RM] : 0;
            int tins = m - oe_ins;
            int U = act ? tins + j * e_ins : NEG_SCAN;
            int P = wave_prefix_max(U, lane);
            int Pex = __shfl_up(P, 1);
            int f = fcarry - (j - c) * e_ins;
            if (lane > 0) { int g = Pex - (j - 1) * e_ins; f = f > g ? f : g; }
            int d = m >= e ? 0 : 1;
            int h = m >= e ? m : e;
            d = h >= f ? d : 2;
            h = h >= f ? h : f;
            int t = m - oe_del;
            int e2 = e - e_del;
            d |= e2 > t ? 1 << 2 : 0;
            e2 = e2 > t ? e2 : t;
            d |= (f - e_ins) > tins ? 2 << 4 : 0;
            int last = (end - 1 - c) < 63 ? (end - 1 - c) : 63;
            hlast = wave_bcast(h, last);
            int Plast = wave_bcast(P, 63);
            { int f1 = fcarry - WAVE * e_ins, f2 = Plast - (c + 63) * e_ins; fcarry = f1 > f2 ? f1 : f2; }
            if (act) { z_put(zi + (j - beg), (uint8_t)d, z_lds); L.eh_e[j & RM] = e2; L.eh_h[(j + 1) & RM] = h; }
        }
        if (lane == 0) {
            if (end > beg) L.eh_h[beg & RM] = h1i;
            else L.eh_h[end & RM] = h1i;
            L.eh_e[end & RM] = MINUS_INF;
        }
        (void)hlast;
        __syncthreads();
    }
    const int score = L.eh_h[qlen & RM];
    __syncthreads();
    return score;
}

// The same DP for bands of at most 64 columns (2w + 1 <= 64: every short-read job), with the band across the lanes:
// lane l owns column j = i - w + l of row i, i.e. one diagonal.  H(i-1,j-1) is then the lane's own value of the previous
// row, E(i,j) comes from the lane above (one DPP shift), F is the max-plus prefix over the lanes of the row (DPP scan),
// and the query slides down the lanes one position per row.  Nothing of the DP state lives in LDS and no barrier is
// needed; only the direction bytes are stored (z, in upstream's [row][column - beg] layout) for the traceback.
static __device__ __forceinline__ int global_wave_diag(const DevIndex& ix, const MemOpt& opt, const uint8_t* sq, int lane, const SeqAcc& A, int w,
                                       uint8_t* z, bool z_lds, int n_col)
{
    const int qlen = A.qlen, tlen = A.tlen;
    const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const ScoreTab ST = score_tab(opt);
    const int j0 = lane - w;
    int hd = j0 == 0 ? 0 : (j0 > 0 && j0 <= w ? -(o_ins + e_ins * j0) : MINUS_INF);   // H(-1, j-1): upstream's initial eh[j].h
    int e = MINUS_INF;
    int qv = j0 >= 0 && j0 < qlen ? sq[j0] : 4;
    int tch = 4, h = MINUS_INF;
    for (int i = 0; i < tlen; ++i) {
        if ((i & 63) == 0) { int ii = i + lane; tch = ii < tlen ? acc_t(ix, A, ii) : 4; }
        const int tb = wave_readlane(tch, i & 63);
        const int ms0 = score_at(ST.p[0], ST.n[0], tb), ms1 = score_at(ST.p[1], ST.n[1], tb), ms2 = score_at(ST.p[2], ST.n[2], tb), ms3 = score_at(ST.p[3], ST.n[3], tb), ms4 = score_at(ST.p[4], ST.n[4], tb);
        const int inj = i + 64 - w;                                  // query position entering lane 63 for the next row
        const int qin = inj >= 0 && inj < qlen ? sq[inj] : 4;
        const int j = i - w + lane;
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        const int lb = beg - (i - w);                                // first lane of the band in this row
        const bool act = j >= beg && j < end;
        const int sc = qv == 0 ? ms0 : qv == 1 ? ms1 : qv == 2 ? ms2 : qv == 3 ? ms3 : ms4;
        const int m = hd + sc;
        const int tins = m - oe_ins;
        const int U = act ? tins + lane * e_ins : NEG_SCAN;
        const int P = dpp_prefix_max(U, NEG_SCAN);
        const int Pex = dpp_shr1(P, NEG_SCAN);
        int f = MINUS_INF - (lane - lb) * e_ins;
        { const int g = Pex - (lane - 1) * e_ins; f = f > g ? f : g; }
        int d = m >= e ? 0 : 1;
        h = m >= e ? m : e;
        d = h >= f ? d : 2;
        h = h >= f ? h : f;
        const int t = m - oe_del;
        int e2 = e - e_del;
        d |= e2 > t ? 1 << 2 : 0;
        e2 = e2 > t ? e2 : t;
        d |= (f - e_ins) > tins ? 2 << 4 : 0;
        if (act) z_put(z + ((int64_t)i * n_col + (lane - lb)), (uint8_t)d, z_lds);
        // state of the next row: the lane moves one column to the right along its diagonal
        hd = act ? h : (j == -1 ? -(o_del + e_del * (i + 1)) : MINUS_INF);
        e = dpp_shl1(act ? e2 : MINUS_INF, MINUS_INF);
        qv = dpp_shl1(qv, qin);
    }
    const int score = wave_bcast(h, qlen - 1 - (tlen - 1 - w));     // H(tlen-1, qlen-1); the caller checked that lane is in the band
    return score;
}

// The diagonal form for bands of up to 64 NCH columns (long reads: 2w + 1 runs to several hundred): slot s = 64 c + lane of
// chunk c owns column i - w + s of row i.  Same recurrence, tie rules and direction bytes as above; the max-plus prefix
// carries from chunk to chunk through a scalar, the E and query shifts take the first lane of the next chunk as their fill.
// No LDS rows, no barriers: one DPP scan per chunk and row.
template <int NCH>
static __device__ __forceinline__ int global_wave_diag_n(const DevIndex& ix, const MemOpt& opt, const uint8_t* sq, int lane, const SeqAcc& A, int w,
                                         uint8_t* z, bool z_lds, int n_col)
{
    const int qlen = A.qlen, tlen = A.tlen;
    const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const ScoreTab ST = score_tab(opt);
    int hd[NCH], e[NCH], qv[NCH], hrow[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int j0 = c * 64 + lane - w;
        hd[c] = j0 == 0 ? 0 : (j0 > 0 && j0 <= w ? -(o_ins + e_ins * j0) : MINUS_INF);   // H(-1, j-1): upstream's initial eh[j].h
        e[c] = MINUS_INF; hrow[c] = MINUS_INF;
        qv[c] = j0 >= 0 && j0 < qlen ? sq[j0] : 4;
    }
    int tch = 4;
    for (int i = 0; i < tlen; ++i) {
        if ((i & 63) == 0) { int ii = i + lane; tch = ii < tlen ? acc_t(ix, A, ii) : 4; }
        const int tb = wave_readlane(tch, i & 63);
        const int ms0 = score_at(ST.p[0], ST.n[0], tb), ms1 = score_at(ST.p[1], ST.n[1], tb), ms2 = score_at(ST.p[2], ST.n[2], tb), ms3 = score_at(ST.p[3], ST.n[3], tb), ms4 = score_at(ST.p[4], ST.n[4], tb);
        const int inj = i + 64 * NCH - w;                            // query position entering the last slot for the next row
        const int qin = inj >= 0 && inj < qlen ? sq[inj] : 4;
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        const int lb = beg - (i - w);                                // first slot of the band in this row
        uint8_t* zi = z + (int64_t)i * n_col;
        int pall = NEG_SCAN;                                         // best tins + s e_ins over the chunks to the left
        int e2s[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int s = c * 64 + lane, j = i - w + s;
            const bool act = j >= beg && j < end;
            const int q = qv[c];
            const int sc = q == 0 ? ms0 : q == 1 ? ms1 : q == 2 ? ms2 : q == 3 ? ms3 : ms4;
            const int m = hd[c] + sc;
            const int tins = m - oe_ins;
            const int U = act ? tins + s * e_ins : NEG_SCAN;
            const int P = dpp_prefix_max(U, NEG_SCAN);
            int Pex = dpp_shr1(P, NEG_SCAN);
            Pex = Pex > pall ? Pex : pall;
            int f = MINUS_INF - (s - lb) * e_ins;
            { const int g = Pex - (s - 1) * e_ins; f = f > g ? f : g; }
            int d = m >= e[c] ? 0 : 1;
            int h = m >= e[c] ? m : e[c];
            d = h >= f ? d : 2;
            h = h >= f ? h : f;
            const int t = m - oe_del;
            int e2 = e[c] - e_del;
            d |= e2 > t ? 1 << 2 : 0;
            e2 = e2 > t ? e2 : t;
            d |= (f - e_ins) > tins ? 2 << 4 : 0;
            if (act) z_put(zi + (s - lb), (uint8_t)d, z_lds);
            hd[c] = act ? h : (j == -1 ? -(o_del + e_del * (i + 1)) : MINUS_INF);   // the lane moves one column to the right along its diagonal
            hrow[c] = h;
            e2s[c] = act ? e2 : MINUS_INF;
            { const int tot = wave_readlane(P, 63); pall = pall > tot ? pall : tot; }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int efill = c + 1 < NCH ? wave_readlane(e2s[c + 1 < NCH ? c + 1 : c], 0) : MINUS_INF;
            const int qfill = c + 1 < NCH ? wave_readlane(qv[c + 1 < NCH ? c + 1 : c], 0) : qin;
            e[c] = dpp_shl1(e2s[c], efill);
            qv[c] = dpp_shl1(qv[c], qfill);
        }
    }
    const int l_end = qlen - 1 - (tlen - 1 - w);                     // H(tlen-1, qlen-1); the caller checked that slot is in the band
    int score = MINUS_INF;
#pragma unroll
    for (int c = 0; c < NCH; ++c) if ((l_end >> 6) == c) score = wave_bcast(hrow[c], l_end & 63);
    return score;
}

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
template <int NP>
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
            const int qv = j0 >= 0 && j0 < qlen ? sq[j0] : 4;
            h2 |= (uint32_t)(uint16_t)hv << (hf << 4);
            q2 |= (uint32_t)(qv | 0x0c00) << (hf << 4);
        }
        hd[p] = h2; qs[p] = q2; e[p] = SENT2; hrow[p] = SENT2; am[p] = 0;
        SE[p] = pk_pair((p * 64 + lane) * e_ins, ((p + NP) * 64 + lane) * e_ins);
    }
    int base = 0, tch = 4, lb_prev = -1, le_prev = -1;
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
        const int qin = inj >= 0 && inj < qlen ? sq[inj] : 4;
        uint8_t* zi = z + (int64_t)i * n_col;
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
            if (am[p] & 0xffffu) z_put(zi + (p * 64 + lane - lb), (uint8_t)d, z_lds);
            if (am[p] >> 16) z_put(zi + ((p + NP) * 64 + lane - lb), (uint8_t)(d >> 16), z_lds);
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

// Lane-per-job form for narrow bands (w <= WMAX: the regions with three or four mismatches, i.e. most jobs).  The wave
// forms above spend ~100 vector instructions per row on a band that fills a third of the lanes.  Here every lane runs
// the scalar recurrence for its own job with the whole band in registers: slot l of a lane holds column i - w + l of the
// current row (one diagonal per slot, as in global_wave_diag), so H(i-1,j-1) is the slot's own value, E(i,j) the value
// left by the next slot in the previous row, and F runs along the unrolled slots.  64 jobs per wavefront, ~30 vector
// instructions per 64 cells.  Direction bits go to the HBM pool as one nibble per slot (20 bytes per row) and the
// traceback of each lane reads them back.  Jobs it cannot take (wider band, retry with a doubled band needed) are left
// marked for k_gcigar.
struct GRowTab { uint32_t p[5]; int n[5]; };             // per target base: scores against query bases 0..3 (bytes), and against N
DEV GRowTab g_row_tab(const MemOpt& opt)
{
    GRowTab T;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        T.p[t] = (uint32_t)(uint8_t)opt.mat[t * 5] | (uint32_t)(uint8_t)opt.mat[t * 5 + 1] << 8 | (uint32_t)(uint8_t)opt.mat[t * 5 + 2] << 16 | (uint32_t)(uint8_t)opt.mat[t * 5 + 3] << 24;
        T.n[t] = opt.mat[t * 5 + 4];
    }
    return T;
}

template <int WMAX>
__global__ void __launch_bounds__(64, 2) k_gcigar_lane(DevIndex ix, MemOpt opt, TileView tv, const DpJob* jobs, DpOut* outs, int n_jobs, uint32_t* cig_pool, int cig_cap,
                                                    uint8_t* zpool, unsigned long long zpool_cap, unsigned long long* zpool_cur)
{
    constexpr int NS = 2 * WMAX + 1;                      // band slots
    constexpr int ZW = (NS + 7) / 8;                      // dwords of direction nibbles per row
    const int job = blockIdx.x * 64 + threadIdx.x;
    if (job >= n_jobs) return;
    const DpJob jb = jobs[job];
    const AlnReg ar = tv.regs[tv.seed_off[jb.read] + jb.reg];
    const uint8_t* query = tv.seq + tv.seq_off[jb.read];
    SeqAcc A; A.q = query + ar.qb; A.qlen = ar.qe - ar.qb; A.rev = ar.rb >= ix.l_pac; A.t0 = ar.rb; A.tlen = (int)(ar.re - ar.rb);
    const int qlen = A.qlen, tlen = A.tlen;
    DpOut o; o.score = 0; o.n_cigar = -1;                 // -1: left for k_gcigar
    const bool usable = !(qlen <= 0 || ar.rb >= ar.re || (ar.rb < ix.l_pac && ar.re > ix.l_pac) || ar.re > ix.l_pac << 1 || ar.rb < 0);
    int w2 = first_w2(opt, ar), w = 0;
    if (usable) {                                         // band of the first bwa_gen_cigar2 call
        w2 = w2 < opt.w << 2 ? w2 : opt.w << 2;
        int max_ins = div_plus(((qlen + 1) >> 1) * opt.mat[0] - opt.o_ins, opt.e_ins, 1);
        int max_del = div_plus(((qlen + 1) >> 1) * opt.mat[0] - opt.o_del, opt.e_del, 1);
        int max_gap = max_ins > max_del ? max_ins : max_del;
        max_gap = max_gap > 1 ? max_gap : 1;
        int d = tlen - qlen; d = d < 0 ? -d : d;
        w = (max_gap + d + 1) >> 1;
        w = w < w2 ? w : w2;
        const int min_w = d + 3;
        w = w > min_w ? w : min_w;
    }
    const int l_end = qlen - 1 - (tlen - 1 - w);           // slot of the final cell
    uint32_t* z = 0;
    bool take = usable && w <= WMAX && l_end >= 0 && l_end <= 2 * w;
    if (take) {
        const unsigned long long need = ((unsigned long long)tlen * ZW * 4 + 63ull) & ~63ull;
        const unsigned long long at = atomicAdd(zpool_cur, need);
        if (at + need > zpool_cap) { atomicOr(tv.err, ERR_ZPOOL); take = false; }
        else z = (uint32_t*)(zpool + at);
    }
    if (!take) { outs[job] = o; return; }

    const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const GRowTab RT = g_row_tab(opt);
    SeqCache sc; seq_cache_init(sc);
    int H[NS], E[NS], Q[NS];
#pragma unroll
    for (int l = 0; l < NS; ++l) {
        const int j0 = l - w;
        H[l] = j0 == 0 ? 0 : (j0 > 0 && j0 <= w ? -(o_ins + e_ins * j0) : MINUS_INF);   // upstream's initial eh[j].h
        E[l] = MINUS_INF;
        Q[l] = j0 >= 0 && j0 < qlen ? acc_q_c(A, sc, j0) : 4;
    }
    for (int i = 0; i < tlen; ++i) {
        const int tb = acc_t_c(ix, A, sc, i);
        const uint32_t R = tb == 0 ? RT.p[0] : tb == 1 ? RT.p[1] : tb == 2 ? RT.p[2] : tb == 3 ? RT.p[3] : RT.p[4];
        const int Rn = tb == 0 ? RT.n[0] : tb == 1 ? RT.n[1] : tb == 2 ? RT.n[2] : tb == 3 ? RT.n[3] : RT.n[4];
        const int c0 = i - w;                                // column of slot 0
        const int hb = -(o_del + e_del * (i + 1));           // H(i,-1), the diagonal predecessor of column 0 in the next row
        int f = MINUS_INF;
        uint32_t zw[ZW];
#pragma unroll
        for (int k = 0; k < ZW; ++k) zw[k] = 0;
#pragma unroll
        for (int l = 0; l < NS; ++l) {
            const int j = c0 + l;
            const bool act = j >= 0 && j < qlen && l <= 2 * w;
            const int q = Q[l];
            const int s = q < 4 ? (int)(int8_t)(R >> (q << 3)) : Rn;
            const int m = H[l] + s, e = E[l];
            int d = m >= e ? 0 : 1;
            int h = m >= e ? m : e;
            d = h >= f ? d : 2;
            h = h >= f ? h : f;
            const int t = m - oe_del;
            int e2 = e - e_del;
            d |= e2 > t ? 4 : 0;
            e2 = e2 > t ? e2 : t;
            const int t2 = m - oe_ins;
            int f2 = f - e_ins;
            d |= f2 > t2 ? 8 : 0;
            f2 = f2 > t2 ? f2 : t2;
            H[l] = act ? h : (j == -1 ? hb : MINUS_INF);
            if (l > 0) E[l - 1] = act ? e2 : MINUS_INF;      // E(i+1, j) is read by slot l-1 in the next row
            f = act ? f2 : f;
            zw[l >> 3] |= act ? (uint32_t)d << ((l & 7) << 2) : 0u;
        }
        E[NS - 1] = MINUS_INF;
#pragma unroll
        for (int k = 0; k < ZW; ++k) z[i * ZW + k] = zw[k];
        const int inj = i + 1 - w + NS - 1;                  // query position entering the last slot
        const int qin = inj >= 0 && inj < qlen ? acc_q_c(A, sc, inj) : 4;
#pragma unroll
        for (int l = 0; l + 1 < NS; ++l) Q[l] = Q[l + 1];
        Q[NS - 1] = qin;
    }
    int score = MINUS_INF;
#pragma unroll
    for (int l = 0; l < NS; ++l) score = l == l_end ? H[l] : score;   // H(tlen-1, qlen-1)

    // is this the CIGAR mem_reg2aln keeps, or does its loop retry with a doubled band?
    if (!(w2 == opt.w << 2) && score < ar.truesc - opt.a) { outs[job] = o; return; }

    uint32_t* cigar = cig_pool + (size_t)job * cig_cap;
    int n = 0, cur_op = -1, which = 0, i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
    uint32_t cur_len = 0;
    bool ovf = false;
#define TB_PUSH(OP, LEN) do { if ((OP) == cur_op) cur_len += (uint32_t)(LEN); else { \
        if (cur_len) { if (n >= cig_cap) ovf = true; else cigar[n++] = cur_len << 4 | (uint32_t)cur_op; } cur_op = (OP); cur_len = (uint32_t)(LEN); } } while (0)
    while (i >= 0 && k >= 0 && !ovf) {
        const int l = k - (i - w);
        const uint32_t nib = z[i * ZW + (l >> 3)] >> ((l & 7) << 2) & 15u;
        which = which == 0 ? (int)(nib & 3u) : which == 1 ? (int)(nib >> 2 & 1u) : (int)(nib >> 3 & 1u) << 1;
        if (which == 0) { TB_PUSH(0, 1); --i; --k; }
        else if (which == 1) { TB_PUSH(2, 1); --i; }
        else { TB_PUSH(1, 1); --k; }
    }
    if (!ovf && i >= 0) TB_PUSH(2, i + 1);
    if (!ovf && k >= 0) TB_PUSH(1, k + 1);
    if (!ovf) TB_PUSH(-1, 0);
#undef TB_PUSH
    for (int a = 0; a < n >> 1; ++a) { uint32_t tmp = cigar[a]; cigar[a] = cigar[n - 1 - a]; cigar[n - 1 - a] = tmp; }
    if (ovf) { atomicOr(tv.err, ERR_CIGAR_CAP); n = 0; }
    o.score = score; o.n_cigar = n;
    outs[job] = o;
}

// HBM: as in k_extend -- rows of the general form in a slice of tv.dp_rows instead of LDS, a bounded grid walking the jobs
// HBM = false: rows as rings of `ring` entries in LDS; a job whose band needs more than that is left for the HBM kernel
// (launched after this one whenever the tile's reads are longer than the ring).
// MAXCH: the widest register form compiled in (chunks of 64 diagonals).  Tiles of short reads use 2: the 13-chunk form needs over a
// hundred registers, and the kernel's register count -- hence its occupancy -- is that of its hungriest path.
template <bool HBM, int MAXCH>
__global__ void __launch_bounds__(64, MAXCH >= 13 ? 2 : 4) k_gcigar(DevIndex ix, MemOpt opt, TileView tv, const DpJob* jobs, DpOut* outs, int n_jobs, uint32_t* cig_pool, int cig_cap,
                                               uint8_t* zpool, unsigned long long zpool_cap, unsigned long long* zpool_cur, int z_lds_cap, int ring,
                                               uint8_t* slabs, unsigned long long slab_bytes, int* queue)
{
    HIP_DYNAMIC_SHARED(int32_t, smem)
    const int lane = threadIdx.x;
    // slabs != null (tiles of long reads): a resident grid pulls jobs from a queue and every workgroup keeps the traceback
    // matrix of its current job in a slab of its own -- megabytes per job, but bounded by the grid, not by the job count
  for (int job = blockIdx.x, first = 1; ; first = 0) {
    if (slabs) { int nx = 0; if (lane == 0) nx = atomicAdd(queue, 1); job = __shfl(nx, 0); }
    else if (!first) job += HBM ? (int)gridDim.x : n_jobs;
    if (job >= n_jobs) break;
    __syncthreads();                                                   // the previous job of this workgroup is done with the rows
    if (outs[job].n_cigar >= 0) continue;                              // done by k_gcigar_lane
    const DpJob jb = jobs[job];
    const AlnReg ar = tv.regs[tv.seed_off[jb.read] + jb.reg];
    const uint8_t* query = tv.seq + tv.seq_off[jb.read];
    const int cap = tv.max_len + 2;
    GLds L;
    uint8_t* sq;                                                       // the region's query bases, in alignment order
    if (HBM) {
        int32_t* rows = tv.dp_rows + (size_t)blockIdx.x * 3 * (size_t)cap;
        L.eh_h = rows; L.eh_e = rows + cap; L.tmpM = rows + 2 * cap; L.rm = 0x7fffffff;
        sq = (uint8_t*)smem;
    } else {
        L.eh_h = smem; L.eh_e = smem + ring; L.tmpM = smem + 2 * ring; L.rm = ring - 1;
        sq = (uint8_t*)(smem + 3 * ring);
    }
    uint8_t* z_lds = sq + ((cap + 15) & ~15);
    int err = 0;
    const bool use_pk = !(tv.debug & 0x10000);                         // (BWAMEM_HIP_DEBUGK bit 0x10000: the 32-bit forms only)
    GClk K; K.c = (tv.debug & 0x2000) ? tv.cnt->dbg : nullptr; K.t = K.c ? clock64() : 0;
    SeqAcc A; A.q = query + ar.qb; A.qlen = ar.qe - ar.qb; A.rev = ar.rb >= ix.l_pac; A.t0 = ar.rb; A.tlen = (int)(ar.re - ar.rb);
    uint32_t* cigar = cig_pool + (size_t)job * cig_cap;
    // the retry loop of mem_reg2aln around bwa_gen_cigar2 (all lanes take the same decisions)
    int w2 = first_w2(opt, ar), i = 0, last_sc = -(1 << 30), score = 0, n_cigar = 0;
    const int l_query = A.qlen, rlen = A.tlen;
    const bool usable = !(l_query <= 0 || ar.rb >= ar.re || (ar.rb < ix.l_pac && ar.re > ix.l_pac) || ar.re > ix.l_pac << 1 || ar.rb < 0);
    bool deferred = false;                                             // band too wide for the LDS rings: the HBM kernel takes the job
    if (usable) {
        for (int j = lane; j < l_query; j += WAVE) sq[j] = (uint8_t)acc_q(A, j);
        __syncthreads();
        do {
            w2 = w2 < opt.w << 2 ? w2 : opt.w << 2;
            int w, max_gap, max_ins, max_del, min_w, d;                // band of bwa_gen_cigar2
            max_ins = div_plus(((l_query + 1) >> 1) * opt.mat[0] - opt.o_ins, opt.e_ins, 1);
            max_del = div_plus(((l_query + 1) >> 1) * opt.mat[0] - opt.o_del, opt.e_del, 1);
            max_gap = max_ins > max_del ? max_ins : max_del;
            max_gap = max_gap > 1 ? max_gap : 1;
            d = rlen - l_query; d = d < 0 ? -d : d;
            w = (max_gap + d + 1) >> 1;
            w = w < w2 ? w : w2;
            min_w = d + 3;
            w = w > min_w ? w : min_w;
            const int n_col = l_query < 2 * w + 1 ? l_query : 2 * w + 1;
            const unsigned long long need = (unsigned long long)n_col * (unsigned long long)rlen;
            uint8_t* z = z_lds;
            if (need > (unsigned long long)z_lds_cap && slabs && need <= slab_bytes) z = slabs + (unsigned long long)blockIdx.x * slab_bytes;
            else if (need > (unsigned long long)z_lds_cap) {             // traceback matrix too big for LDS: bump-allocate HBM
                unsigned long long at = 0;
                if (lane == 0) at = atomicAdd(zpool_cur, (need + 63ull) & ~63ull);
                at = __shfl(at, 0);
                if (at + need > zpool_cap) { err |= ERR_ZPOOL; break; }
                z = zpool + at;
            }
            const int l_end = l_query - 1 - (rlen - 1 - w);            // lane of the final cell in the diagonal form
            uint8_t* tile = z == z_lds || z_lds_cap < 4096 ? nullptr : z_lds;   // a matrix in global memory is walked through 64 x 64 tiles staged where the small ones live
            const bool diag_ok = l_end >= 0 && l_end <= 2 * w;
            const int nch = (2 * w + 1 + 63) >> 6;
            if (!HBM && !(diag_ok && nch <= MAXCH) && 2 * w + 4 > ring && l_query + 2 > ring) { deferred = true; break; }   // needs rows longer than the rings
            bool pk_done = false;                                      // two chunks per instruction stream where 16 bits hold the values (see global_wave_diag_pk)
            GpkFit fit;
            if (use_pk && diag_ok && nch >= 2 && nch <= (MAXCH >= 13 ? 14 : 2) && gpk_fit(opt, w, (nch + 1) >> 1, fit)) {
                if (nch <= 2) score = global_wave_diag_pk<1>(ix, opt, sq, lane, A, w, fit, z, z == z_lds, n_col, pk_done);
                else if (MAXCH >= 13 && nch <= 4) score = global_wave_diag_pk<2>(ix, opt, sq, lane, A, w, fit, z, z == z_lds, n_col, pk_done);
                else if (MAXCH >= 13 && nch <= 8) score = global_wave_diag_pk<4>(ix, opt, sq, lane, A, w, fit, z, z == z_lds, n_col, pk_done);
                else if (MAXCH >= 13) score = global_wave_diag_pk<7>(ix, opt, sq, lane, A, w, fit, z, z == z_lds, n_col, pk_done);
            }
            if (pk_done) { }
            else if (diag_ok && nch <= 1) score = global_wave_diag(ix, opt, sq, lane, A, w, z, z == z_lds, n_col);
            else if (MAXCH >= 2 && diag_ok && nch <= 2) score = global_wave_diag_n<2>(ix, opt, sq, lane, A, w, z, z == z_lds, n_col);
            else if (MAXCH >= 4 && diag_ok && nch <= 4) score = global_wave_diag_n<4>(ix, opt, sq, lane, A, w, z, z == z_lds, n_col);
            else if (MAXCH >= 7 && diag_ok && nch <= 7) score = global_wave_diag_n<7>(ix, opt, sq, lane, A, w, z, z == z_lds, n_col);
            else if (MAXCH >= 13 && diag_ok && nch <= 13) score = global_wave_diag_n<13>(ix, opt, sq, lane, A, w, z, z == z_lds, n_col);
            else score = global_wave(ix, opt, L, lane, A, w, z, z == z_lds, n_col);
            n_cigar = traceback(z, z == z_lds, n_col, w, rlen, l_query, lane, cigar, cig_cap, err, tile, K);
            __syncthreads();
            if (score == last_sc || w2 == opt.w << 2) break;
            last_sc = score;
            w2 <<= 1;
        } while (++i < 3 && score < ar.truesc - opt.a);
    }
    if (lane == 0 && !deferred) {
        DpOut o; o.score = score; o.n_cigar = n_cigar;
        outs[job] = o;
        if (err) atomicOr(tv.err, err);
    }
  }
}

// bytes of one traceback slab of the wave form for tiles of long reads (0: none -- short reads keep theirs in LDS / the pool)
size_t gcigar_slab_bytes(const MemOpt& opt, int max_len)
{
    if (max_len <= 1000) return 0;
    const size_t w_max = 4 * (size_t)(opt.w > 0 ? opt.w : 0) + 3;
    return ((2 * w_max + 1) * ((size_t)max_len + 2 * w_max + 64) + 255) & ~(size_t)255;
}
int gcigar_slab_grid(const DevIndex& ix, int n_jobs) { const int g = (ix.n_cu > 0 ? ix.n_cu : 256) * 8; return n_jobs < g ? n_jobs : g; }

void launch_gcigar(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, int n_jobs, const void* jobs, void* outs, uint32_t* cig_pool, int cig_cap,
                   uint8_t* zpool, unsigned long long zpool_cap, unsigned long long* zpool_cur, uint8_t* slabs, size_t slab_bytes, int* queue)
{
    if (n_jobs <= 0) return;
    int z_lds_cap = 6144;                                            // covers bands of ~40 columns x 150 rows; larger matrices go to the HBM pool
    { const char* e = getenv("BWAMEM_HIP_ZLDS"); if (e && atoi(e) >= 0) z_lds_cap = atoi(e); }
    size_t cap = (size_t)tv.max_len + 2;
    // rows of the wave form: rings that hold the widest band mem_reg2aln asks for (opt.w << 2) -- or the whole read when that
    // is shorter.  Jobs whose band is wider still (regions with a long net indel) go to the HBM kernel behind it.
    int ring = 64;
    {   // the widest band bwa_gen_cigar2 can take here is 4 w + 3 (w2 <= opt.w << 2, and regions are only merged across gaps of up to 4 w)
        long long need = 8ll * (opt.w > 0 ? opt.w : 0) + 16;
        if (need > (long long)cap) need = (long long)cap;            // (rows as long as the read never wrap)
        if (need > 4096) need = 4096;                                // 48 KB of rows at most
        { const char* e = getenv("BWAMEM_HIP_GCIGAR_RING"); if (e && atoi(e) > 0 && atoi(e) < need) need = atoi(e); }   // (tests: force jobs over to the HBM kernel)
        if (slab_bytes) need = 64;                                   // long reads: the register forms cover bands of up to 832 columns; the rest goes to the HBM kernel
        while (ring < need) ring <<= 1;
    }
    const size_t tail = ((cap + 15) & ~(size_t)15) + (size_t)z_lds_cap + 64;
    hipLaunchKernelGGL(k_gcigar_lane<16>, dim3((n_jobs + 63) / 64), dim3(64), 0, st, ix, opt, tv, (const DpJob*)jobs, (DpOut*)outs, n_jobs, cig_pool, cig_cap, zpool, zpool_cap, zpool_cur);
    if (!tv.gcigar_hbm_only) {
        if (!slab_bytes) slabs = nullptr;
        const int grid = slabs ? gcigar_slab_grid(ix, n_jobs) : n_jobs;
        if (slabs) hipLaunchKernelGGL((k_gcigar<false, 13>), dim3(grid), dim3(64), 3 * (size_t)ring * sizeof(int32_t) + tail, st, ix, opt, tv, (const DpJob*)jobs, (DpOut*)outs, n_jobs, cig_pool, cig_cap, zpool, zpool_cap, zpool_cur, z_lds_cap, ring,
                                      slabs, (unsigned long long)slab_bytes, queue);
        else hipLaunchKernelGGL((k_gcigar<false, 2>), dim3(grid), dim3(64), 3 * (size_t)ring * sizeof(int32_t) + tail, st, ix, opt, tv, (const DpJob*)jobs, (DpOut*)outs, n_jobs, cig_pool, cig_cap, zpool, zpool_cap, zpool_cur, z_lds_cap, ring,
                                (uint8_t*)nullptr, 0ull, queue);
    }
    if (tv.dp_rows) {
        const int grid = n_jobs < tv.dp_rows_blocks ? n_jobs : tv.dp_rows_blocks;
        hipLaunchKernelGGL((k_gcigar<true, 13>), dim3(grid), dim3(64), tail, st, ix, opt, tv, (const DpJob*)jobs, (DpOut*)outs, n_jobs, cig_pool, cig_cap, zpool, zpool_cap, zpool_cur, z_lds_cap, 0, (uint8_t*)nullptr, 0ull, (int*)nullptr);
    }
}
