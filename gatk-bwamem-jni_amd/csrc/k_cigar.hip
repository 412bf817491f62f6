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

#define NEG_SCAN (-2000000000)

// BWAMEM_HIP_DEBUGK bit 0x2000: shader clocks of the wave form's phases, summed over wavefronts into DevCounters::dbg[5..7]
// (DP rows, staging of traceback tiles, lane 0's walk); clk == 0: no clock is read
struct GClk { unsigned long long* c; long long t; };
DEV void gclk(GClk& K, int k) { if (K.c) { const long long t = clock64(); if (threadIdx.x == 0) atomicAdd(K.c + k, (unsigned long long)(t - K.t)); K.t = clock64(); } }

struct GLds { int32_t* eh_h; int32_t* eh_e; int32_t* tmpM; int rm; };   // rm: ring mask of the rows (only columns i - w .. i + w + 1 are live: see ExtLds in k_extend.hip); ~0 for rows in global memory

// backtrack through the direction bytes; ops come out end-to-start and are reversed in place.  The run being built stays in
// registers: the CIGAR pool is global memory, and a read-modify-write per step would put a global round trip on every one of
// the ~tlen steps.  The walk is one dependent load per step, but most steps are matches: in state "match" the next state is the
// low two bits of the cell up-left, so the wave reads the next 64 cells of the current diagonal at once (lane t: t steps ahead)
// and a ballot says how far the run of matches goes -- one round trip per run instead of one per base.  Every lane carries the
// same walk state; lane 0 does the stores.

static __device__ __forceinline__ int traceback(const uint8_t* z, bool z_lds, int n_col, int w, int tlen, int qlen, int lane, uint32_t* cigar, int cig_cap, int& err, uint8_t* tile, GClk& K)
{
    gclk(K, 5);
    int n = 0;
    int which = 0, i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
    int cur_op = -1;
    uint32_t cur_len = 0;
    bool ovf = false;
#define TB_PUSH(OP, LEN) do { if ((OP) == cur_op) cur_len += (uint32_t)(LEN); else { \
        if (cur_len) { if (n >= cig_cap) ovf = true; else { if (lane == 0) cigar[n] = cur_len << 4 | (uint32_t)cur_op; ++n; } } cur_op = (OP); cur_len = (uint32_t)(LEN); } } while (0)
    // one stretch of the walk: from (i, k) while the cells are available (AVAIL(r, c)) through GET(r, c)
#define TB_WALK(AVAIL, GET) \
    while (i >= 0 && k >= 0 && !ovf && AVAIL(i, k)) { \
        if (which == 0) { \
            const int r_ = i - lane, c_ = k - lane; \
            const bool in_ = r_ >= 0 && c_ >= 0 && AVAIL(r_, c_); \
            const int d_ = in_ ? (GET(r_, c_) & 3) : 3;                       /* 3: not a state -- the run ends where the cells do */ \
            const unsigned long long nz_ = __ballot(d_ != 0); \
            const int run_ = nz_ ? __ffsll((long long)nz_) - 1 : 64; \
            if (run_ > 0) { TB_PUSH(0, run_); i -= run_; k -= run_; } \
            if (run_ < 64) { \
                const int dn_ = __shfl(d_, run_); \
                if (dn_ == 1) { TB_PUSH(2, 1); --i; which = 1; } \
                else if (dn_ == 2) { TB_PUSH(1, 1); --k; which = 2; } \
            } \
        } else { \
            which = GET(i, k) >> (which << 1) & 3; \
            if (which == 0) { TB_PUSH(0, 1); --i; --k; } \
            else if (which == 1) { TB_PUSH(2, 1); --i; } \
            else { TB_PUSH(1, 1); --k; } \
        } \
    }
    if (!tile) {                                                  // the matrix is in LDS (or small): walked directly
#define TB_ALL(r, c) true
#define TB_Z(r, c) z_get(z + ((int64_t)(r) * n_col + ((c) - ((r) > w ? (r) - w : 0))), z_lds)
        TB_WALK(TB_ALL, TB_Z)
#undef TB_ALL
#undef TB_Z
    } else {
        // The matrix is in global memory (long reads: megabytes per job), and a walk of dependent global loads costs a
        // microsecond a load.  The path moves up and/or left by one cell per step, so the 64 x 64 cells up-left of the current
        // one hold its next 64 steps at least: the wave stages that tile in LDS (64 coalesced byte loads in flight), walks
        // inside it, and so on.
        for (;;) {
            const int i0 = i, k0 = k;
            if (i0 < 0 || k0 < 0 || ovf) break;
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
#define TB_IN_TILE(r, c) ((r) > i0 - 64 && (c) > k0 - 64)
#define TB_TILE(r, c) ((int)*AS_LDS(const uint8_t, tile + ((i0 - (r)) * 64 + ((c) - (k0 - 63)))))
            TB_WALK(TB_IN_TILE, TB_TILE)
#undef TB_IN_TILE
#undef TB_TILE
            gclk(K, 7);
        }
    }
#undef TB_WALK
    if (!ovf && i >= 0) TB_PUSH(2, i + 1);
    if (!ovf && k >= 0) TB_PUSH(1, k + 1);
    if (!ovf) TB_PUSH(-1, 0);                                     // flush the last run
    if (ovf) { if (lane == 0) err |= ERR_CIGAR_CAP; n = 0; }
#undef TB_PUSH
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
