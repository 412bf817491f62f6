// k_pe.hip -- paired-end path: insert-size candidates, mate rescue, pairing, PE records.
//
// Replaces, for the reference call at jnibwa.c:214 with MEM_F_PE set (BwaMemAligner.java:73),
// upstream bwamem_pair.c mem_pestat (candidate selection; the per-orientation statistics are a
// batch-global reduction done on the host), mem_matesw + ksw.c ksw_align2, mem_pair and
// mem_sam_pe.  One lane per read pair.  The local SW upstream is an SSE2
// striped kernel whose observable quirks depend on the striping, so the striped layout is kept
// (lane-by-lane, scalar) exactly as the oracle does.
#include <math.h>
#include "dev_common.h"
#include "kernels.h"
#include "post_common.h"
#include "sw_common.h"
#include "wave_ops.h"


// ------------------------------------------------------------------ insert-size candidates
DEV int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t* dist)
{
    int r1 = (b1 >= l_pac), r2 = (b2 >= l_pac);
    int64_t p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
    *dist = p2 > b1 ? p2 - b1 : b1 - p2;
    return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

DEV int cal_sub(const MemOpt& opt, int n, const AlnReg* a)
{
    int j;
    for (j = 1; j < n; ++j) {
        int b_max = a[j].qb > a[0].qb ? a[j].qb : a[0].qb;
        int e_min = a[j].qe < a[0].qe ? a[j].qe : a[0].qe;
        if (e_min > b_max) {
            int min_l = a[j].qe - a[j].qb < a[0].qe - a[0].qb ? a[j].qe - a[j].qb : a[0].qe - a[0].qb;
            if ((float)(e_min - b_max) >= (float)min_l * opt.mask_level) break;
        }
    }
    return j < n ? a[j].score : opt.min_seed_len * opt.a;
}

// per pair: orientation (0..3, or -1) and insert size that mem_pestat would collect
__global__ void k_pestat_cand(DevIndex ix, MemOpt opt, TileView tv, int8_t* cand_dir, int64_t* cand_is)
{
    int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= tv.n_reads >> 1) return;
    int r0 = pi << 1, r1 = r0 | 1;
    const AlnReg* a0 = tv.regs + tv.seed_off[r0];
    const AlnReg* a1 = tv.regs + tv.seed_off[r1];
    int n0 = tv.n_regs[r0], n1 = tv.n_regs[r1];
    int dir = -1; int64_t is = 0;
    if (n0 && n1 && !(cal_sub(opt, n0, a0) > 0.8 * a0[0].score) && !(cal_sub(opt, n1, a1) > 0.8 * a1[0].score) && a0[0].rid == a1[0].rid) {
        int d = infer_dir(ix.l_pac, a0[0].rb, a1[0].rb, &is);
        if (is && is <= opt.max_ins) dir = d;
    }
    cand_dir[pi] = (int8_t)dir; cand_is[pi] = is;
}

// ------------------------------------------------------------------ mate rescue (mem_matesw)
// Mate rescue in three steps.  For a pair whose mate is missing from the expected window upstream runs ksw_align2 over that
// window, anchor after anchor (up to max_matesw per end), inserting what it finds into the mate's hit list before looking
// at the next anchor.  The alignment itself depends only on the anchor, the orientation and the mate's sequence, so
// (1) k_pe_rescue_plan lists, per pair, every alignment the sequential procedure could ask for -- all (end, anchor,
// orientation) whose window is not already covered by a hit *before* any rescue; hits are only ever added, so this is a
// superset -- (2) k_pe_rescue_sw runs the listed alignments, four per wavefront, spread over the whole GPU, and
// (3) k_pe_pair replays upstream's sequence and picks the results up.  A read with dozens of equally good hits makes
// dozens of requests; done inside the pairing kernel they ran one after another while every other pair had long finished.
size_t pe_rescue_bytes(int what, int cap);
typedef SwJob RescueJob;           // (bwamem_types.h) read: the mate (tile index); tag = end << 16 | anchor << 2 | orientation; q_off = 0
#define RESCUE_NOT_RUN ((int)0x81818181)     // (the byte pattern the result array is preset with)

// window and parameters of the rescue of `a`'s mate in orientation r (mem_matesw); false: no alignment for this orientation
DEV bool rescue_req(const DevIndex& ix, const MemOpt& opt, const MemPestat* pes, const AlnReg& a, int r, int l_ms, int64_t& rb, int64_t& re, int& is_rev, int& xtra)
{
    const int64_t l_pac = ix.l_pac;
    int rid = -1;
    is_rev = (r >> 1 != (r & 1));
    const int is_larger = !(r >> 1);
    if (!is_rev) {
        rb = is_larger ? a.rb + pes[r].low : a.rb - pes[r].high;
        re = (is_larger ? a.rb + pes[r].high : a.rb - pes[r].low) + l_ms;
    } else {
        rb = (is_larger ? a.rb + pes[r].low : a.rb - pes[r].high) - l_ms;
        re = is_larger ? a.rb + pes[r].high : a.rb - pes[r].low;
    }
    if (rb < 0) rb = 0;
    if (re > l_pac << 1) re = l_pac << 1;
    if (rb < re) bns_clamp(ix, rb, (rb + re) >> 1, re, rid);
    if (!(a.rid == rid && re - rb >= opt.min_seed_len)) return false;
    xtra = KSW_XSUBO | KSW_XSTART | (l_ms * opt.a < 250 ? KSW_XBYTE : 0) | (opt.min_seed_len * opt.a);
    return true;
}

// orientations already covered by a hit of the mate (or without statistics)
DEV void rescue_skip(const DevIndex& ix, const MemPestat* pes, const AlnReg& a, int n_ma, const AlnReg* ma, int skip[4])
{
    for (int r = 0; r < 4; ++r) skip[r] = pes[r].failed ? 1 : 0;
    for (int i = 0; i < n_ma; ++i) {
        int64_t dist;
        const int r = infer_dir(ix.l_pac, a.rb, ma[i].rb, &dist);
        if (dist >= pes[r].low && dist <= pes[r].high) skip[r] = 1;
    }
}

// mem_matesw for one anchor; the alignments come from the job list of the pair (jobs[q .. q_end), in tag order).
// settled: the mate's list has been through mem_sort_dedup_patch (with at least two regions) since it was last touched by
// anything else; incr: use matesw_insert (off: BWAMEM_HIP_DEBUGK bit 0x800, every call runs the procedure itself)
DEV int matesw(const DevIndex& ix, const MemOpt& opt, PostScratch& S, SwScratch& W, const MemPestat* pes, const AlnReg& a,
               int l_ms, const uint8_t* ms, int& n_ma, AlnReg* ma, int cap_ma, int& err,
               const RescueJob* jobs, const KswR* results, int& q, int q_end, int tag0, SortKey* keys, bool& settled, bool incr, int* stat)
{
    const int64_t l_pac = ix.l_pac;
    int i, r, skip[4], n = 0;
    rescue_skip(ix, pes, a, n_ma, ma, skip);
    if (skip[0] + skip[1] + skip[2] + skip[3] == 4) return 0;
    for (r = 0; r < 4; ++r) {
        if (skip[r]) continue;
        int is_rev, xtra;
        int64_t rb, re;
        bool ins = false;
        AlnReg b;
        if (rescue_req(ix, opt, pes, a, r, l_ms, rb, re, is_rev, xtra)) {
            KswR aln; aln.score = RESCUE_NOT_RUN;
            while (q < q_end && jobs[q].tag < (tag0 | r)) ++q;
            if (q < q_end && jobs[q].tag == (tag0 | r)) aln = results[q];
            if (aln.score == RESCUE_NOT_RUN) {                 // not listed or too long for the wave kernel: the lane's own scalar kernel
                SwIn I; I.ms = ms; I.l_ms = l_ms; I.is_rev = is_rev; I.qrev = 0; I.t0 = rb; I.trev = 0;
                aln = sw_align2(ix, opt, I, l_ms, (int)(re - rb), xtra, W, err);
            }
            if (aln.score >= opt.min_seed_len && aln.qb >= 0) {
                b.rb = b.re = 0; b.qb = b.qe = 0; b.rid = 0; b.score = b.truesc = b.sub = b.alt_sc = b.csub = b.sub_n = b.w = b.seedcov = 0;
                b.secondary = b.secondary_all = b.seedlen0 = b.n_comp = b.is_alt = 0; b.frac_rep = 0.f; b.pad_ = 0; b.hash = 0;
                b.rid = a.rid;
                b.is_alt = a.is_alt;
                b.qb = is_rev ? l_ms - (aln.qe + 1) : aln.qb;
                b.qe = is_rev ? l_ms - aln.qb : aln.qe + 1;
                b.rb = is_rev ? (l_pac << 1) - (rb + aln.te + 1) : rb + aln.tb;
                b.re = is_rev ? (l_pac << 1) - (rb + aln.tb) : rb + aln.te + 1;
                b.score = aln.score;
                b.csub = aln.score2;
                b.secondary = -1;
                b.seedcov = (int)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
                if (n_ma >= cap_ma) { err |= ERR_SCRATCH; return n; }
                ins = true;
            }
            ++n;
        }
        if (ins) ++stat[0];
        if (ins && !(incr && settled && n_ma >= 1 && matesw_insert(opt, b, n_ma, ma))) {
            if (incr && settled && n_ma >= 1) ++stat[1];
            ++n_ma;
            for (i = 0; i < n_ma - 1; ++i) if (ma[i].score < b.score) break;
            int tmp = i;
            for (i = n_ma - 1; i > tmp; --i) ma[i] = ma[i - 1];
            ma[i] = b;
            settled = false;
        }
        if (n && !(incr && settled)) {
            ++stat[2];
            settled = n_ma >= 2;                               // (below two regions the procedure returns at once)
            n_ma = sort_dedup_patch(ix, opt, S, 0, n_ma, ma, 0, keys);
        }
    }
    return n;
}

// ------------------------------------------------------------------ pairing (mem_pair)
struct Pair64 { uint64_t x, y; };
struct Pair64Lt { __device__ bool operator()(const Pair64& a, const Pair64& b) const { return a.x < b.x || (a.x == b.x && a.y < b.y); } };

// The insert-size term of a pair's score, log(2 erfc(|dist - avg| / std / sqrt 2)), is a libm value: it comes from a
// host-built (glibc) table over the integer distances an orientation admits (PairTab, pipeline.cpp: build_pair_tab), so
// that the truncation below sees the very doubles the reference's host code sees.  Distances outside the
// table are those for which glibc's erfc is exactly 0.
DEV int mem_pair(const DevIndex& ix, const MemOpt& opt, const MemPestat* pes, const PairTab& pt, const AlnReg* a0, const AlnReg* a1, int id,
                 int* sub, int* n_sub, int z[2], const int n_pri[2], Pair64* v, Pair64* u, int cap_u, int& err)
{
    const int64_t l_pac = ix.l_pac;
    int r, i, k, y[4], ret, nv = 0, nu = 0;
    for (r = 0; r < 2; ++r) {
        const AlnReg* a = r ? a1 : a0;
        for (i = 0; i < n_pri[r]; ++i) {
            Pair64 key;
            const AlnReg* e = &a[i];
            key.x = (uint64_t)(e->rb < l_pac ? e->rb : (l_pac << 1) - 1 - e->rb);
            key.x = (uint64_t)e->rid << 32 | (key.x - (uint64_t)ix.ann_offset[e->rid]);
            key.y = (uint64_t)e->score << 32 | (uint64_t)(int64_t)(i << 2 | (e->rb >= l_pac) << 1 | r);
            v[nv++] = key;
        }
    }
    ks_introsort((size_t)nv, v, Pair64Lt());
    y[0] = y[1] = y[2] = y[3] = -1;
    for (i = 0; i < nv; ++i) {
        for (r = 0; r < 2; ++r) {
            int dir = r << 1 | (int)(v[i].y >> 1 & 1), which;
            if (pes[dir].failed) continue;
            which = r << 1 | (int)((v[i].y & 1) ^ 1);
            if (y[which] < 0) continue;
            for (k = y[which]; k >= 0; --k) {
                int64_t dist;
                int q;
                double lg;
                if ((int)(v[k].y & 3) != which) continue;
                dist = (int64_t)v[i].x - (int64_t)v[k].x;
                if (dist > pes[dir].high) break;
                if (dist < pes[dir].low) continue;
                {
                    const int64_t o = dist - pt.lo[dir];
                    lg = o >= 0 && o < (int64_t)pt.n[dir] ? pt.t[pt.off[dir] + o] : -__builtin_inf();
                }
                q = (int)((v[i].y >> 32) + (v[k].y >> 32) + .721 * lg * opt.a + .499);
                if (q < 0) q = 0;
                if (nu >= cap_u) { err |= ERR_SCRATCH; return 0; }
                Pair64* p = &u[nu++];
                p->y = (uint64_t)k << 32 | (uint64_t)i;
                p->x = (uint64_t)q << 32 | (hash_64(p->y ^ (uint64_t)(int64_t)(id << 8)) & 0xffffffffU);
            }
        }
        y[v[i].y & 3] = i;
    }
    if (nu) {
        int tmp = opt.a + opt.b;
        tmp = tmp > opt.o_del + opt.e_del ? tmp : opt.o_del + opt.e_del;
        tmp = tmp > opt.o_ins + opt.e_ins ? tmp : opt.o_ins + opt.e_ins;
        ks_introsort((size_t)nu, u, Pair64Lt());
        i = (int)(u[nu - 1].y >> 32); k = (int)(u[nu - 1].y << 32 >> 32);
        z[v[i].y & 1] = (int)(v[i].y << 32 >> 34);
        z[v[k].y & 1] = (int)(v[k].y << 32 >> 34);
        ret = (int)(u[nu - 1].x >> 32);
        *sub = nu > 1 ? (int)(u[nu - 2].x >> 32) : 0;
        for (i = nu - 2, *n_sub = 0; i >= 0; --i)
            if (*sub - (int)(u[i].x >> 32) <= tmp) ++*n_sub;
    } else { ret = 0; *sub = 0; *n_sub = 0; }
    return ret;
}

DEV MateInfo mate_of(const AlnRec& h) { MateInfo m; m.rid = h.rid; m.pos = h.pos; m.is_rev = h.is_rev; m.ref_len = h.ref_len; return m; }

#define RAW_MAPQ(diff, a) ((int)(6.02 * (diff) / (a) + .499))

// per-pair scratch layout handed to k_final_pe
struct PeView {
    AlnReg* regs;             // pool with room for rescued regions
    const int64_t* reg_off;   // [n_reads+1]
    int32_t* n_regs;          // [n_reads] in/out
    int32_t* ints;            // 2 ints per region slot (zbuf)
    uint8_t* scratch;         // per pair
    int64_t scratch_per_pair;
    void* vpool;              // Pair64 per region slot (indexed by reg_off of the pair's first read)
    void* keys;               // SortKey per region slot (post_common.h: sort_regs), or null
    int cap_h, cap_b, cap_u;
    PairTab ptab;             // host-built insert-size score terms (mem_pair)
};

// what the pairing stage decides for one pair and the record stage needs back
struct PeState { int32_t paired, z0, z1, n_pri0, n_pri1, extra_flag, q_se0, q_se1; };

// common per-pair set-up of the two stages
struct PeCtx {
    int rd[2]; const uint8_t* seq[2]; int l_seq[2]; AlnReg* a[2]; int n[2], cap[2]; int32_t* zb[2];
};
DEV PeCtx pe_ctx(const TileView& tv, const PeView& pv, int pi)
{
    PeCtx c;
    c.rd[0] = pi << 1; c.rd[1] = pi << 1 | 1;
    for (int i = 0; i < 2; ++i) {
        c.seq[i] = tv.seq + tv.seq_off[c.rd[i]];
        c.l_seq[i] = (int)(tv.seq_off[c.rd[i] + 1] - tv.seq_off[c.rd[i]] - 1);
        c.a[i] = pv.regs + pv.reg_off[c.rd[i]];
        c.cap[i] = (int)(pv.reg_off[c.rd[i] + 1] - pv.reg_off[c.rd[i]]);
        c.n[i] = pv.n_regs[c.rd[i]];
        c.zb[i] = pv.ints + 2 * pv.reg_off[c.rd[i]];
    }
    return c;
}

// per-pair scratch of the pairing stage: the scalar SW arrays, the rescue anchors of both ends, mem_pair's u array
struct PeScratch { SwScratch W; AlnReg* anchors[2]; Pair64* u; };
DEV PeScratch pe_scratch(const MemOpt& opt, const PeView& pv, int pi)
{
    PeScratch P;
    uint8_t* sp = pv.scratch + (size_t)pi * pv.scratch_per_pair;
    P.W.cap_h = pv.cap_h; P.W.cap_b = pv.cap_b;
    P.W.H0 = (int32_t*)sp; sp += (size_t)P.W.cap_h * 4; P.W.H1 = (int32_t*)sp; sp += (size_t)P.W.cap_h * 4;
    P.W.E = (int32_t*)sp; sp += (size_t)P.W.cap_h * 4; P.W.Hmax = (int32_t*)sp; sp += (size_t)P.W.cap_h * 4;
    P.W.b = (uint64_t*)sp; sp += (size_t)P.W.cap_b * 8;
    P.anchors[0] = (AlnReg*)sp; sp += (size_t)opt.max_matesw * sizeof(AlnReg);
    P.anchors[1] = (AlnReg*)sp; sp += (size_t)opt.max_matesw * sizeof(AlnReg);
    P.u = (Pair64*)sp;
    return P;
}
// the rescue anchors of both ends: hits within pen_unpaired of the best (copies: the hit lists change during rescue)
DEV void pe_anchors(const MemOpt& opt, const PeCtx& c, PeScratch& P, int n_anch[2])
{
    for (int i = 0; i < 2; ++i) {
        n_anch[i] = 0;
        for (int j = 0; j < c.n[i]; ++j)
            if (c.a[i][j].score >= c.a[i][0].score - opt.pen_unpaired) { if (n_anch[i] < opt.max_matesw) P.anchors[i][n_anch[i]] = c.a[i][j]; ++n_anch[i]; }
    }
}

// Pairs whose ends carry many regions (reads in repeat families) get a wavefront each for the list work of mate rescue --
// which orientations an anchor's window already holds a hit for, where a rescued region goes, what it displaces: all of it
// linear in the mate's list and repeated per anchor -- instead of one lane (k_pe_rescue_plan_wave, k_pe_matesw_wave).
#define PE_HEAVY_MIN 32
#define PE_WAVE_MAX_ANCHORS 128
DEV bool pe_heavy(const MemOpt& opt, int debug, int n0, int n1) { return n0 + n1 >= PE_HEAVY_MIN && n0 > 0 && n1 > 0 && opt.max_matesw <= PE_WAVE_MAX_ANCHORS && !(debug & 0x4000); }

// mate rescue, step 1 (one lane per pair): list the alignments the rescue of this pair may ask for
__global__ void k_pe_rescue_plan(DevIndex ix, MemOpt opt, TileView tv, PeView pv, MemPestat p0, MemPestat p1, MemPestat p2, MemPestat p3,
                                 RescueJob* jobs, int32_t* job_first, int32_t* job_num, int32_t* counter, int cap, int32_t* heavy_list, int32_t* heavy_cnt)
{
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = pi < tv.n_reads >> 1;
    const MemPestat pes[4] = { p0, p1, p2, p3 };
    PeCtx c = pe_ctx(tv, pv, valid ? pi : 0);
    {   // the heavy pairs are the wavefront kernels': listed here (one atomic per wavefront), so that those kernels are launched
        // over the list and not over every pair (an empty workgroup of theirs costs ~8 ns: 2.4 % of a paired-end step on a genome
        // that has no heavy pair at all)
        const bool heavy = valid && pe_heavy(opt, tv.debug, c.n[0], c.n[1]);
        const unsigned long long hm = __ballot(heavy);
        if (hm) {
            const int l_ = threadIdx.x & 63, src = __ffsll((long long)hm) - 1;
            int base = 0;
            if (l_ == src) base = atomicAdd(heavy_cnt, __popcll(hm));
            base = __shfl(base, src);
            if (heavy) heavy_list[base + __popcll(hm & ((1ull << l_) - 1ull))] = pi;
        }
        if (!valid || heavy) return;
    }
    PeScratch P = pe_scratch(opt, pv, pi);
    const long long t0 = (tv.debug & 0x2000) ? clock64() : 0;
    int n_anch[2];
    pe_anchors(opt, c, P, n_anch);
    int first = 0, cnt = 0;
    for (int pass = 0; pass < 2; ++pass) {                      // count, reserve, write
        int k = 0;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < n_anch[i] && j < opt.max_matesw; ++j) {
                const AlnReg anc = P.anchors[i][j];
                int skip[4];
                rescue_skip(ix, pes, anc, c.n[!i], c.a[!i], skip);
                for (int r = 0; r < 4; ++r) {
                    if (skip[r]) continue;
                    int is_rev, xtra; int64_t rb, re;
                    if (!rescue_req(ix, opt, pes, anc, r, c.l_seq[!i], rb, re, is_rev, xtra)) continue;
                    if (pass == 1 && k < cnt) {
                        RescueJob jb; jb.rb = rb; jb.read = c.rd[!i]; jb.tag = i << 16 | j << 2 | r; jb.l_ms = c.l_seq[!i]; jb.is_rev = is_rev; jb.tlen = (int)(re - rb); jb.xtra = xtra; jb.q_off = 0; jb.pad_ = 0;
                        jobs[first + k] = jb;
                    }
                    ++k;
                }
            }
        if (pass == 0) {
            cnt = k;
            if (cnt > 0) {
                first = atomicAdd(counter, cnt);
                if (first + cnt > cap) { atomicOr(tv.err, ERR_RESCUE_CAP); cnt = 0; }     // the list is then incomplete *and* has unwritten slots below cap: nobody may run it
            }
            if (cnt == 0) break;
        }
    }
    job_first[pi] = first; job_num[pi] = cnt;
    if ((tv.debug & 0x2000) && __ffsll((long long)__ballot(1)) - 1 == (int)(threadIdx.x & 63)) atomicAdd(&tv.cnt->dbg[4], (unsigned long long)(clock64() - t0));
}

// rescue_skip by the wavefront: bit r set = orientation r needs no rescue
DEV int rescue_skip_wave(const DevIndex& ix, const MemPestat* pes, const AlnReg& a, int n_ma, const AlnReg* ma, int lane)
{
    int m = 0, mine = 0;
    for (int r = 0; r < 4; ++r) m |= pes[r].failed ? 1 << r : 0;
    for (int k = lane; k < n_ma; k += 64) {
        int64_t dist;
        const int r = infer_dir(ix.l_pac, a.rb, ma[k].rb, &dist);
        if (dist >= pes[r].low && dist <= pes[r].high) mine |= 1 << r;
    }
    for (int r = 0; r < 4; ++r) if (__ballot(mine >> r & 1)) m |= 1 << r;
    return m;
}

// mate rescue, step 1 for a heavy pair: one wavefront (workgroup) per pair; light pairs return at once
__global__ void __launch_bounds__(64) k_pe_rescue_plan_wave(DevIndex ix, MemOpt opt, TileView tv, PeView pv, MemPestat p0, MemPestat p1, MemPestat p2, MemPestat p3,
                                                            RescueJob* jobs, int32_t* job_first, int32_t* job_num, int32_t* counter, int cap, const int32_t* heavy_list, const int32_t* heavy_cnt)
{
    __shared__ uint8_t skip_s[2 * PE_WAVE_MAX_ANCHORS];
    const int lane = threadIdx.x;
    const int n_heavy = *heavy_cnt;
  for (int hi = blockIdx.x; hi < n_heavy; hi += (int)gridDim.x) {       // (a bounded grid walks the list of heavy pairs)
    const int pi = heavy_list[hi];
    __syncthreads();                                                    // the previous pair is done with skip_s
    PeCtx c = pe_ctx(tv, pv, pi);
    const MemPestat pes[4] = { p0, p1, p2, p3 };
    PeScratch P = pe_scratch(opt, pv, pi);
    int n_anch[2] = { 0, 0 };
    if (lane == 0) pe_anchors(opt, c, P, n_anch);
    for (int i = 0; i < 2; ++i) { n_anch[i] = __shfl(n_anch[i], 0); n_anch[i] = n_anch[i] < opt.max_matesw ? n_anch[i] : opt.max_matesw; }
    __syncthreads();
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < n_anch[i]; ++j) {
            const int m = rescue_skip_wave(ix, pes, P.anchors[i][j], c.n[!i], c.a[!i], lane);
            if (lane == 0) skip_s[i * PE_WAVE_MAX_ANCHORS + j] = (uint8_t)m;
        }
    __syncthreads();
    const int n_combo = 4 * (n_anch[0] + n_anch[1]);
    int first = 0, cnt = 0;
    for (int pass = 0; pass < 2; ++pass) {                      // count, reserve, write (in tag order: end, anchor, orientation)
        int k = 0;
        for (int c0 = 0; c0 < n_combo; c0 += 64) {
            const int cc = c0 + lane;
            bool ok = false;
            RescueJob jb;
            if (cc < n_combo) {
                const int ai = cc >> 2, r = cc & 3, i = ai >= n_anch[0] ? 1 : 0, j = ai - (i ? n_anch[0] : 0);
                if (!(skip_s[i * PE_WAVE_MAX_ANCHORS + j] >> r & 1)) {
                    int is_rev, xtra; int64_t rb, re;
                    ok = rescue_req(ix, opt, pes, P.anchors[i][j], r, c.l_seq[!i], rb, re, is_rev, xtra);
                    jb.rb = rb; jb.read = c.rd[!i]; jb.tag = i << 16 | j << 2 | r; jb.l_ms = c.l_seq[!i]; jb.is_rev = is_rev; jb.tlen = (int)(re - rb); jb.xtra = xtra; jb.q_off = 0; jb.pad_ = 0;
                }
            }
            const uint64_t bal = __ballot(ok);
            if (pass == 1 && ok && cnt > 0) jobs[first + k + __popcll(bal & ((1ull << lane) - 1))] = jb;
            k += __popcll(bal);
        }
        if (pass == 0) {
            cnt = k;
            if (cnt > 0) {
                if (lane == 0) first = atomicAdd(counter, cnt);
                first = __shfl(first, 0);
                if (first + cnt > cap) { if (lane == 0) atomicOr(tv.err, ERR_RESCUE_CAP); cnt = 0; }
            }
            if (cnt == 0) break;
        }
    }
    if (lane == 0) { job_first[pi] = first; job_num[pi] = cnt; }
  }
}

// mate rescue, step 2: ksw_align2 for the listed alignments, four per wavefront (one per 16-lane group, sw_common.h).
// Instantiated for queries of up to 10 segments (150 bp mates) and up to 32 (250 bp mates in 16-bit mode); an instance
// leaves the jobs outside (LO, NSEG] segments alone.
template <int LO, int NSEG, int GW>
__global__ void __launch_bounds__(64) k_pe_rescue_sw(DevIndex ix, MemOpt opt, TileView tv, const RescueJob* jobs, const int32_t* counter, int cap, KswR* results, int cap_b)
{
    HIP_DYNAMIC_SHARED(uint64_t, blists)
    constexpr int PER = 64 / GW;                                 // alignments per wavefront: 4 in byte mode (GW = 16), 8 in 16-bit mode (GW = 8)
    const int lane = threadIdx.x;
    if (tv.err[0] & ERR_RESCUE_CAP) return;                      // the job list did not fit: the host sizes it from the count and runs the stage again
    const int n = *counter < cap ? *counter : cap;
    if ((int)blockIdx.x * PER >= n) return;
    const int job = blockIdx.x * PER + lane / GW;
    const bool on = job < n;
    RescueJob jb; jb.rb = 0; jb.read = 0; jb.tag = 0; jb.l_ms = 0; jb.is_rev = 0; jb.tlen = 0; jb.xtra = 0; jb.q_off = 0; jb.pad_ = 0;
    if (on) jb = jobs[job];
    SwLds L; L.b = blists; L.cap_b = cap_b;
    SwIn I; I.ms = tv.seq + tv.seq_off[jb.read] + jb.q_off; I.l_ms = jb.l_ms; I.is_rev = jb.is_rev; I.qrev = 0; I.t0 = jb.rb; I.trev = 0;
    const bool u8 = (jb.xtra & KSW_XBYTE) != 0;
    const int p = u8 ? 16 : 8;
    const int slen = (jb.l_ms + p - 1) / p;
    const bool fits = on && u8 == (GW == 16) && slen > LO && slen <= NSEG;       // this instance's mode and range of segment counts
    if (__ballot(fits) == 0ull) return;
    int err = 0;
    const KswR res = sw_align2_wave4<NSEG, GW>(ix, opt, I, fits, GW == 16 ? 1 : 2, jb.l_ms, jb.tlen, jb.xtra, L, lane, err);
    if (fits && lane % GW == 0) results[job] = res;
    if (err) atomicOr(tv.err, err);
}

// the same with two alignments per lane (sw_common.h: sw_core_packed): byte-mode jobs (U8) of LO < segments <= NSEG eight per
// wavefront, 16-bit jobs sixteen; leaves every other job alone
template <int LO, int NSEG, bool U8>
__global__ void __launch_bounds__(64, NSEG <= 16 ? 2 : 1) k_pe_rescue_sw2(DevIndex ix, MemOpt opt, TileView tv, const RescueJob* jobs, const int32_t* counter, int cap, KswR* results, int cap_b)
{
    HIP_DYNAMIC_SHARED(uint64_t, blists)
    constexpr int GW = U8 ? 16 : 8, PER = 2 * (64 / GW);
    const int lane = threadIdx.x;
    if (tv.err[0] & ERR_RESCUE_CAP) return;
    const int n = *counter < cap ? *counter : cap;
    if ((int)blockIdx.x * PER >= n) return;
    SwPair P;
    int job[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        job[h] = blockIdx.x * PER + (lane / GW) * 2 + h;
        RescueJob jb; jb.rb = 0; jb.read = 0; jb.tag = 0; jb.l_ms = 0; jb.is_rev = 0; jb.tlen = 0; jb.xtra = 0; jb.q_off = 0; jb.pad_ = 0;
        if (job[h] < n) jb = jobs[job[h]];
        const int slen = (jb.l_ms + GW - 1) / GW;
        P.on[h] = job[h] < n && ((jb.xtra & KSW_XBYTE) != 0) == U8 && slen > LO && slen <= NSEG;
        P.I[h].ms = tv.seq + tv.seq_off[jb.read] + jb.q_off; P.I[h].l_ms = jb.l_ms; P.I[h].is_rev = jb.is_rev; P.I[h].qrev = 0; P.I[h].t0 = jb.rb; P.I[h].trev = 0;
        P.qlen[h] = jb.l_ms; P.tlen[h] = jb.tlen; P.xtra[h] = jb.xtra;
    }
    if (__ballot(P.on[0] || P.on[1]) == 0ull) return;
    SwLds L; L.b = blists; L.cap_b = cap_b;
    int err = 0;
    KswR R[2];
    sw_align2_packed<NSEG, U8>(ix, opt, P, L, lane, err, R);
#pragma unroll
    for (int h = 0; h < 2; ++h) if (P.on[h] && lane % GW == 0) results[job[h]] = R[h];
    if (err) atomicOr(tv.err, err);
}

// mem_matesw for one anchor of a heavy pair, by the wavefront (uniform arguments; see matesw)
DEV void matesw_wave(const DevIndex& ix, const MemOpt& opt, SwScratch& W, const MemPestat* pes, const AlnReg& a,
                     int l_ms, const uint8_t* ms, int& n_ma, AlnReg* ma, int cap_ma, int& err,
                     const RescueJob* jobs, const KswR* results, int& q, int q_end, int tag0, SortKey* keys, bool& settled, int lane)
{
    const int64_t l_pac = ix.l_pac;
    const int skip = rescue_skip_wave(ix, pes, a, n_ma, ma, lane);
    if (skip == 15) return;
    int n = 0;
    for (int r = 0; r < 4; ++r) {
        if (skip >> r & 1) continue;
        int is_rev, xtra;
        int64_t rb, re;
        bool ins = false;
        AlnReg b;
        if (rescue_req(ix, opt, pes, a, r, l_ms, rb, re, is_rev, xtra)) {
            KswR aln; aln.score = RESCUE_NOT_RUN; aln.te = aln.qe = aln.score2 = aln.te2 = aln.tb = aln.qb = 0;
            while (q < q_end && jobs[q].tag < (tag0 | r)) ++q;
            if (q < q_end && jobs[q].tag == (tag0 | r)) aln = results[q];
            if (aln.score == RESCUE_NOT_RUN) {                 // a mate too long for the wave kernel: one lane's scalar kernel
                int e = 0;
                if (lane == 0) { SwIn I; I.ms = ms; I.l_ms = l_ms; I.is_rev = is_rev; I.qrev = 0; I.t0 = rb; I.trev = 0; aln = sw_align2(ix, opt, I, l_ms, (int)(re - rb), xtra, W, e); }
                aln.score = __shfl(aln.score, 0); aln.te = __shfl(aln.te, 0); aln.qe = __shfl(aln.qe, 0); aln.score2 = __shfl(aln.score2, 0);
                aln.te2 = __shfl(aln.te2, 0); aln.tb = __shfl(aln.tb, 0); aln.qb = __shfl(aln.qb, 0);
                err |= __shfl(e, 0);
            }
            if (aln.score >= opt.min_seed_len && aln.qb >= 0) {
                b.rb = b.re = 0; b.qb = b.qe = 0; b.rid = 0; b.score = b.truesc = b.sub = b.alt_sc = b.csub = b.sub_n = b.w = b.seedcov = 0;
                b.secondary = b.secondary_all = b.seedlen0 = b.n_comp = b.is_alt = 0; b.frac_rep = 0.f; b.pad_ = 0; b.hash = 0;
                b.rid = a.rid;
                b.is_alt = a.is_alt;
                b.qb = is_rev ? l_ms - (aln.qe + 1) : aln.qb;
                b.qe = is_rev ? l_ms - aln.qb : aln.qe + 1;
                b.rb = is_rev ? (l_pac << 1) - (rb + aln.te + 1) : rb + aln.tb;
                b.re = is_rev ? (l_pac << 1) - (rb + aln.tb) : rb + aln.te + 1;
                b.score = aln.score;
                b.csub = aln.score2;
                b.secondary = -1;
                b.seedcov = (int)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
                if (n_ma >= cap_ma) { err |= ERR_SCRATCH; return; }
                ins = true;
            }
            ++n;
        }
        if (ins && !(settled && n_ma >= 1 && matesw_insert_wave(opt, b, n_ma, ma, lane))) {
            int pos = n_ma;                                     // in front of the first region that scores lower
            for (int base = 0; base < n_ma; base += 64) {
                const int k = base + lane;
                const uint64_t lower = __ballot(k < n_ma && ma[k].score < b.score);
                if (lower) { pos = base + __ffsll((long long)lower) - 1; break; }
            }
            regs_insert_wave(ma, n_ma, pos, b, lane);
            ++n_ma;
            settled = false;
        }
        if (n && !settled) {
            settled = n_ma >= 2;
            n_ma = sort_dedup_nq_wave(opt, n_ma, ma, keys, lane);
        }
    }
}

// mate rescue, step 3 for the heavy pairs: one wavefront per pair replays upstream's sequence on both ends' lists and leaves
// the new region counts; k_pe_pair then finds job_num < 0 and goes straight on to primary marking
__global__ void __launch_bounds__(64) k_pe_matesw_wave(DevIndex ix, MemOpt opt, TileView tv, PeView pv, MemPestat p0, MemPestat p1, MemPestat p2, MemPestat p3,
                                                       const RescueJob* rjobs, const KswR* rres, const int32_t* job_first, int32_t* job_num, const int32_t* heavy_list, const int32_t* heavy_cnt)
{
    const int lane = threadIdx.x;
    if (tv.err[0] & ERR_RESCUE_CAP) return;
    const int n_heavy = *heavy_cnt;
  for (int hi = blockIdx.x; hi < n_heavy; hi += (int)gridDim.x) {
    const int pi = heavy_list[hi];
    __syncthreads();
    PeCtx c = pe_ctx(tv, pv, pi);
    const MemPestat pes[4] = { p0, p1, p2, p3 };
    PeScratch P = pe_scratch(opt, pv, pi);
    SortKey* const kbase = pv.keys && !(tv.debug & 0x400) ? (SortKey*)pv.keys : nullptr;
    int n_anch[2] = { 0, 0 };
    if (lane == 0) pe_anchors(opt, c, P, n_anch);
    for (int i = 0; i < 2; ++i) n_anch[i] = __shfl(n_anch[i], 0);
    __syncthreads();
    int err = 0, q = job_first[pi];
    const int q_end = q + job_num[pi];
    bool settled[2] = { false, false };
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < n_anch[i] && j < opt.max_matesw; ++j)
            matesw_wave(ix, opt, P.W, pes, P.anchors[i][j], c.l_seq[!i], c.seq[!i], c.n[!i], c.a[!i], c.cap[!i], err, rjobs, rres, q, q_end, i << 16 | j << 2,
                        kbase ? kbase + pv.reg_off[c.rd[!i]] : nullptr, settled[!i], lane);
    if (lane == 0) {
        pv.n_regs[c.rd[0]] = c.n[0]; pv.n_regs[c.rd[1]] = c.n[1];
        job_num[pi] = -1;
        if (err) atomicOr(tv.err, err);
    }
  }
}

// mem_sam_pe, first half (one lane per pair): mate rescue (step 3: upstream's sequence with the alignments precomputed),
// primary marking, pairing and the mapping-quality decisions.
// Regions whose CIGAR needs a banded global alignment are then listed as jobs for k_gcigar_lane / k_gcigar (those the record
// stage will use: records, XA entries, mate summaries), and the record stage picks the results up: a one-lane DP inside
// this kernel would stall the other 63 pairs of the wave.
__global__ void k_pe_pair(DevIndex ix, MemOpt opt, TileView tv, PeView pv, MemPestat p0, MemPestat p1, MemPestat p2, MemPestat p3, PeState* states,
                          const RescueJob* rjobs, const KswR* rres, const int32_t* job_first, const int32_t* job_num)
{
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= tv.n_reads >> 1) return;
    if (tv.err[0] & ERR_RESCUE_CAP) return;                      // rescue list incomplete (see k_pe_rescue_plan): this attempt is void, and without the
                                                                 // precomputed alignments every lane would run them itself, one cell at a time
    const MemPestat pes[4] = { p0, p1, p2, p3 };
    PeCtx c = pe_ctx(tv, pv, pi);
    const int* rd = c.rd; const uint8_t** seq = c.seq; int* l_seq = c.l_seq; AlnReg** a = c.a; int* n = c.n; int* cap = c.cap; int32_t** zb = c.zb;
    PostScratch S = post_scratch_for(tv, rd[0]);
    int err = 0;
    PeScratch P = pe_scratch(opt, pv, pi);
    Pair64* v = (Pair64*)pv.vpool + pv.reg_off[rd[0]];      // cap[0] + cap[1] entries >= n_pri[0] + n_pri[1]
    Pair64* u = P.u;
    const int cap_u = pv.cap_u;

    // key records for the region sorts of each end (a pair in a repeat family sorts hundreds of regions once per rescued anchor)
    // (offsets, not an array of pointers: a pointer reloaded from private memory loses its address space)
    SortKey* const kbase = pv.keys && !(tv.debug & 0x400) ? (SortKey*)pv.keys : nullptr;
    const int64_t koff[2] = { pv.reg_off[rd[0]], pv.reg_off[rd[1]] };
    const uint64_t id = (uint64_t)((tv.read_id0 >> 1) + pi);
    int z[2] = { 0, 0 }, o = 0, subo = 0, n_sub = 0, extra_flag = 1, n_pri[2], q_se[2] = { 0, 0 };
    const bool clk = (tv.debug & 0x2000) != 0;
    long long t_prev = clk ? clock64() : 0;
#define PE_CLK(k) do { if (clk) { const long long t_ = clock64(); if (__ffsll((long long)__ballot(1)) - 1 == (int)(threadIdx.x & 63)) atomicAdd(&tv.cnt->dbg[k], (unsigned long long)(t_ - t_prev)); t_prev = t_; } } while (0)
    if (!(opt.flag & MEM_F_NO_RESCUE) && job_num[pi] >= 0) {  // mate rescue from the best hits of each end (< 0: done by k_pe_matesw_wave)
        int n_anch[2];
        pe_anchors(opt, c, P, n_anch);
        int q = job_first[pi];
        const int q_end = q + job_num[pi];
        bool settled[2] = { false, false };
        const bool incr = !(tv.debug & 0x800);
        int stat[3] = { 0, 0, 0 };                               // rescued regions, of which declined by matesw_insert, full mem_sort_dedup_patch calls
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < n_anch[i] && j < opt.max_matesw; ++j)
                matesw(ix, opt, S, P.W, pes, P.anchors[i][j], l_seq[!i], seq[!i], n[!i], a[!i], cap[!i], err, rjobs, rres, q, q_end, i << 16 | j << 2, kbase ? kbase + koff[!i] : nullptr, settled[!i], incr, stat);
        if ((tv.debug & 0x1000) && stat[0] + stat[2]) printf("[k_pe_pair] pair %d: %d + %d regions, %d rescued, %d declined, %d full calls\n", pi, n[0], n[1], stat[0], stat[1], stat[2]);
    }
    PE_CLK(0);
    n_pri[0] = mark_primary_se(opt, n[0], a[0], (int64_t)(id << 1 | 0), zb[0], kbase ? kbase + koff[0] : nullptr);
    n_pri[1] = mark_primary_se(opt, n[1], a[1], (int64_t)(id << 1 | 1), zb[1], kbase ? kbase + koff[1] : nullptr);
    if (opt.flag & MEM_F_PRIMARY5) { reorder_primary5(opt.T, n[0], a[0]); reorder_primary5(opt.T, n[1], a[1]); }

    PE_CLK(1);
    bool paired = false;
    if (!(opt.flag & MEM_F_NOPAIRING) && n_pri[0] && n_pri[1]
        && (o = mem_pair(ix, opt, pes, pv.ptab, a[0], a[1], (int)id, &subo, &n_sub, z, n_pri, v, u, cap_u, err)) > 0) {
        int is_multi[2], q_pe, score_un;
        for (int i = 0; i < 2; ++i) {
            int j;
            for (j = 1; j < n_pri[i]; ++j)
                if (a[i][j].secondary < 0 && a[i][j].score >= opt.T) break;
            is_multi[i] = j < n_pri[i] ? 1 : 0;
        }
        if (!(is_multi[0] || is_multi[1])) {
            paired = true;
            score_un = a[0][0].score + a[1][0].score - opt.pen_unpaired;
            subo = subo > score_un ? subo : score_un;
            q_pe = RAW_MAPQ(o - subo, opt.a);
            if (n_sub > 0) {
                if (n_sub + 1 >= ix.log_tab_n) err |= ERR_SCRATCH;
                else q_pe -= (int)(4.343 * ix.log_tab[n_sub + 1] + .499);
            }
            if (q_pe < 0) q_pe = 0;
            if (q_pe > 60) q_pe = 60;
            q_pe = (int)(q_pe * (1. - .5 * (a[0][0].frac_rep + a[1][0].frac_rep)) + .499);
            if (o > score_un) {                                 // the paired alignment is preferred
                AlnReg* cc[2] = { &a[0][z[0]], &a[1][z[1]] };
                for (int i = 0; i < 2; ++i) {
                    if (cc[i]->secondary >= 0) { cc[i]->sub = a[i][cc[i]->secondary].score; cc[i]->secondary = -2; }
                    q_se[i] = approx_mapq_se(ix, opt, S, *cc[i]);
                }
                q_se[0] = q_se[0] > q_pe ? q_se[0] : q_pe < q_se[0] + 40 ? q_pe : q_se[0] + 40;
                q_se[1] = q_se[1] > q_pe ? q_se[1] : q_pe < q_se[1] + 40 ? q_pe : q_se[1] + 40;
                extra_flag |= 2;
                q_se[0] = q_se[0] < RAW_MAPQ(cc[0]->score - cc[0]->csub, opt.a) ? q_se[0] : RAW_MAPQ(cc[0]->score - cc[0]->csub, opt.a);
                q_se[1] = q_se[1] < RAW_MAPQ(cc[1]->score - cc[1]->csub, opt.a) ? q_se[1] : RAW_MAPQ(cc[1]->score - cc[1]->csub, opt.a);
            } else {
                z[0] = z[1] = 0;
                q_se[0] = approx_mapq_se(ix, opt, S, a[0][0]);
                q_se[1] = approx_mapq_se(ix, opt, S, a[1][0]);
            }
            for (int i = 0; i < 2; ++i) {
                int k = a[i][z[i]].secondary_all;
                if (k >= 0 && k < n_pri[i]) {                   // swap primary and secondary when both are non-ALT
                    for (int j = 0; j < n[i]; ++j)
                        if (a[i][j].secondary_all == k || j == k) a[i][j].secondary_all = z[i];
                    a[i][z[i]].secondary_all = -1;
                }
            }
        }
    }
    PE_CLK(2);
    PeState st; st.paired = paired; st.z0 = z[0]; st.z1 = z[1]; st.n_pri0 = n_pri[0]; st.n_pri1 = n_pri[1]; st.extra_flag = extra_flag; st.q_se0 = q_se[0]; st.q_se1 = q_se[1];
    states[pi] = st;
    // Global-alignment jobs: the regions the record stage will turn into a record, an XA entry or a mate summary and whose
    // CIGAR needs DP (same selection as k_pe_out, evaluated here with the final z / n_pri).  A region missed here would
    // still come out right -- reg2aln then runs the one-lane DP itself -- so the list only has to be tight, not proven complete.
    DpJob* jobs = (DpJob*)tv.jobs;
    for (int i = 0; i < 2; ++i) {
        pv.n_regs[rd[i]] = n[i];
        int32_t *cnt = 0, *has_alt = 0;
        if (!(opt.flag & MEM_F_ALL) && n[i] > 0) {
            cnt = zb[i]; has_alt = zb[i] + n[i];
            if (xa_prepare(opt, n[i], a[i], cnt, has_alt) == 0) cnt = has_alt = 0;
        }
        int alt_k = -1, which = -1;
        if (paired) {
            if (n_pri[i] < n[i]) {
                const AlnReg* p = &a[i][n_pri[i]];
                if (!(p->score < opt.T || p->secondary >= 0 || !p->is_alt)) alt_k = n_pri[i];
            }
        } else if (n[i]) {                                       // the region behind this end's mate summary
            if (a[i][0].score >= opt.T) which = 0;
            else if (n_pri[i] < n[i] && a[i][n_pri[i]].score >= opt.T) which = n_pri[i];
        }
        for (int j = 0; j < n[i]; ++j) {
            AlnReg* p = &a[i][j];
            p->pad_ = 0;
            const bool rec = paired ? (j == z[i] || j == alt_k) : (j == which || reg2sam_selects(opt, a[i], j));
            bool xa = false;
            if (cnt) {
                const int pr = get_pri_idx(opt.XA_drop_ratio, a[i], j);
                xa = pr >= 0 && (!paired || pr == z[i] || pr == alt_k) && !(cnt[pr] > opt.max_XA_hits_alt || (!has_alt[pr] && cnt[pr] > opt.max_XA_hits));
            }
            if ((rec || xa) && region_needs_dp(opt, *p)) {
                // (one slot per reservation: every slot below job_cap that was handed out gets written, so an overflow leaves no
                // unwritten slot behind -- unlike the several-slot reservations of k_pe_rescue_plan / k_rescore_plan)
                int job = atomicAdd(tv.job_cnt, 1);
                if (job < tv.job_cap) { DpJob jb; jb.read = rd[i]; jb.reg = j; jobs[job] = jb; p->pad_ = job + 1; }
                else err |= ERR_JOB_CAP;
            }
        }
    }
    PE_CLK(3);
#undef PE_CLK
    if (S.err | err) atomicOr(tv.err, S.err | err);
}

// mem_sam_pe, second half: the records of both mates.  One lane per read (lanes 2k and 2k+1 hold the mates of a pair):
// nothing at this stage writes to the regions, so a mate only needs the other's position summary (a `light` reg2aln of
// the other mate's chosen region, recomputed here rather than exchanged) and the two records are otherwise independent --
// twice the waves and half the dependent chain of a lane-per-pair layout, which is what this latency-bound stage wants
__global__ void __launch_bounds__(64, 6) k_pe_out(DevIndex ix, MemOpt opt, TileView tv, PeView pv, MemPestat p0, MemPestat p1, MemPestat p2, MemPestat p3, const PeState* states, JobView jvv)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= (tv.n_reads & ~1)) return;
    const int pi = r >> 1, i = r & 1, m = r ^ 1;
    const JobView* jv = &jvv;
    // own read / mate: plain pointers derived from the kernel arguments (no pointer arrays: a pointer reloaded from
    // private memory loses its address space and every access through it becomes a flat access -- on this
    // latency-bound stage that alone was a factor of three)
    const uint8_t* seq_i = tv.seq + tv.seq_off[r];
    const uint8_t* seq_m = tv.seq + tv.seq_off[m];
    const int l_i = (int)(tv.seq_off[r + 1] - tv.seq_off[r] - 1), l_m = (int)(tv.seq_off[m + 1] - tv.seq_off[m] - 1);
    const AlnReg* ai = pv.regs + pv.reg_off[r];
    const AlnReg* am = pv.regs + pv.reg_off[m];
    const int n_i = pv.n_regs[r], n_m = pv.n_regs[m];
    int32_t* zb_i = pv.ints + 2 * pv.reg_off[r];
    PostScratch S = post_scratch_for(tv, r);
    OutBuf ob;
    ob.p = tv.out + (size_t)r * tv.out_cap; ob.cap = tv.out_cap; ob.len = 0; ob.ovf = false;
    const PeState st = states[pi];
    const int z_i = i ? st.z1 : st.z0, z_m = i ? st.z0 : st.z1, npri_i = i ? st.n_pri1 : st.n_pri0, npri_m = i ? st.n_pri0 : st.n_pri1;
    const int q_se_i = i ? st.q_se1 : st.q_se0;
    int extra_flag = st.extra_flag;
    const bool paired = st.paired != 0;
    // Both outcomes of the pairing stage (the pair's records h/g, or each end through mem_reg2sam) run through the same
    // two loops below -- mate summary, then records -- so that a wave holding both kinds of pairs executes the per-base
    // work (reg2aln, aln2out) once, not once per kind.
    // 1. the mate's position summary (and, for the unpaired case, this end's contig for the proper-pair flag)
    MateInfo mate; mate.rid = -1; mate.pos = -1; mate.is_rev = 0; mate.ref_len = 0;
    int rid_i = -1, rid_m = -1;
    for (int k = 0; k < 2; ++k) {                               // k = 0: the mate, k = 1: this end
        if (paired && k == 1) break;                            // h of this end is written below
        const AlnReg* ak = k ? ai : am;
        const int nk = k ? n_i : n_m, nprik = k ? npri_i : npri_m;
        const AlnReg* sel = 0;
        if (paired) sel = &ak[z_m];
        else if (nk) {
            if (ak[0].score >= opt.T) sel = &ak[0];
            else if (nprik < nk && ak[nprik].score >= opt.T) sel = &ak[nprik];
        }
        AlnRec h = reg2aln(ix, opt, S, k ? l_i : l_m, k ? seq_i : seq_m, sel, jv, true);
        if (k) rid_i = h.rid; else { mate = mate_of(h); rid_m = h.rid; }
    }
    if (!paired) {
        if (!(opt.flag & MEM_F_NOPAIRING) && rid_i == rid_m && rid_i >= 0) {
            int64_t dist;
            const int64_t rb0 = i ? am[0].rb : ai[0].rb, rb1 = i ? ai[0].rb : am[0].rb;
            const int d = infer_dir(ix.l_pac, rb0, rb1, &dist);
            const MemPestat pd = d == 0 ? p0 : d == 1 ? p1 : d == 2 ? p2 : p3;
            if (!pd.failed && dist >= pd.low && dist <= pd.high) extra_flag |= 2;
        }
        extra_flag |= i ? 0x81 : 0x41;
    }
    // 2. which regions become records: h = z and the ALT supplementary g (paired), or mem_reg2sam's selection
    int alt_k = -1;
    if (paired && npri_i < n_i) {
        const AlnReg* p = &ai[npri_i];
        if (!(p->score < opt.T || p->secondary >= 0 || !p->is_alt)) alt_k = npri_i;
    }
    int32_t *cnt = 0, *has_alt = 0;
    if (!(opt.flag & MEM_F_ALL) && n_i > 0) {
        cnt = zb_i; has_alt = zb_i + n_i;
        if (xa_prepare(opt, n_i, ai, cnt, has_alt) == 0) cnt = has_alt = 0;
    }
    int n_aa = 0;
    for (int k = 0; k < n_i; ++k) n_aa += paired ? (k == z_i || k == alt_k) : reg2sam_selects(opt, ai, k);
    if (n_aa == 0) {                                            // unpaired only: h always exists for a paired end
        AlnRec t = reg2aln(ix, opt, S, l_i, seq_i, 0);
        t.flag |= extra_flag;
        aln2out(ix, opt, S, ob, 1, 0, t, &mate, l_i, seq_i, n_i, ai, 0, 0, -1);
    }
    int l = 0, mapq0 = 0;
    for (int k = 0; k < n_i && n_aa; ++k) {
        if (!(paired ? (k == z_i || k == alt_k) : reg2sam_selects(opt, ai, k))) continue;
        const AlnReg* p = &ai[k];
        AlnRec q = reg2aln(ix, opt, S, l_i, seq_i, p, jv);
        if (paired) {
            if (k == z_i) q.mapq = q_se_i; else q.flag |= 0x800;
            q.flag |= 0x40 << i | extra_flag;
        } else {
            q.flag |= extra_flag;
            if (p->secondary >= 0) q.sub = -1;
            if (l && p->secondary < 0) q.flag |= (opt.flag & MEM_F_NO_MULTI) ? 0x10000 : 0x800;
            if (l && !p->is_alt && q.mapq > mapq0) q.mapq = mapq0;
            if (l == 0) mapq0 = q.mapq;
        }
        aln2out(ix, opt, S, ob, n_aa, l, q, &mate, l_i, seq_i, n_i, ai, cnt, has_alt, cnt ? k : -1, jv);
        ++l;
    }
    tv.out_len[r] = ob.ovf ? 0 : ob.len;
    if (ob.ovf) atomicOr(tv.err, ERR_OUT_CAP);
    if (S.err) atomicOr(tv.err, S.err);
}

// an odd trailing read of a PE call is never processed upstream (n>>1 pairs): it produces no bytes
__global__ void k_pe_tail(TileView tv) { if (threadIdx.x == 0 && blockIdx.x == 0 && (tv.n_reads & 1)) tv.out_len[tv.n_reads - 1] = 0; }

// capacity of each read's region pool in phase 2: its own regions + what mate rescue may add
__global__ void k_pe_caps(MemOpt opt, TileView tv, int32_t* caps)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= tv.n_reads) return;
    int mate = r ^ 1;
    int nm = mate < tv.n_reads ? tv.n_regs[mate] : 0;
    caps[r] = tv.n_regs[r] + 4 * (nm < opt.max_matesw ? nm : opt.max_matesw) + 1;
}

// regions from the phase-1 pool (indexed by seed_off) into the phase-2 pool (indexed by reg_off)
__global__ void k_pe_copy_regs(TileView tv, const AlnReg* src, const int64_t* src_off, AlnReg* dst, const int64_t* dst_off, const int32_t* n_regs)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= tv.n_reads) return;
    for (int i = 0; i < n_regs[r]; ++i) dst[dst_off[r] + i] = src[src_off[r] + i];
}

void launch_pestat_cand(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, int8_t* cand_dir, int64_t* cand_is)
{
    int np = tv.n_reads >> 1;
    if (np <= 0) return;
    hipLaunchKernelGGL(k_pestat_cand, dim3((np + 63) / 64), dim3(64), 0, st, ix, opt, tv, cand_dir, cand_is);
}
void launch_pe_caps(hipStream_t st, const MemOpt& opt, const TileView& tv, int32_t* caps)
{
    if (tv.n_reads <= 0) return;
    hipLaunchKernelGGL(k_pe_caps, dim3((tv.n_reads + 255) / 256), dim3(256), 0, st, opt, tv, caps);
}
void launch_pe_copy_regs(hipStream_t st, const TileView& tv, const AlnReg* src, const int64_t* src_off, AlnReg* dst, const int64_t* dst_off, const int32_t* n_regs)
{
    if (tv.n_reads <= 0) return;
    hipLaunchKernelGGL(k_pe_copy_regs, dim3((tv.n_reads + 255) / 256), dim3(256), 0, st, tv, src, src_off, dst, dst_off, n_regs);
}
// mate rescue steps 1 and 2 + the pairing stage.  rescue: RescueJob[cap], KswR[cap], per-pair first/count, a counter (zeroed here)
void launch_pe_pair(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, AlnReg* regs, const int64_t* reg_off,
                    int32_t* n_regs, int32_t* ints, void* vpool, void* keys, uint8_t* scratch, int64_t scratch_per_pair, int cap_h, int cap_b, int cap_u, const MemPestat* pes, const PairTab& ptab, void* states,
                    void* rescue_jobs, void* rescue_res, int32_t* rescue_first, int32_t* rescue_num, int32_t* rescue_cnt, int rescue_cap)
{
    int np = tv.n_reads >> 1;
    hipLaunchKernelGGL(k_pe_tail, dim3(1), dim3(64), 0, st, tv);
    if (np <= 0) return;
    PeView pv; pv.regs = regs; pv.reg_off = reg_off; pv.n_regs = n_regs; pv.ints = ints; pv.vpool = vpool; pv.keys = keys; pv.scratch = scratch;
    pv.scratch_per_pair = scratch_per_pair; pv.cap_h = cap_h; pv.cap_b = cap_b; pv.cap_u = cap_u; pv.ptab = ptab;
    (void)hipMemsetAsync(rescue_cnt, 0, 8, st);                      // [0] rescue jobs, [1] heavy pairs
    int32_t* const heavy_list = rescue_num + (np + 1);               // behind the per-pair arrays (pipeline.cpp sizes the buffer)
    int32_t* const heavy_cnt = rescue_cnt + 1;
    const int wave_grid = np < 8192 ? np : 8192;
    if (!(opt.flag & MEM_F_NO_RESCUE)) {
        hipLaunchKernelGGL(k_pe_rescue_plan, dim3((np + 127) / 128), dim3(128), 0, st, ix, opt, tv, pv, pes[0], pes[1], pes[2], pes[3],
                           (RescueJob*)rescue_jobs, rescue_first, rescue_num, rescue_cnt, rescue_cap, heavy_list, heavy_cnt);
        hipLaunchKernelGGL(k_pe_rescue_plan_wave, dim3(wave_grid), dim3(64), 0, st, ix, opt, tv, pv, pes[0], pes[1], pes[2], pes[3],
                           (RescueJob*)rescue_jobs, rescue_first, rescue_num, rescue_cnt, rescue_cap, heavy_list, heavy_cnt);
        launch_sw_jobs(st, ix, opt, tv, rescue_jobs, rescue_cnt, rescue_cap, rescue_res, cap_b, tv.max_len);
        hipLaunchKernelGGL(k_pe_matesw_wave, dim3(wave_grid), dim3(64), 0, st, ix, opt, tv, pv, pes[0], pes[1], pes[2], pes[3],
                           (const RescueJob*)rescue_jobs, (const KswR*)rescue_res, (const int32_t*)rescue_first, rescue_num, heavy_list, heavy_cnt);
    }
    hipLaunchKernelGGL(k_pe_pair, dim3((np + 63) / 64), dim3(64), 0, st, ix, opt, tv, pv, pes[0], pes[1], pes[2], pes[3], (PeState*)states,
                       (const RescueJob*)rescue_jobs, (const KswR*)rescue_res, (const int32_t*)rescue_first, (const int32_t*)rescue_num);
}
// ksw_align2 for a list of jobs (mate rescue, seed re-scoring); results preset to RESCUE_NOT_RUN, cnt = device job count
void launch_sw_jobs(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, const void* jobs, const int32_t* cnt, int cap, void* results, int cap_b, int max_qlen)
{
    if (cap <= 0) return;
    (void)hipMemsetAsync(results, 0x81, pe_rescue_bytes(1, cap), st);
    // one instance per mode and range of segment counts (the stripes are unrolled to the instance's maximum).  Byte mode
    // (scores below 250: 150 bp mates), 16 bases per segment, four alignments per wave: up to 10, 11..16 segments.  16-bit mode
    // (seed re-scoring windows of long reads, 250 bp mates), 8 bases per segment, eight alignments per wave: up to 10, 11..16,
    // 17..25, 26..32 segments.
#define SW_LAUNCH(LO, HI, GW) hipLaunchKernelGGL((k_pe_rescue_sw<LO, HI, GW>), dim3((cap + 64 / GW - 1) / (64 / GW)), dim3(64), (size_t)(64 / GW) * cap_b * 8, st, ix, opt, tv, (const SwJob*)jobs, cnt, cap, (KswR*)results, cap_b)
    // two alignments per lane in packed halves (k_pe_rescue_sw2): byte mode up to 10 segments (150 bp mates), 16-bit mode up to 16 and 17..25
    // segments (seed re-scoring windows, 250 bp mates).  BWAMEM_HIP_SW_PACKED=0: everything through the one-value-per-register form
    static const bool packed = []{ const char* e = getenv("BWAMEM_HIP_SW_PACKED"); return !(e && atoi(e) == 0); }();
#define SW2_LAUNCH(LO, HI, U8) hipLaunchKernelGGL((k_pe_rescue_sw2<LO, HI, U8>), dim3((cap + (U8 ? 8 : 16) - 1) / (U8 ? 8 : 16)), dim3(64), (size_t)(U8 ? 8 : 16) * cap_b * 8, st, ix, opt, tv, (const SwJob*)jobs, cnt, cap, (KswR*)results, cap_b)
    if (packed) SW2_LAUNCH(0, 10, true); else SW_LAUNCH(0, 10, 16);
    if (max_qlen > 160) SW_LAUNCH(10, 16, 16);
    if (packed) {
        SW2_LAUNCH(0, 16, false);
        if (max_qlen > 128) SW2_LAUNCH(16, 25, false);
    } else {
        SW_LAUNCH(0, 10, 8);
        if (max_qlen > 80) SW_LAUNCH(10, 16, 8);
        if (max_qlen > 128) SW_LAUNCH(16, 25, 8);
    }
    if (max_qlen > 200) SW_LAUNCH(25, 32, 8);
#undef SW2_LAUNCH
#undef SW_LAUNCH
}
size_t pe_rescue_bytes(int what, int cap) { return what == 0 ? (size_t)cap * sizeof(RescueJob) : (size_t)cap * sizeof(KswR); }
void launch_pe_out(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, AlnReg* regs, const int64_t* reg_off,
                   int32_t* n_regs, int32_t* ints, const MemPestat* pes, const void* states, const void* job_out, const uint32_t* job_cig, int cig_cap)
{
    int np = tv.n_reads >> 1;
    if (np <= 0) return;
    PeView pv = PeView();
    pv.regs = regs; pv.reg_off = reg_off; pv.n_regs = n_regs; pv.ints = ints;
    JobView jv; jv.out = (const DpOut*)job_out; jv.cig = job_cig; jv.cig_cap = cig_cap;
    hipLaunchKernelGGL(k_pe_out, dim3((2 * np + 63) / 64), dim3(64), 0, st, ix, opt, tv, pv, pes[0], pes[1], pes[2], pes[3], (const PeState*)states, jv);
}
size_t pe_state_bytes(int n_reads) { return (size_t)((n_reads >> 1) + 1) * sizeof(PeState); }
