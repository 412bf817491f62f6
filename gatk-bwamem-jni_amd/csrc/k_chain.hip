// k_chain.hip -- seed chaining and chain filtering, one lane per read.
//
// Replaces, for the reference call at jnibwa.c:214, upstream bwamem.c mem_chain (the
// kbtree-keyed greedy chaining + test_and_merge), mem_chain_weight and mem_chain_flt
// (SURVEY.md rows a8, a9).  The B-tree is kept (t = 5, keys = chain ids ordered by chain.pos)
// because its insertion rule decides the order of chains with equal keys, and that order
// feeds the unstable weight sort downstream.  Chains hold their seeds as a
// linked list threaded through the occurrence array, so no per-chain allocation exists.
#include <math.h>
#include "dev_common.h"
#include "kernels.h"
#include "sw_common.h"
#include "wave_ops.h"
#include "chain_flt.h"

#define BT_T 5
#define BT_MAXK (2 * BT_T - 1)
// BT_NODE_INTS (bwamem_types.h) = 40: is_internal, n, key[9], ptr[10], pad, pos[9] (the keys' chain positions, 2 ints each: a comparison reads the node only)

struct BTree {
    int32_t* pool; int n_nodes, cap_nodes, root, n_keys;
    const Chain* cs;         // key -> chain (creation order)
    bool ovf;
    __device__ int32_t* node(int i) const { return pool + (size_t)i * BT_NODE_INTS; }
    __device__ int alloc(int is_internal) {
        if (n_nodes >= cap_nodes) { ovf = true; return 0; }
        int32_t* x = node(n_nodes);
        for (int i = 0; i < BT_NODE_INTS; ++i) x[i] = 0;
        x[0] = is_internal;
        return n_nodes++;
    }
};
#define BT_N(x) ((x)[1])
#define BT_KEY(x) ((x) + 2)
#define BT_PTR(x) ((x) + 11)
#define BT_POS(x) ((int64_t*)((x) + 22))

DEV int bt_cmp_pos(int64_t a, int64_t b) { return (b < a) - (a < b); }

// index of the last key < k (or the first key == k); *r = sign(k - key[idx])
DEV int bt_getp_aux(const BTree& b, const int32_t* x, int64_t kpos, int* r)
{
    int tr, *rr = r ? r : &tr, begin = 0, end = BT_N(x);
    if (BT_N(x) == 0) return -1;
    while (begin < end) {
        int mid = (begin + end) >> 1;
        if (bt_cmp_pos(BT_POS(x)[mid], kpos) < 0) begin = mid + 1;
        else end = mid;
    }
    if (begin == BT_N(x)) { *rr = 1; return BT_N(x) - 1; }
    if ((*rr = bt_cmp_pos(kpos, BT_POS(x)[begin])) < 0) --begin;
    return begin;
}

// chain with the largest pos <= kpos along the search path (upstream kb_intervalp's "lower")
DEV int bt_lower(const BTree& b, int64_t kpos)
{
    int lower = -1, r = 0;
    int xi = b.root;
    for (;;) {
        const int32_t* x = b.node(xi);
        int i = bt_getp_aux(b, x, kpos, &r);
        if (i >= 0 && r == 0) return BT_KEY(x)[i];
        if (i >= 0) lower = BT_KEY(x)[i];
        if (x[0] == 0) return lower;
        xi = BT_PTR(x)[i + 1];
    }
}

DEV void bt_split(BTree& b, int xi, int i, int yi)
{
    int zi = b.alloc(b.node(yi)[0]);
    if (b.ovf) return;
    int32_t *x = b.node(xi), *y = b.node(yi), *z = b.node(zi);
    BT_N(z) = BT_T - 1;
    for (int j = 0; j < BT_T - 1; ++j) { BT_KEY(z)[j] = BT_KEY(y)[j + BT_T]; BT_POS(z)[j] = BT_POS(y)[j + BT_T]; }
    if (y[0]) for (int j = 0; j < BT_T; ++j) BT_PTR(z)[j] = BT_PTR(y)[j + BT_T];
    BT_N(y) = BT_T - 1;
    for (int j = BT_N(x); j > i; --j) BT_PTR(x)[j + 1] = BT_PTR(x)[j];
    BT_PTR(x)[i + 1] = zi;
    for (int j = BT_N(x) - 1; j >= i; --j) { BT_KEY(x)[j + 1] = BT_KEY(x)[j]; BT_POS(x)[j + 1] = BT_POS(x)[j]; }
    BT_KEY(x)[i] = BT_KEY(y)[BT_T - 1]; BT_POS(x)[i] = BT_POS(y)[BT_T - 1];
    ++BT_N(x);
}

DEV void bt_put(BTree& b, int key)
{
    int64_t kpos = b.cs[key].pos;
    ++b.n_keys;
    int ri = b.root;
    if (BT_N(b.node(ri)) == BT_MAXK) {
        int si = b.alloc(1);
        if (b.ovf) return;
        b.root = si;
        BT_PTR(b.node(si))[0] = ri;
        bt_split(b, si, 0, ri);
        if (b.ovf) return;
        ri = si;
    }
    int xi = ri;
    for (;;) {
        int32_t* x = b.node(xi);
        if (x[0] == 0) {
            int i = bt_getp_aux(b, x, kpos, 0);
            for (int j = BT_N(x) - 1; j > i; --j) { BT_KEY(x)[j + 1] = BT_KEY(x)[j]; BT_POS(x)[j + 1] = BT_POS(x)[j]; }
            BT_KEY(x)[i + 1] = key; BT_POS(x)[i + 1] = kpos;
            ++BT_N(x);
            return;
        }
        int i = bt_getp_aux(b, x, kpos, 0) + 1;
        if (BT_N(b.node(BT_PTR(x)[i])) == BT_MAXK) {
            bt_split(b, xi, i, BT_PTR(x)[i]);
            if (b.ovf) return;
            x = b.node(xi);
            if (bt_cmp_pos(kpos, BT_POS(x)[i]) > 0) ++i;
        }
        xi = BT_PTR(x)[i];
    }
}

// in-order traversal into out[]; returns the number of keys written
DEV int bt_traverse(const BTree& b, Chain* out)
{
    int st_node[24], st_idx[24], sp = 0, n = 0;
    st_node[0] = b.root; st_idx[0] = 0; sp = 1;
    while (sp > 0) {
        const int32_t* x = b.node(st_node[sp - 1]);
        int i = st_idx[sp - 1];
        if (x[0] == 0) {
            for (int j = 0; j < BT_N(x); ++j) out[n++] = b.cs[BT_KEY(x)[j]];
            --sp;
            continue;
        }
        if (i > BT_N(x)) { --sp; continue; }
        if (i > 0) out[n++] = b.cs[BT_KEY(x)[i - 1]];   // key between child i-1 and child i
        st_idx[sp - 1] = i + 1;
        if (sp >= 24) return -1;
        st_node[sp] = BT_PTR(x)[i]; st_idx[sp] = 0; ++sp;
    }
    return n;
}

DEV int test_and_merge(const MemOpt& opt, int64_t l_pac, Chain& c, Seed* seeds, int si, int seed_rid)
{
    const Seed p = seeds[si];
    const Seed first = seeds[c.seed0], last = seeds[c.last];
    int64_t qend = last.qbeg + last.len, rend = last.rbeg + last.len;
    if (seed_rid != c.rid) return 0;
    if (p.qbeg >= first.qbeg && p.qbeg + p.len <= qend && p.rbeg >= first.rbeg && p.rbeg + p.len <= rend)
        return 1;                                    // contained: absorbed, not stored
    if ((last.rbeg < l_pac || first.rbeg < l_pac) && p.rbeg >= l_pac) return 0;
    int64_t x = p.qbeg - last.qbeg, y = p.rbeg - last.rbeg;
    if (y >= 0 && x - y <= opt.w && y - x <= opt.w && x - last.len < opt.max_chain_gap && y - last.len < opt.max_chain_gap) {
        seeds[c.last].next = si;
        c.last = si;
        ++c.n;
        return 1;
    }
    return 0;
}

DEV int chain_weight(const Chain& c_, const Seed* seeds)
{
    struct { int n, seed0; } c; c.n = c_.n; c.seed0 = c_.seed0;      // (read once: the loops below would otherwise reload them through the reference)
    int64_t end = 0;
    int w = 0, tmp, j, si;
    for (j = 0, si = c.seed0; j < c.n; ++j, si = seeds[si].next) {
        const Seed s = seeds[si];
        if (s.qbeg >= end) w += s.len;
        else if (s.qbeg + s.len > end) w += (int)(s.qbeg + s.len - end);
        end = end > s.qbeg + s.len ? end : s.qbeg + s.len;
    }
    tmp = w; w = 0;
    for (j = 0, end = 0, si = c.seed0; j < c.n; ++j, si = seeds[si].next) {
        const Seed s = seeds[si];
        if (s.rbeg >= end) w += s.len;
        else if (s.rbeg + s.len > end) w += (int)(s.rbeg + s.len - end);
        end = end > s.rbeg + s.len ? end : s.rbeg + s.len;
    }
    w = w < tmp ? w : tmp;
    return w < 1 << 30 ? w : (1 << 30) - 1;
}

struct ChainWLt { __device__ bool operator()(const Chain& a, const Chain& b) const { return a.w > b.w; } };
// The weight sort through 8-byte records {weight, index} (chain weights tie all the time and the introsort is unstable: the
// comparator sees the weight only, so the records take the very permutation the 40-byte chains would, and each chain then
// moves once).  recs: n records of scratch.
struct ChainRec { uint32_t w; int32_t idx; };
struct ChainRecLt { __device__ bool operator()(const ChainRec& a, const ChainRec& b) const { return a.w > b.w; } };
DEV void sort_chains_by_weight(int n, Chain* a, ChainRec* recs)
{
    if (n < 12) { ks_introsort((size_t)n, a, ChainWLt()); return; }
    for (int i = 0; i < n; ++i) { ChainRec k; k.w = a[i].w; k.idx = i; recs[i] = k; }
    ks_introsort((size_t)n, recs, ChainRecLt());
    for (int i = 0; i < n; ++i) {
        int src = recs[i].idx;
        if (src == i) continue;
        const Chain first = a[i];
        int j = i;
        while (src != i) { a[j] = a[src]; recs[j].idx = j; j = src; src = recs[j].idx; }
        a[j] = first; recs[j].idx = j;
    }
}

// chaining + chain weights + the weight sort of one read (one lane).  Returns the number of chains that enter mem_chain_flt's
// overlap loop (sorted in a[]), 0 when the read has none.  kept: room for the packed kept chains, in the B-tree's node pool
// (free since the traversal; 40 bytes per seed).
DEV int chain_build(const DevIndex& ix, const MemOpt& opt, const TileView& tv, Chain* chain_store, int r, int64_t s0, int64_t s1, int64_t node0)
{
    int len = (int)(tv.seq_off[r + 1] - tv.seq_off[r] - 1);
    Seed* seeds = tv.seeds;              // tile-global indices
    Chain* cs = chain_store + s0;        // creation order
    Chain* a = tv.chains + s0;           // B-tree order, then filtered
    int n_cs = 0;

    BTree bt;
    bt.pool = tv.bt_nodes + node0 * BT_NODE_INTS;
    bt.cap_nodes = (int)((s1 / 4 + 3 * (int64_t)(r + 1)) - node0);
    bt.n_nodes = 0; bt.n_keys = 0; bt.cs = cs; bt.ovf = false;
    bt.root = bt.alloc(0);

    for (int64_t g = s0; g < s1; ++g) {
        int rid = tv.seed_rid[g];
        if (rid < 0) continue;
        int to_add = 0;
        if (bt.n_keys) {
            int lower = bt_lower(bt, seeds[g].rbeg);
            if (lower < 0 || !test_and_merge(opt, ix.l_pac, cs[lower], seeds, (int)g, rid)) to_add = 1;
        } else to_add = 1;
        if (to_add) {
            Chain c;
            c.pos = seeds[g].rbeg; c.n = 1; c.first = -1; c.rid = rid; c.w = 0; c.kept = 0;
            c.is_alt = ix.ann_is_alt[rid] != 0;
            c.seed0 = (int)g; c.last = (int)g; c.frac_rep = 0.f; c.pad_ = 0;
            cs[n_cs] = c;
            bt_put(bt, n_cs);
            ++n_cs;
            if (bt.ovf) { atomicOr(tv.err, ERR_BTREE); return 0; }
        }
    }
    int n_chn = bt_traverse(bt, a);
    if (n_chn < 0) { atomicOr(tv.err, ERR_BTREE); return 0; }
    float frac_rep = (float)tv.l_rep[r] / len;
    for (int i = 0; i < n_chn; ++i) a[i].frac_rep = frac_rep;

    // ---- mem_chain_flt (row a9): weights, weight sort
    int k = 0;
    for (int i = 0; i < n_chn; ++i) {
        Chain c = a[i];
        c.first = -1; c.kept = 0;
        c.w = (uint32_t)chain_weight(c, seeds);
        if ((int)c.w >= opt.min_chain_weight) a[k++] = c;
    }
    n_chn = k;
    if (n_chn == 0) return 0;
    const int kept_cap = (int)(((size_t)bt.cap_nodes * BT_NODE_INTS * 4 - 16) / 16);
    sort_chains_by_weight(n_chn, a, (ChainRec*)cs);          // cs (the chains in creation order) is dead since the traversal
    if (n_chn > kept_cap) { atomicOr(tv.err, ERR_BTREE); return 0; }
    return n_chn;
}

#define CHAIN_FLT_WAVE_MIN 48       // chains: from here on a read's overlap loop is run by the wavefront

__global__ void __launch_bounds__(64, 8) k_chain(DevIndex ix, MemOpt opt, TileView tv, Chain* chain_store)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    Seed* seeds = tv.seeds;
    int64_t s0 = 0, s1 = 0, node0 = 0;
    int n_chn = 0, n_kept = 0;
    if (r < tv.n_reads) {
        tv.n_chains[r] = 0;
        s0 = tv.seed_off[r]; s1 = tv.seed_off[r + 1];
        node0 = s0 / 4 + 3 * (int64_t)r;
        if (s1 > s0) n_chn = chain_build(ix, opt, tv, chain_store, r, s0, s1, node0);
    }
    Chain* a = tv.chains + s0;
    // (aligned by pointer arithmetic: through an integer the pointer would lose its address space and every access become a flat one)
    int32_t* const kept_raw = tv.bt_nodes + node0 * BT_NODE_INTS;
    int4* kept = (int4*)((char*)kept_raw + ((16 - (int)((uintptr_t)kept_raw & 15)) & 15));

    // ---- mem_chain_flt's overlap loop: quadratic in the chains of a read, so reads with many chains get the wavefront, one
    // after the other, and the rest a lane each
    const bool by_wave = n_chn >= CHAIN_FLT_WAVE_MIN && !(tv.debug & 0x200);          // BWAMEM_HIP_DEBUGK=512: lane form only (tests)
    if (n_chn > 0 && !by_wave) n_kept = chain_flt_lane(opt, seeds, a, n_chn, kept);
    for (uint64_t heavy = __ballot(by_wave); heavy; heavy &= heavy - 1) {
        const int src = __ffsll((long long)heavy) - 1;
        const int64_t s0w = __shfl(s0, src), node0w = __shfl(node0, src);
        const int nw = __shfl(n_chn, src);
        int32_t* const kw = tv.bt_nodes + node0w * BT_NODE_INTS;
        const int got = chain_flt_wave(opt, seeds, tv.chains + s0w, nw, (int4*)((char*)kw + ((16 - (int)((uintptr_t)kw & 15)) & 15)));
        if (lane == src) n_kept = got;
    }
    if (n_chn == 0) return;

    {
        int i, k;
        for (i = 0; i < n_kept; ++i) {
            int f = kept[i].w;
            if (f >= 0) a[f].kept = 1;
        }
        for (i = k = 0; i < n_chn; ++i) {
            if (a[i].kept == 0 || a[i].kept == 3) continue;
            if (++k >= opt.max_chain_extend) break;
        }
        for (; i < n_chn; ++i)
            if (a[i].kept < 3) a[i].kept = 0;
        for (i = k = 0; i < n_chn; ++i)
            if (a[i].kept != 0) a[k++] = a[i];
        n_chn = k;
    }

    // ---- lay the kept chains' seeds out contiguously, chain by chain
    {
        Seed* out = tv.cseeds;
        int64_t o = s0;
        for (int i = 0; i < n_chn; ++i) {
            int si = a[i].seed0;
            a[i].seed0 = (int32_t)(o - s0);
            for (int j = 0; j < a[i].n; ++j) {
                Seed s = seeds[si];
                si = s.next;
                s.next = -1;
                out[o++] = s;
            }
        }
    }
    tv.n_chains[r] = n_chn;
}

// mem_flt_chained_seeds + mem_seed_sw (row a10): for reads long enough that 5.5 ln L <= 0.05 L, every short seed is
// re-scored by a local SW in a +-50 bp window and dropped when it scores below the HSP threshold.  The alignments are
// independent of one another, so a lane per read lists them (k_rescore_plan), the wave SW kernel of mate rescue runs
// them four per wavefront (k_pe.hip), and a lane per read applies the scores (k_rescore_apply).  Launched only for
// tiles that contain such reads.
#define MEM_SHORT_EXT 50
#define MEM_SHORT_LEN 200

// window of mem_seed_sw; false: the seed keeps its exact-match score (upstream returns -1)
DEV bool seed_sw_window(const DevIndex& ix, int l_query, const Seed& s, int& qb, int& qe, int64_t& rb, int64_t& re)
{
    const int64_t l_pac = ix.l_pac;
    if (s.len >= MEM_SHORT_LEN) return false;
    int rid;
    qb = s.qbeg; qe = s.qbeg + s.len;
    rb = s.rbeg; re = s.rbeg + s.len;
    const int64_t mid = (rb + re) >> 1;
    qb -= MEM_SHORT_EXT; qb = qb > 0 ? qb : 0;
    qe += MEM_SHORT_EXT; qe = qe < l_query ? qe : l_query;
    rb -= MEM_SHORT_EXT; rb = rb > 0 ? rb : 0;
    re += MEM_SHORT_EXT; re = re < l_pac << 1 ? re : l_pac << 1;
    if (rb < l_pac && l_pac < re) { if (mid < l_pac) re = l_pac; else rb = l_pac; }
    if (qe - qb >= MEM_SHORT_LEN || re - rb >= MEM_SHORT_LEN) return false;
    bns_clamp(ix, rb, mid, re, rid);
    return true;
}

// does this read go through mem_flt_chained_seeds, and with which threshold?
DEV bool rescore_read(const DevIndex& ix, const MemOpt& opt, const TileView& tv, int r, int& l_query, int& min_HSP_score)
{
    l_query = (int)(tv.seq_off[r + 1] - tv.seq_off[r] - 1);
    if (tv.n_chains[r] == 0) return false;
    if (l_query >= ix.log_tab_n) { atomicOr(tv.err, ERR_SCRATCH); return false; }
    const double min_l = opt.min_chain_weight ? 1.1f * opt.min_chain_weight : 5.5f * ix.log_tab[l_query];
    min_HSP_score = (int)(opt.a * min_l + .499);
    return !(min_l > 0.05f * l_query);               // short reads: nothing to do
}

__global__ void k_rescore_plan(DevIndex ix, MemOpt opt, TileView tv, SwJob* jobs, int32_t* first_num, int32_t* counter, int cap)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= tv.n_reads) return;
    int l_query, min_HSP_score, first = 0, cnt = 0;
    if (rescore_read(ix, opt, tv, r, l_query, min_HSP_score)) {
        const Chain* chains = tv.chains + tv.seed_off[r];
        const int n_chn = tv.n_chains[r];
        for (int pass = 0; pass < 2; ++pass) {             // count, reserve, write
            int k = 0;
            for (int i = 0; i < n_chn; ++i) {
                const Chain c = chains[i];
                const Seed* seeds = tv.cseeds + tv.seed_off[r] + c.seed0;
                for (int j = 0; j < c.n; ++j) {
                    int qb, qe; int64_t rb, re;
                    if (!seed_sw_window(ix, l_query, seeds[j], qb, qe, rb, re)) continue;
                    if (pass == 1 && k < cnt) {
                        // ksw_align2(..., KSW_XSTART) in 16-bit mode; only the score of the first pass is used, so the job asks for that pass alone
                        SwJob jb; jb.rb = rb; jb.read = r; jb.tag = k; jb.l_ms = qe - qb; jb.is_rev = 0; jb.tlen = (int)(re - rb); jb.xtra = 0; jb.q_off = qb; jb.pad_ = 0;
                        jobs[first + k] = jb;
                    }
                    ++k;
                }
            }
            if (pass == 0) {
                cnt = k;
                if (cnt > 0) {
                    // a reservation of several slots that straddles cap leaves slots below cap unwritten: flag the list as void
                    // (ERR_RESCUE_CAP: k_pe_rescue_sw and k_rescore_apply return at once, the host re-runs the tile with room)
                    first = atomicAdd(counter, cnt);
                    if (first + cnt > cap) { atomicOr(tv.err, ERR_RESCUE_CAP); cnt = 0; }
                }
                if (cnt == 0) break;
            }
        }
    }
    first_num[r] = first; first_num[tv.n_reads + r] = cnt;
}

__global__ void k_rescore_apply(DevIndex ix, MemOpt opt, TileView tv, const KswR* results, const int32_t* first_num)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= tv.n_reads) return;
    int l_query, min_HSP_score;
    if (tv.err[0] & ERR_RESCUE_CAP) return;                 // the job list did not fit (k_rescore_plan): this attempt is void
    if (!rescore_read(ix, opt, tv, r, l_query, min_HSP_score)) return;
    const uint8_t* query = tv.seq + tv.seq_off[r];
    int q = first_num[r];
    const int q_end = q + first_num[tv.n_reads + r];
    int32_t hbuf[4 * (MEM_SHORT_LEN + 16)];               // (only used when a job was not run: list overflow, retried by the host)
    SwScratch W; W.cap_h = MEM_SHORT_LEN + 16; W.cap_b = 0; W.b = 0;
    W.H0 = hbuf; W.H1 = hbuf + W.cap_h; W.E = hbuf + 2 * W.cap_h; W.Hmax = hbuf + 3 * W.cap_h;
    Chain* chains = tv.chains + tv.seed_off[r];
    const int n_chn = tv.n_chains[r];
    int err = 0;
    for (int i = 0; i < n_chn; ++i) {
        Chain& c = chains[i];
        Seed* seeds = tv.cseeds + tv.seed_off[r] + c.seed0;
        int k = 0;
        for (int j = 0; j < c.n; ++j) {
            Seed s = seeds[j];
            int qb, qe; int64_t rb, re;
            int sc = -1;
            if (seed_sw_window(ix, l_query, s, qb, qe, rb, re)) {
                sc = q < q_end ? results[q].score : (int)0x81818181;
                ++q;
                if (sc == (int)0x81818181) {
                    SwIn I; I.ms = query + qb; I.l_ms = qe - qb; I.is_rev = 0; I.qrev = 0; I.t0 = rb; I.trev = 0;
                    sc = sw_core(ix, opt, I, 2, qe - qb, (int)(re - rb), KSW_XSTART, W, err).score;
                }
            }
            s.score = sc;
            if (s.score < 0 || s.score >= min_HSP_score) {
                s.score = s.score < 0 ? s.len * opt.a : s.score;
                seeds[k++] = s;
            }
        }
        c.n = k;
    }
    if (err) atomicOr(tv.err, err);
}

bool rescore_needed(const MemOpt& opt, const TileView& tv)
{   // the shortest read that can trigger re-scoring has 5.5 ln L <= 0.05 L
    double L = tv.max_len > 1 ? (double)tv.max_len : 2.0;
    double min_l = opt.min_chain_weight ? 1.1f * opt.min_chain_weight : 5.5f * log(L);
    return tv.n_reads > 0 && !(min_l > 0.05f * L);
}

void launch_rescore(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, void* jobs, void* results, int32_t* first_num, int32_t* cnt, int cap)
{
    if (!rescore_needed(opt, tv)) return;
    (void)hipMemsetAsync(cnt, 0, 4, st);
    hipLaunchKernelGGL(k_rescore_plan, dim3((tv.n_reads + 63) / 64), dim3(64), 0, st, ix, opt, tv, (SwJob*)jobs, first_num, cnt, cap);
    launch_sw_jobs(st, ix, opt, tv, jobs, cnt, cap, results, 16, MEM_SHORT_LEN);
    hipLaunchKernelGGL(k_rescore_apply, dim3((tv.n_reads + 63) / 64), dim3(64), 0, st, ix, opt, tv, (const KswR*)results, (const int32_t*)first_num);
}

void launch_chain(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, Chain* chain_store)
{
    if (tv.n_reads <= 0) return;
    hipLaunchKernelGGL(k_chain, dim3((tv.n_reads + 63) / 64), dim3(64), 0, st, ix, opt, tv, chain_store);
}
