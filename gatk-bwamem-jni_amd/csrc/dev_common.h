// dev_common.h -- device-side building blocks shared by the kernels: FM-index rank
// queries on the 32-byte occ blocks, bidirectional interval extension, sampled-SA
// lookup, packed-reference access, contig lookup, and the order-exact introsort.
//
// Behavioural contract: upstream lh3/bwa bwt.c / bntseq.c / ksort.h as reached from the
// reference at jnibwa.c:214 (App. B).
#pragma once
#include <hip/hip_runtime.h>
#include "bwamem_types.h"

#define DEV static __device__ inline

// A pointer whose origin the compiler cannot see (chosen at run time between LDS and global memory, or read back from a struct
// passed by reference to a function that is not inlined) makes every access a *flat* one: slower, and counted on both the
// vector-memory and the LDS counter, so the next LDS read waits for every global store still in flight.  Where the code knows
// the space it says so.  (Device pass only: the host pass of hipcc and the g++ build of tests/emu see plain pointers.)
#if defined(__HIP_DEVICE_COMPILE__)
#define AS_GLOBAL(T, p) ((T __attribute__((address_space(1)))*)(p))
#define AS_LDS(T, p) ((T __attribute__((address_space(3)))*)(p))
#else
#define AS_GLOBAL(T, p) ((T*)(p))
#define AS_LDS(T, p) ((T*)(p))
#endif
// a load that does not trust the CU's vector cache: for words another lane of the workgroup has just stored (after a barrier)
#if defined(__HIP_DEVICE_COMPILE__)
DEV uint32_t load_fresh(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
DEV uint32_t load_fresh(const uint32_t* p) { return *p; }
#endif

// ---------------------------------------------------------------- rank queries (row a2)
// counts of A,C,G,T among the first n (0..16) symbols of a 16-symbol word (MSB first)
DEV void cnt_word(uint32_t x, int n, uint32_t& c1, uint32_t& c2, uint32_t& c3)
{
    uint32_t m = n >= 16 ? 0x55555555u : (n <= 0 ? 0u : (0x55555555u & ~((1u << (32 - 2 * n)) - 1u)));
    uint32_t lo = x & m, hi = (x >> 1) & m;
    c3 += __popc(hi & lo);
    c2 += __popc(hi & ~lo);
    c1 += __popc(~hi & lo);
}

// absolute counts of A,C,G,T before block b from its 16-byte count word: the low 32 bits of C, G, T in x, y, z and
// bits 32..39 of each in the three low bytes of w (a 64-bit value is a register pair, so the low halves need no work)
DEV void occ_unpack(const uint4& c, uint64_t b, uint64_t& a0, uint64_t& a1, uint64_t& a2, uint64_t& a3)
{
    a1 = (uint64_t)(c.w & 0xffu) << 32 | c.x;
    a2 = (uint64_t)(c.w >> 8 & 0xffu) << 32 | c.y;
    a3 = (uint64_t)(c.w >> 16 & 0xffu) << 32 | c.z;
    a0 = (b << 6) - a1 - a2 - a3;
}

// occ(k, .) for all four symbols: # of c in BWT$[0..k] (k in sentinel-inclusive coordinates)
DEV void occ4(const DevIndex& ix, uint64_t k, uint64_t cnt[4])
{
    if (k == (uint64_t)-1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
    k -= (k >= ix.primary);
    const uint64_t b = k >> 6;
    const uint4* p = ix.occ + 2 * b;                           // one 32-byte block
    const uint4 c = p[0], s = p[1];
    const int n = (int)(k & 63) + 1;                           // symbols of this block to count
    uint32_t c1 = 0, c2 = 0, c3 = 0;
    cnt_word(s.x, n, c1, c2, c3);       cnt_word(s.y, n - 16, c1, c2, c3);
    cnt_word(s.z, n - 32, c1, c2, c3);  cnt_word(s.w, n - 48, c1, c2, c3);
    uint64_t a0, a1, a2, a3;
    occ_unpack(c, b, a0, a1, a2, a3);
    cnt[0] = a0 + (uint32_t)(n - (int)(c1 + c2 + c3));
    cnt[1] = a1 + c1;
    cnt[2] = a2 + c2;
    cnt[3] = a3 + c3;
}

// occ(k,.) and occ(l,.) for k <= l.  When both ranks fall into the same 64-symbol block (the common case once an
// interval is small) the block is fetched once -- upstream's bwt_2occ4 makes the same distinction.
DEV void occ4_pair(const DevIndex& ix, uint64_t k, uint64_t l, uint64_t tk[4], uint64_t tl[4])
{
    const uint64_t kk = k - (k >= ix.primary), ll = l - (l >= ix.primary);
    if (k == (uint64_t)-1 || l == (uint64_t)-1 || (kk >> 6) != (ll >> 6)) { occ4(ix, k, tk); occ4(ix, l, tl); return; }
    const uint64_t b = kk >> 6;
    const uint4* p = ix.occ + 2 * b;
    const uint4 c = p[0], s = p[1];
    const int nk = (int)(kk & 63) + 1, nl = (int)(ll & 63) + 1;
    uint32_t a1 = 0, a2 = 0, a3 = 0, b1 = 0, b2 = 0, b3 = 0;
    cnt_word(s.x, nk, a1, a2, a3);       cnt_word(s.x, nl, b1, b2, b3);
    cnt_word(s.y, nk - 16, a1, a2, a3);  cnt_word(s.y, nl - 16, b1, b2, b3);
    cnt_word(s.z, nk - 32, a1, a2, a3);  cnt_word(s.z, nl - 32, b1, b2, b3);
    cnt_word(s.w, nk - 48, a1, a2, a3);  cnt_word(s.w, nl - 48, b1, b2, b3);
    uint64_t c0, c1, c2, c3;
    occ_unpack(c, b, c0, c1, c2, c3);
    tk[0] = c0 + (uint32_t)(nk - (int)(a1 + a2 + a3)); tk[1] = c1 + a1; tk[2] = c2 + a2; tk[3] = c3 + a3;
    tl[0] = c0 + (uint32_t)(nl - (int)(b1 + b2 + b3)); tl[1] = c1 + b1; tl[2] = c2 + b2; tl[3] = c3 + b3;
}

// one interval of a bidirectional extension, kept in registers (no runtime-indexed ok[4] array):
// is_back = 1: the interval of bP for b = c; is_back = 0: upstream's ok[c] of bwt_extend(..., 0), i.e. the
// interval of P followed by base 3 - c.
DEV Intv extend_one(const DevIndex& ix, const Intv& ik, int c, int is_back)
{
    const uint64_t xa = is_back ? ik.x0 : ik.x1, xb = is_back ? ik.x1 : ik.x0;   // xa: the side that is rank-extended
    uint64_t tk[4], tl[4];
    occ4_pair(ix, xa - 1, xa - 1 + ik.size, tk, tl);
    const uint64_t s0 = tl[0] - tk[0], s1 = tl[1] - tk[1], s2 = tl[2] - tk[2], s3 = tl[3] - tk[3];
    const uint64_t tkc = c == 0 ? tk[0] : c == 1 ? tk[1] : c == 2 ? tk[2] : tk[3];
    const uint64_t sc = c == 0 ? s0 : c == 1 ? s1 : c == 2 ? s2 : s3;
    const uint64_t l2c = c == 0 ? ix.L2[0] : c == 1 ? ix.L2[1] : c == 2 ? ix.L2[2] : ix.L2[3];
    uint64_t other = xb + (xa <= ix.primary && xa + ik.size - 1 >= ix.primary);
    other += (c < 3 ? s3 : 0) + (c < 2 ? s2 : 0) + (c < 1 ? s1 : 0);
    Intv r;
    const uint64_t na = l2c + 1 + tkc;
    r.x0 = is_back ? na : other;
    r.x1 = is_back ? other : na;
    r.size = sc;
    r.info = ik.info;
    return r;
}

DEV void set_intv(const DevIndex& ix, int c, Intv& ik)
{
    ik.x0 = ix.L2[c] + 1;
    ik.size = ix.L2[c + 1] - ix.L2[c];
    ik.x1 = ix.L2[3 - c] + 1;
    ik.info = 0;
}

// bwt_invPsi: the rank of the suffix that starts one text position before the suffix of rank k
DEV uint64_t lf_step(const DevIndex& ix, uint64_t k)
{
    if (k == ix.primary) return 0;
    const uint64_t x = k - (k > ix.primary), b = x >> 6;
    const uint4* p = ix.occ + 2 * b;
    const uint4 cv = p[0], s = p[1];
    const int off = (int)(x & 63);
    const uint32_t w = (off >> 4) == 0 ? s.x : (off >> 4) == 1 ? s.y : (off >> 4) == 2 ? s.z : s.w;
    const int c = (int)(w >> ((~off & 15) << 1) & 3);           // BWT symbol at rank k
    uint32_t c1 = 0, c2 = 0, c3 = 0;
    cnt_word(s.x, off + 1, c1, c2, c3);  cnt_word(s.y, off - 15, c1, c2, c3);
    cnt_word(s.z, off - 31, c1, c2, c3); cnt_word(s.w, off - 47, c1, c2, c3);
    uint64_t a0, a1, a2, a3;
    occ_unpack(cv, b, a0, a1, a2, a3);
    const uint64_t base = c == 0 ? a0 : c == 1 ? a1 : c == 2 ? a2 : a3;
    const uint32_t add = c == 0 ? (uint32_t)(off + 1 - (int)(c1 + c2 + c3)) : c == 1 ? c1 : c == 2 ? c2 : c3;
    return ix.L2[c] + base + add;
}

// text position of rank k: LF-walk to the next stored rank (row a7); n_lf counts the steps.  With the suffix array
// densified to every rank (the default when HBM allows, see upload_index) the walk is empty and this is one gather
DEV uint64_t sa_lookup(const DevIndex& ix, uint64_t k, uint32_t& n_lf)
{
    uint64_t sa = 0;
    const uint64_t mask = (uint64_t)ix.sa_intv - 1;
    while (k & mask) { ++sa; ++n_lf; k = lf_step(ix, k); }
    const uint64_t e = k >> ix.sa_shift;
    const uint64_t v = (uint64_t)ix.sa_lo[e] | (uint64_t)ix.sa_hi[e] << 32;
    return sa + (uint64_t)((int64_t)(v << 24) >> 24);           // 40-bit two's complement: entry 0 is -1
}

// ---------------------------------------------------------------- reference access (U8)
DEV int pac_base(const uint8_t* pac, int64_t l) { return *AS_GLOBAL(const uint8_t, pac + (l >> 2)) >> ((~l & 3) << 1) & 3; }   // (the packed reference is always in global memory)

// base at position p of the doubled (forward + reverse-complement) coordinate system
DEV int ref_base2(const DevIndex& ix, int64_t p)
{
    return p < ix.l_pac ? pac_base(ix.pac, p) : 3 - pac_base(ix.pac, (ix.l_pac << 1) - 1 - p);
}

// ref_base2 through a one-word cache: 16 reference bases per global load for loops that walk a stretch of the reference
struct PacCache { int64_t w; uint32_t v; };
DEV int ref_base2_cp(const uint8_t* pac, int64_t l_pac, PacCache& c, int64_t p)
{
    const bool rev = p >= l_pac;
    const int64_t l = rev ? (l_pac << 1) - 1 - p : p;
    const int64_t w = l >> 4;
    if (w != c.w) { c.w = w; c.v = *AS_GLOBAL(const uint32_t, (const uint32_t*)pac + w); }
    const int k = (int)(l & 15);
    const int b = (int)(c.v >> (((k >> 2) << 3) + 6 - ((k & 3) << 1)) & 3u);
    return rev ? 3 - b : b;
}
DEV int ref_base2_c(const DevIndex& ix, PacCache& c, int64_t p) { return ref_base2_cp(ix.pac, ix.l_pac, c, p); }

DEV int64_t bns_depos(const DevIndex& ix, int64_t pos, int& is_rev)
{
    return (is_rev = (pos >= ix.l_pac)) ? (ix.l_pac << 1) - 1 - pos : pos;
}

DEV int bns_pos2rid(const DevIndex& ix, int64_t pos_f)
{
    if (pos_f >= ix.l_pac) return -1;
    int left = 0, mid = 0, right = ix.n_seqs;
    while (left < right) {
        mid = (left + right) >> 1;
        if (pos_f >= ix.ann_offset[mid]) {
            if (mid == ix.n_seqs - 1) break;
            if (pos_f < ix.ann_offset[mid + 1]) break;
            left = mid + 1;
        } else right = mid;
    }
    return mid;
}

DEV int bns_intv2rid(const DevIndex& ix, int64_t rb, int64_t re)
{
    int is_rev;
    if (rb < ix.l_pac && re > ix.l_pac) return -2;
    int rid_b = bns_pos2rid(ix, bns_depos(ix, rb, is_rev));
    int rid_e = rb < re ? bns_pos2rid(ix, bns_depos(ix, re - 1, is_rev)) : rid_b;
    return rid_b == rid_e ? rid_b : -1;
}

// clamp [beg,end) to the contig (and strand) that holds mid; upstream bns_fetch_seq's window rule
DEV void bns_clamp(const DevIndex& ix, int64_t& beg, int64_t mid, int64_t& end, int& rid)
{
    int is_rev;
    if (end < beg) { int64_t t = beg; beg = end; end = t; }
    rid = bns_pos2rid(ix, bns_depos(ix, mid, is_rev));
    int64_t far_beg = ix.ann_offset[rid], far_end = far_beg + ix.ann_len[rid];
    if (is_rev) {
        int64_t tmp = far_beg;
        far_beg = (ix.l_pac << 1) - far_end;
        far_end = (ix.l_pac << 1) - tmp;
    }
    beg = beg > far_beg ? beg : far_beg;
    end = end < far_end ? end : far_end;
}

DEV uint64_t hash_64(uint64_t key)
{
    key += ~(key << 32);
    key ^= (key >> 22);
    key += ~(key << 13);
    key ^= (key >> 8);
    key += (key << 3);
    key ^= (key >> 15);
    key += ~(key << 27);
    key ^= (key >> 31);
    return key;
}

// Scoring-matrix access for the row loops of the DP kernels.  The matrix lives in the kernel argument block; indexing it
// with run-time values makes the compiler issue byte loads through the vector memory path in every DP row.  Instead the
// 25 entries are repacked once (wave-uniform, scalar registers) into one word per query base holding the scores against
// target bases 0..3, plus the score against an ambiguous target base; a lane then keeps the word of its own query base
// and a row costs one bit-field extract.
struct ScoreTab { uint32_t p[5]; int n[5]; };
DEV ScoreTab score_tab(const MemOpt& opt)
{
    ScoreTab T;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        T.p[q] = (uint32_t)(uint8_t)opt.mat[q] | (uint32_t)(uint8_t)opt.mat[5 + q] << 8 | (uint32_t)(uint8_t)opt.mat[10 + q] << 16 | (uint32_t)(uint8_t)opt.mat[15 + q] << 24;
        T.n[q] = opt.mat[20 + q];
    }
    return T;
}
DEV void score_lane(const ScoreTab& T, int q, uint32_t& p, int& n)
{
    p = q == 0 ? T.p[0] : q == 1 ? T.p[1] : q == 2 ? T.p[2] : q == 3 ? T.p[3] : T.p[4];
    n = q == 0 ? T.n[0] : q == 1 ? T.n[1] : q == 2 ? T.n[2] : q == 3 ? T.n[3] : T.n[4];
}
// score of the lane's query base against target base tb (0..4)
DEV int score_at(uint32_t p, int n, int tb) { return tb < 4 ? (int)(int8_t)(p >> (tb << 3)) : n; }
DEV int score_max(const MemOpt& opt) { int mx = 0; for (int k = 0; k < 25; ++k) mx = mx > opt.mat[k] ? mx : opt.mat[k]; return mx; }

// (int)((double)x / e + k) as upstream writes its gap bounds, in integers: x / e + k = (x + k e) / e exactly, a double
// quotient of two ints is nowhere near far enough from that to cross an integer, and both the cast and C division
// truncate toward zero.  (A double division costs ~30 vector instructions on this hardware; e is 1 by default.)
DEV int div_plus(int x, int e, int k)
{
    if (e == 1) return x + k;
    if (e <= 0) return (int)((double)x / e + (double)k);
    return (x + k * e) / e;
}

DEV int cal_max_gap(const MemOpt& opt, int qlen)
{
    int l_del = div_plus(qlen * opt.a - opt.o_del, opt.e_del, 1);
    int l_ins = div_plus(qlen * opt.a - opt.o_ins, opt.e_ins, 1);
    int l = l_del > l_ins ? l_del : l_ins;
    l = l > 1 ? l : 1;
    return l < opt.w << 1 ? l : opt.w << 1;
}

// ---------------------------------------------------------------- order-exact sort (U11)
// The unstable introsort whose tie permutation decides chain / region order:
// n==2 special case, median-of-three quicksort with explicit stack and depth limit
// 2*ceil(log2 n) falling back to combsort11, partitions <=16 finished by one insertion pass.
template <typename T, typename LT>
DEV void ks_insertsort(T* s, T* t, LT lt)
{
    for (T* i = s + 1; i < t; ++i)
        for (T* j = i; j > s && lt(*j, *(j - 1)); --j) { T tmp = *j; *j = *(j - 1); *(j - 1) = tmp; }
}

template <typename T, typename LT>
DEV void ks_combsort(size_t n, T* a, LT lt)
{
    const double shrink = 1.2473309501039786540366528676643;
    int do_swap;
    size_t gap = n;
    do {
        if (gap > 2) { gap = (size_t)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
        do_swap = 0;
        for (T* i = a; i < a + n - gap; ++i) {
            T* j = i + gap;
            if (lt(*j, *i)) { T tmp = *i; *i = *j; *j = tmp; do_swap = 1; }
        }
    } while (do_swap || gap > 2);
    if (gap != 1) ks_insertsort(a, a + n, lt);
}

// NB: the partition scans are written with explicit bounds and the pivot is read through a pointer.
// hipcc -O3 (ROCm 7.2, gfx950) miscompiles the upstream-style sentinel scans
// "do ++i; while (lt(*i, rp));" into a non-terminating loop (tests/gpu_units/sort_unit.hip);
// the decisions taken are identical because a[t] == pivot stops the scan at t at the latest.
template <typename T, typename LT>
DEV void ks_introsort(size_t n, T* a, LT lt)
{
    struct Frame { int left, right, depth; };        // offsets from a, not pointers: a pointer reloaded from private memory loses its address space (flat accesses)
    Frame stack[66];
    int sp = 0;
    if (n < 1) return;
    if (n == 2) {
        if (lt(a[1], a[0])) { T tmp = a[0]; a[0] = a[1]; a[1] = tmp; }
        return;
    }
    int d;
    for (d = 2; 1ul << d < n; ++d);
    T *s = a, *t = a + (n - 1);
    d <<= 1;
    bool done = false;
    while (!done) {
        if (s < t) {
            --d;
            if (d == 0) { ks_combsort((size_t)(t - s + 1), s, lt); t = s; }
            else {
                T *i = s, *j = t, *k = i + ((j - i) >> 1) + 1;
                if (lt(*k, *i)) { if (lt(*k, *j)) k = j; }
                else k = lt(*j, *i) ? i : j;
                if (k != t) { T tmp = *k; *k = *t; *t = tmp; }
                const T* rp = t;                      // the pivot now sits at t and is not moved by the scans
                bool more = true;
                while (more) {
                    ++i; while (i < t && lt(*i, *rp)) ++i;
                    --j; while (i <= j && lt(*rp, *j)) --j;
                    if (j <= i) more = false;
                    else { T tmp = *i; *i = *j; *j = tmp; }
                }
                { T tmp = *i; *i = *t; *t = tmp; }
                if (i - s > t - i) {
                    if (i - s > 16) { stack[sp].left = (int)(s - a); stack[sp].right = (int)(i - 1 - a); stack[sp].depth = d; ++sp; }
                    s = t - i > 16 ? i + 1 : t;
                } else {
                    if (t - i > 16) { stack[sp].left = (int)(i + 1 - a); stack[sp].right = (int)(t - a); stack[sp].depth = d; ++sp; }
                    t = i - s > 16 ? i - 1 : s;
                }
            }
        } else if (sp == 0) {
            ks_insertsort(a, a + n, lt);
            done = true;
        } else { --sp; s = a + stack[sp].left; t = a + stack[sp].right; d = stack[sp].depth; }
    }
}

// one atomic per wave for the instrumentation counters
DEV void count_add(unsigned long long* dst, unsigned long long v)
{
    if (v) atomicAdd(dst, v);
}
