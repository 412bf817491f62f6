// k_seed.hip -- seeding kernels: base encoding, three-pass SMEM collection over the
// HBM-resident occ table, seed-occurrence counting + scan, and sampled-SA lookup.
//
// Replaces, for the reference call at jnibwa.c:214, upstream bwamem.c mem_collect_intv and
// bwt.c bwt_smem1 / bwt_seed_strategy1 / bwt_extend / bwt_2occ4 / bwt_sa (SURVEY.md rows
// a2-a7).  One lane walks one read: every interval extension is a pair of dependent random
// 32-byte gathers, so throughput comes from the number of lanes that have a gather in flight,
// not from parallelism inside a read.
#include "dev_common.h"
#include "wave_ops.h"
#include "kernels.h"

// The symbol words and packed counts of the two blocks an extension needs are requested together (four 16-byte
// lane loads in flight), whether or not the two ranks share a block: one code path for the whole wave, and a repeated
// block is a cache hit.  Ranks are >= 0 here -- the seeding walk never asks for rank -1: every interval it extends
// starts at 1 or later (set by L2[c] + 1, kept by the extension formulas).
DEV uint32_t mask16(int n)                                  // bit-plane mask of the first n (clamped to 0..16) symbols of a word
{
    n = n < 0 ? 0 : n > 16 ? 16 : n;
    return (uint32_t)(0x5555555500000000ull >> (n << 1));
}
DEV void cnt_word_m(uint32_t x, uint32_t m, uint32_t& c1, uint32_t& c2, uint32_t& c3)
{
    const uint32_t lo = x & m, hi = (x >> 1) & m;
    c3 += __popc(hi & lo);
    c2 += __popc(hi & ~lo);
    c1 += __popc(~hi & lo);
}
// bwt_extend for one symbol, both directions through one code path (cf. extend_one / occ4 in dev_common.h; this form
// always issues the loads of both blocks at once and never forms the A counts from the block position: the counts of
// all four symbols up to a rank add up to the rank, so the A column follows from the other three).
// is_back = 1: the interval of bP for b = c; is_back = 0: upstream's ok[c] of bwt_extend(..., 0), i.e. the interval of
// P followed by base 3 - c.
DEV void extend_sm(const DevIndex& ix, uint64_t x0, uint64_t x1, uint64_t size, int c, bool is_back, uint64_t& o0, uint64_t& o1, uint64_t& osz)
{
    const uint64_t xa = is_back ? x0 : x1, xb = is_back ? x1 : x0;
    const uint64_t k = xa - 1, l = xa - 1 + size;
    const uint64_t kk = k - (k >= ix.primary), ll = l - (l >= ix.primary);
    const uint64_t bk = kk >> 6, bl = ll >> 6;
    const uint4* pk = ix.occ + 2 * bk;
    const uint4* pl = ix.occ + 2 * bl;
    const uint4 ck = pk[0], sk = pk[1], cl = pl[0], sl = pl[1];
    const int nk = (int)(kk & 63) + 1, nl = (int)(ll & 63) + 1;
    uint32_t a1 = 0, a2 = 0, a3 = 0, b1 = 0, b2 = 0, b3 = 0;
    cnt_word_m(sk.x, mask16(nk), a1, a2, a3);       cnt_word_m(sl.x, mask16(nl), b1, b2, b3);
    cnt_word_m(sk.y, mask16(nk - 16), a1, a2, a3);  cnt_word_m(sl.y, mask16(nl - 16), b1, b2, b3);
    cnt_word_m(sk.z, mask16(nk - 32), a1, a2, a3);  cnt_word_m(sl.z, mask16(nl - 32), b1, b2, b3);
    cnt_word_m(sk.w, mask16(nk - 48), a1, a2, a3);  cnt_word_m(sl.w, mask16(nl - 48), b1, b2, b3);
    // occ(k, b) and occ(l, b) for b = C, G, T
    const uint64_t tk1 = ((uint64_t)(ck.w & 0xffu) << 32 | ck.x) + a1, tk2 = ((uint64_t)(ck.w >> 8 & 0xffu) << 32 | ck.y) + a2, tk3 = ((uint64_t)(ck.w >> 16 & 0xffu) << 32 | ck.z) + a3;
    const uint64_t tl1 = ((uint64_t)(cl.w & 0xffu) << 32 | cl.x) + b1, tl2 = ((uint64_t)(cl.w >> 8 & 0xffu) << 32 | cl.y) + b2, tl3 = ((uint64_t)(cl.w >> 16 & 0xffu) << 32 | cl.z) + b3;
    const uint64_t s1 = tl1 - tk1, s2 = tl2 - tk2, s3 = tl3 - tk3;
    const uint64_t s0 = (ll - kk) - s1 - s2 - s3;          // ranks kk+1 and ll+1 are the totals over the four symbols
    const uint64_t tk0 = kk + 1 - tk1 - tk2 - tk3;
    const uint64_t tkc = c == 0 ? tk0 : c == 1 ? tk1 : c == 2 ? tk2 : tk3;
    const uint64_t sc = c == 0 ? s0 : c == 1 ? s1 : c == 2 ? s2 : s3;
    const uint64_t l2c = c == 0 ? ix.L2[0] : c == 1 ? ix.L2[1] : c == 2 ? ix.L2[2] : ix.L2[3];
    uint64_t other = xb + (xa <= ix.primary && xa + size - 1 >= ix.primary);
    other += (c < 3 ? s3 : 0) + (c < 2 ? s2 : 0) + (c < 1 ? s1 : 0);
    const uint64_t na = l2c + 1 + tkc;
    o0 = is_back ? na : other;
    o1 = is_back ? other : na;
    osz = sc;
}

// SMEM candidates of the search in progress, one private stack per lane: {x0:37, size:37, end:17} in 12 bytes.  The x1
// side of an interval is not kept: the backward phase of bwt_smem1 never reads it, and neither does anything after
// seeding (mem_collect_intv and mem_chain use x0, size and info only).  The top K entries of every lane's stack (K a
// power of two) sit in LDS as three [K][64] word planes addressed as a ring; entries pushed out of the ring by deeper
// stacks (many distinct interval sizes: repeats, long reads) live in a lane-interleaved global area.  The top is the
// hot part: the backward phase keeps its survivors packed against the top of the stack.  Limits checked on the host:
// text < 2^37 symbols, reads < 2^17 bases.
struct CandStack {
    uint32_t *v0, *v1, *v2;     // LDS planes, already offset by the lane
    uint16_t* v2n;              // the third plane as 16-bit entries when `narrow` (same base address as v2 would have)
    uint4* spill;               // [spill_cap][64], already offset by the lane
    int K, spill_cap;
    bool narrow;                // text < 2^35 symbols and reads < 1023 bases: {x0 hi:3, size hi:3, end:10} fit 16 bits -- 2 KB less LDS per wave, one more wave per SIMD pair
    static __device__ uint32_t pack2(uint64_t x0, uint64_t size, int end) { return (uint32_t)(x0 >> 32) | (uint32_t)(size >> 32) << 5 | (uint32_t)end << 10; }
    // third word of ring slot e, in pack2's layout
    __device__ uint32_t hi_get(int e) const {
        const int s = (e & (K - 1)) * 64;
        if (!narrow) return v2[s];
        const uint32_t h = v2n[s];
        return (h & 7u) | (h >> 3 & 7u) << 5 | (h >> 6) << 10;
    }
    __device__ void hi_put(int e, uint32_t w2) {
        const int s = (e & (K - 1)) * 64;
        if (!narrow) v2[s] = w2;
        else v2n[s] = (uint16_t)((w2 & 7u) | (w2 >> 5 & 7u) << 3 | (w2 >> 10) << 6);
    }
    // forward phase: entry e becomes the new top (height e + 1); the entry leaving the ring goes to the global area
    __device__ bool push(int e, uint64_t x0, uint64_t size, int end) {
        const int s = (e & (K - 1)) * 64;
        if (e >= K) {
            if (e - K >= spill_cap) return false;
            uint4 t; t.x = v0[s]; t.y = v1[s]; t.z = hi_get(e); t.w = 0;
            spill[(size_t)(e - K) * 64] = t;
        }
        v0[s] = (uint32_t)x0; v1[s] = (uint32_t)size; hi_put(e, pack2(x0, size, end));
        return true;
    }
    // backward phase, stack height n fixed: entries n-K .. n-1 are in the ring
    __device__ void put(int e, int n, uint64_t x0, uint64_t size, int end) {
        if (e >= n - K) { const int s = (e & (K - 1)) * 64; v0[s] = (uint32_t)x0; v1[s] = (uint32_t)size; hi_put(e, pack2(x0, size, end)); }
        else { uint4 t; t.x = (uint32_t)x0; t.y = (uint32_t)size; t.z = pack2(x0, size, end); t.w = 0; spill[(size_t)e * 64] = t; }
    }
    __device__ void get(int e, int n, uint64_t& x0, uint64_t& size, int& end) const {
        uint32_t a, b, w2;
        if (e >= n - K) { const int s = (e & (K - 1)) * 64; a = v0[s]; b = v1[s]; w2 = hi_get(e); }
        else { const uint4 t = spill[(size_t)e * 64]; a = t.x; b = t.y; w2 = t.z; }
        x0 = (uint64_t)(w2 & 31) << 32 | a; size = (uint64_t)(w2 >> 5 & 31) << 32 | b; end = (int)(w2 >> 10);
    }
};

// ASCII -> 0..4 in place (upstream nst_nt4_table; bytes < 4 are kept as they are)
__global__ void k_encode(uint8_t* seq, int64_t n_bytes)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n_bytes; i += stride) {
        uint8_t c = seq[i];
        if (c >= 4) {
            uint8_t u = c & 0xdf;   // fold case
            c = u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : u == 'T' ? 3 : 4;
            seq[i] = c;
        }
    }
}

// mem_collect_intv (row a6): the three seeding passes of every read, written as a per-lane state machine fed from a
// work queue.
//
// Upstream's control flow is four nested data-dependent loops per read.  Run one read per lane with that nesting, the
// lanes of a wave sit in different loops most of the time and the wave issues each lane's gathers almost serially
// (measured: ~10 k wave-level extension steps for ~860 extensions per read).  Here every iteration of ONE wave-level
// loop performs at most one interval extension per lane -- the expensive part: four 16-byte gathers from two random
// 32-byte occ blocks -- through a single call site shared by forward, backward and greedy extension, and all
// bookkeeping between two extensions is register/LDS-only state transitions (no loops, no global round trips: the
// candidate list is compacted in place instead of reversed and swapped).  A lane that finishes its read takes the next
// one from the tile's queue, so lanes stay busy until the tile runs dry and a few resident waves per CU keep the
// memory system saturated.  The sequence of extensions and candidate decisions of each read is exactly upstream's
// (bwt_smem1 / bwt_seed_strategy1 / mem_collect_intv); only the order of equal-priority list entries before the final
// sort differs, which the sort removes.
enum { S_IDLE = 0, S_NEXT = 1, S_FWD = 2, S_BWD = 3, S_P3 = 4 };
#define SEED_EL_CAP 4

template <bool LDSQ>
__global__ void __launch_bounds__(64) k_seed(DevIndex ix, MemOpt opt, TileView tv, int K, int refill_min, int narrow)
{
    HIP_DYNAMIC_SHARED(uint32_t, lds)
    const int lane = threadIdx.x;
    CandStack V;
    V.narrow = narrow != 0;
    V.v0 = lds + lane; V.v1 = V.v0 + K * 64; V.v2 = V.v1 + K * 64; V.K = K;
    V.v2n = (uint16_t*)(lds + 2 * K * 64) + lane;
    V.spill = (uint4*)tv.smem_scratch + (size_t)blockIdx.x * tv.smem_cap * 64 + lane; V.spill_cap = tv.smem_cap;
    uint32_t* el = lds + (narrow ? 5 * K * 32 : 3 * K * 64) + lane;   // [SEED_EL_CAP][64 lanes]: pass-1 matches that pass 2 re-seeds (mid:17 | size:15)
    uint32_t* sq = el + SEED_EL_CAP * 64;                // [word][64 lanes]: the lanes' current reads, 8 base codes per word
    unsigned int* work = (unsigned int*)(tv.err + 8);    // next unclaimed read of the tile
    const int min_seed_len = opt.min_seed_len;
    const int split_len = (int)(opt.min_seed_len * opt.split_factor + .499);
    const uint64_t max_intv3 = (uint64_t)(int64_t)(int)opt.max_mem_intv;

    int st = S_IDLE, r = 0, len = 0, pass = 1, x = 0, sx = 0, i = 0, k2 = 0, old_n = 0, mem_n = 0;
    int nf = 0, lo = 0, rd = 0, wr = 0, c = -1, ret = 0, end = 0, n_el = 0;
    uint64_t min_intv = 1, last_start = 0, last_sz = 0, ik0 = 0, ik1 = 0, iks = 0;
    bool any = false, ovf = false, exhausted = false, el_ok = true;
    const uint8_t* qg = tv.seq;
    Intv* mem = tv.intv;
    uint32_t n_ext = 0;

    // reads too long to stage in LDS (LDSQ = false) come from global memory: a lane walks its read one base at a time, forward
    // then backward, so an aligned 8-byte window in registers answers seven of eight look-ups without a load
    uint64_t qwin = 0;
    const uint8_t* qwin_at = nullptr;
    auto qat_g = [&](int p) -> int {
        const uint8_t* a = qg + p;
        const uint8_t* al = a - ((uintptr_t)a & 7);
        if (al != qwin_at) { qwin_at = al; qwin = *(const uint64_t*)al; }
        return (int)(qwin >> (((uintptr_t)a & 7) << 3) & 0xffu);
    };
#define QAT(p) (LDSQ ? (int)(sq[((p) >> 3) * 64] >> (((p) & 7) << 2) & 15u) : qat_g(p))
#define FINISH() do { fin = true; st = S_IDLE; } while (0)       // the read's result is stored once, at the end of the iteration
#define MEM_PUSH(X0, SZ, INFO) do { if (mem_n >= tv.intv_cap) ovf = true; else { Intv v_; v_.x0 = (X0); v_.x1 = 0; v_.size = (SZ); v_.info = (INFO); mem[mem_n++] = v_; } } while (0)
#define PUSH_IK() do { if (V.push(nf, ik0, iks, end)) ++nf; else ovf = true; } while (0)
#define BEGIN_BWD() do { ret = end; lo = 0; rd = wr = nf - 1; i = sx - 1; c = i >= 0 ? QAT(i) : 4; c = c < 4 ? c : -1; st = S_BWD; } while (0)

    for (;;) {
        bool fin = false;
        // ---- work queue: idle lanes claim the next reads of the tile, the wave stages them in LDS together
        const unsigned long long idle = __ballot(st == S_IDLE);
        if (idle != 0ull && !exhausted && (__popcll(idle) >= refill_min || idle == ~0ull)) {
            const int n_need = __popcll(idle);
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(work, (unsigned int)n_need);
            base = (unsigned int)__shfl((int)base, 0);
            if (base + (unsigned int)n_need >= (unsigned int)tv.n_reads) exhausted = true;
            bool fresh = false;
            if (st == S_IDLE) {
                const unsigned int cand = base + (unsigned int)__popcll(idle & ((1ull << lane) - 1ull));
                if (cand < (unsigned int)tv.n_reads) {
                    r = (int)cand; fresh = true;
                    const int64_t off = tv.seq_off[r];
                    len = (int)(tv.seq_off[r + 1] - off - 1);
                    qg = tv.seq + off;
                    mem = tv.intv + (size_t)r * tv.intv_cap;
                    mem_n = 0; ovf = false; pass = 1; x = 0; st = S_NEXT; n_el = 0; el_ok = true;
                }
            }
            if (LDSQ) {
                unsigned long long got = __ballot(fresh);
                while (got) {
                    const int l = __ffsll((long long)got) - 1;
                    got &= got - 1ull;
                    const int rr = __shfl(r, l);
                    const int64_t off = tv.seq_off[rr];
                    const int nb = (int)(tv.seq_off[rr + 1] - off) - 1;
                    for (int k = lane; k * 8 < nb; k += 64) {           // lane k packs bases 8k .. 8k+7 of the read claimed by lane l
                        uint32_t wd = 0;
                        uint32_t by[8];
#pragma unroll
                        for (int b = 0; b < 8; ++b) { const int pos = k * 8 + b; by[b] = tv.seq[off + (pos < nb ? pos : nb)]; }   // eight independent loads (nb: the terminator)
#pragma unroll
                        for (int b = 0; b < 8; ++b) wd |= (k * 8 + b < nb ? by[b] & 15u : 4u) << (b << 2);
                        sq[k * 64 + (l - lane)] = wd;
                    }
                }
                __syncthreads();
            }
            if (fresh && len < min_seed_len) { tv.n_intv[r] = 0; st = S_IDLE; }      // too short to seed
        }
        if (__ballot(st != S_IDLE) == 0ull) break;

        // ---- between two searches: next SMEM start (pass 1), next interval to re-seed (pass 2), next greedy seed (pass 3)
        if (st == S_NEXT) {
            int nx = -1;
            bool p3 = false;
            uint64_t nmin = 1;
            if (pass == 1) {
                if (x >= len) { pass = 2; old_n = mem_n; k2 = 0; }
                else if (QAT(x) > 3) ++x;
                else nx = x;
            } else if (pass == 2 && el_ok) {                        // the common case: candidates remembered in LDS
                if (k2 >= n_el) { pass = 3; x = 0; if (!(opt.max_mem_intv > 0)) FINISH(); }
                else {
                    const uint32_t wd = el[k2 * 64];
                    const int m = (int)(wd & 0x1ffffu);
                    if (QAT(m) > 3) ++k2;                           // bwt_smem1 returns at once on an ambiguous base
                    else { nx = m; nmin = (uint64_t)(wd >> 17) + 1; }
                }
            } else if (pass == 2) {                                 // more candidates than LDS slots: walk the list itself
                if (k2 >= old_n) { pass = 3; x = 0; if (!(opt.max_mem_intv > 0)) FINISH(); }
                else {
                    const uint64_t psz = mem[k2].size, pinfo = mem[k2].info;
                    const int s0 = (int)(pinfo >> 32), e0 = (int)(int32_t)pinfo;
                    if (e0 - s0 < split_len || psz > (uint64_t)(int64_t)opt.split_width) ++k2;
                    else {
                        const int m = (s0 + e0) >> 1;
                        if (QAT(m) > 3) ++k2;
                        else { nx = m; nmin = psz + 1; }
                    }
                }
            } else {
                if (x >= len) FINISH();
                else if (QAT(x) > 3) ++x;
                else { nx = x; p3 = true; }
            }
            if (nx >= 0) {                                          // enter bwt_smem1(nx, nmin) or bwt_seed_strategy1(nx): the interval of one base
                const int b = QAT(nx);
                const uint64_t lb = b == 0 ? ix.L2[0] : b == 1 ? ix.L2[1] : b == 2 ? ix.L2[2] : ix.L2[3];
                const uint64_t lb1 = b == 0 ? ix.L2[1] : b == 1 ? ix.L2[2] : b == 2 ? ix.L2[3] : ix.L2[4];
                const uint64_t lc = b == 0 ? ix.L2[3] : b == 1 ? ix.L2[2] : b == 2 ? ix.L2[1] : ix.L2[0];
                ik0 = lb + 1; iks = lb1 - lb; ik1 = lc + 1;
                sx = nx; i = sx + 1;
                if (p3) st = S_P3;
                else { min_intv = nmin; end = sx + 1; nf = 0; any = false; st = S_FWD; }
            }
        }

        // ---- what this lane extends in this iteration
        bool need = false, bw = false, back = false;
        int rc = 0, pend = 0;
        uint64_t r0 = 0, r1 = 0, rs = 0;
        bool fwd_end = false;                                       // forward phase stopped by the read's end or an ambiguous base
        if (st == S_FWD || st == S_P3) {
            const int cq = i < len ? QAT(i) : 4;
            if (cq < 4) { need = true; rc = 3 - cq; r0 = ik0; r1 = ik1; rs = iks; }
            else if (st == S_FWD) fwd_end = true;
            else { x = i < len ? i + 1 : len; st = S_NEXT; }        // bwt_seed_strategy1 found nothing from sx
        }
        if (st == S_BWD) {
            V.get(rd, nf, r0, rs, pend);
            bw = true;
            if (c >= 0) { need = true; back = true; rc = c; }
        }

        // ---- one interval extension per lane, issued by the whole wave together
        uint64_t o0 = 0, o1 = 0, os = 0, em0 = 0, ems = 0, emi = 0;
        bool emit = false;
        if (need) { extend_sm(ix, r0, r1, rs, rc, back, o0, o1, os); ++n_ext; }

        // ---- consume the result
        if (bw) {                                                   // backward phase: candidate rd of the row for position i
            if (!need || os < min_intv) {                           // cannot be extended to i: a match starts at i + 1
                if (wr == nf - 1 && (!any || (uint64_t)(i + 1) < last_start)) {
                    any = true; last_start = (uint64_t)(i + 1);
                    if (pend - (i + 1) >= min_seed_len) {
                        emit = true; em0 = r0; ems = rs; emi = (uint64_t)(i + 1) << 32 | (uint32_t)pend;
                        // pass 2 re-seeds long matches with few occurrences from their middle: remember those as they
                        // are found, so that pass 2 does not have to read the list back from global memory
                        if (pass == 1 && pend - (i + 1) >= split_len && rs <= (uint64_t)(int64_t)opt.split_width) {
                            if (n_el < SEED_EL_CAP && rs < 32768u) el[n_el * 64] = (uint32_t)((i + 1 + pend) >> 1) | (uint32_t)rs << 17;
                            else el_ok = false;
                            ++n_el;
                        }
                    }
                }
            } else if (wr == nf - 1 || os != last_sz) {             // survivor; rows are compacted in place, longest on top
                V.put(wr, nf, o0, os, pend); --wr; last_sz = os;
            }
            --rd;
            if (rd < lo) {                                          // row complete
                if (wr == nf - 1) { if (pass == 1) x = ret; else ++k2; st = S_NEXT; }
                else { lo = wr + 1; rd = wr = nf - 1; --i; c = i >= 0 ? QAT(i) : 4; c = c < 4 ? c : -1; }
            }
        } else if (st == S_FWD && (need || fwd_end)) {              // upstream pushes the interval whenever its size is about to change
            const bool push = fwd_end || os != iks;
            const bool stop = fwd_end || (push && os < min_intv);
            if (push) PUSH_IK();
            if (stop) BEGIN_BWD();                                  // (moot if the push overflowed: the read is abandoned below)
            else { ik0 = o0; ik1 = o1; iks = os; end = i + 1; ++i; }
        } else if (need) {                                          // S_P3
            if (os < max_intv3 && i - sx >= min_seed_len) {
                if (os > 0) { emit = true; em0 = o0; ems = os; emi = (uint64_t)sx << 32 | (uint32_t)(i + 1); }
                x = i + 1; st = S_NEXT;
            } else { ik0 = o0; ik1 = o1; iks = os; ++i; }
        }
        if (emit) MEM_PUSH(em0, ems, emi);                          // one store site for both kinds of match
        if (ovf) FINISH();
        if (fin) {                                                  // read r is complete (or ran out of room)
            tv.n_intv[r] = ovf ? 0 : mem_n;
            if (ovf) atomicOr(tv.err, ERR_INTV_CAP);
        }
    }
#undef QAT
#undef FINISH
#undef MEM_PUSH
#undef PUSH_IK
#undef BEGIN_BWD
    // one counter atomic per wave
    unsigned long long tot = n_ext;
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_down(tot, o);
    if (lane == 0) { count_add(&tv.cnt->n_ext, tot); }
}

// the per-read bookkeeping between mem_collect_intv and the SA look-ups of mem_chain: intervals in upstream's order
// (sorted by info), the number of occurrences each contributes, and l_rep (repetitive fraction numerator)
__global__ void k_seed_fin(MemOpt opt, TileView tv)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= tv.n_reads) return;
    Intv* a = tv.intv + (size_t)r * tv.intv_cap;
    const int n = tv.n_intv[r];
    // Intervals with equal info are the same substring, hence identical records, so the order upstream's unstable
    // sort leaves them in is unobservable: a plain insertion sort suffices.
    for (int i = 1; i < n; ++i) {
        Intv t = a[i];
        int j = i;
        while (j > 0 && a[j - 1].info > t.info) { a[j] = a[j - 1]; --j; }
        a[j] = t;
    }
    int n_seeds = 0, l_rep = 0, b = 0, e = 0;
    int32_t* iso = tv.intv_seed_off + (size_t)r * tv.intv_cap;
    for (int i = 0; i < n; ++i) {
        const Intv p = a[i];
        const int sb = (int)(p.info >> 32), se = (int)(uint32_t)p.info;
        iso[i] = n_seeds;
        {
            int64_t step = p.size > (uint64_t)(int64_t)opt.max_occ ? (int64_t)(p.size / (uint64_t)opt.max_occ) : 1;
            int64_t c = ((int64_t)p.size + step - 1) / step;
            n_seeds += (int)(c < opt.max_occ ? c : opt.max_occ);
        }
        if (p.size <= (uint64_t)(int64_t)opt.max_occ) continue;
        if (sb > e) { l_rep += e - b; b = sb; e = se; }
        else e = e > se ? e : se;
    }
    l_rep += e - b;
    tv.n_seeds[r] = n_seeds;
    tv.l_rep[r] = l_rep;
}

// exclusive scan int32 -> int64 of a tile's per-read counts.  One workgroup walks the array in tiles of 4096: coalesced
// 16-byte loads, a 4-element serial prefix per thread, a shuffle scan per wave, the 16 wave totals through LDS, and a
// running carry between tiles.
__global__ void __launch_bounds__(1024) k_scan(const int32_t* in, int64_t* out, int n)
{
    __shared__ int64_t wtot[2][16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int64_t carry = 0;
    int ph = 0;
    for (int base = 0; base < n; base += 4096, ph ^= 1) {
        const int i0 = base + t * 4;
        int v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        if (i0 + 3 < n) { const int4 q = *(const int4*)(in + i0); v0 = q.x; v1 = q.y; v2 = q.z; v3 = q.w; }
        else { if (i0 < n) v0 = in[i0]; if (i0 + 1 < n) v1 = in[i0 + 1]; if (i0 + 2 < n) v2 = in[i0 + 2]; }
        const int64_t sum = (int64_t)v0 + v1 + v2 + v3;
        int64_t incl = sum;
        for (int o = 1; o < 64; o <<= 1) { int64_t u = __shfl_up(incl, o); if (lane >= o) incl += u; }
        if (lane == 63) wtot[ph][wv] = incl;
        __syncthreads();
        int64_t pre = carry, tot = 0;
        for (int w = 0; w < 16; ++w) { const int64_t x = wtot[ph][w]; if (w < wv) pre += x; tot += x; }
        int64_t run = pre + incl - sum;
        if (i0 < n) out[i0] = run;
        run += v0; if (i0 + 1 < n) out[i0 + 1] = run;
        run += v1; if (i0 + 2 < n) out[i0 + 2] = run;
        run += v2; if (i0 + 3 < n) out[i0 + 3] = run;
        carry += tot;
    }
    if (t == 0) out[n] = carry;
}

// ---- multi-block form of the scan above (launch_scan)
DEV int64_t block_excl_scan_256(int64_t v, int64_t* lds4, int64_t& total)
{   // exclusive prefix of v over the 256 threads of a workgroup (4 waves); total = sum over the workgroup
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int64_t incl = v;
    for (int o = 1; o < 64; o <<= 1) { int64_t u = __shfl_up(incl, o); if (lane >= o) incl += u; }
    if (lane == 63) lds4[wv] = incl;
    __syncthreads();
    int64_t pre = 0; total = 0;
    for (int w = 0; w < 4; ++w) { const int64_t x = lds4[w]; if (w < wv) pre += x; total += x; }
    __syncthreads();
    return pre + incl - v;
}
__global__ void __launch_bounds__(256) k_scan_part(const int32_t* in, int n, int64_t* part)
{
    __shared__ int64_t w4[4];
    const int i0 = blockIdx.x * 4096 + threadIdx.x * 16;
    int64_t sum = 0;
    for (int k = 0; k < 16; k += 4) {
        const int i = i0 + k;
        if (i + 3 < n) { const int4 q = *(const int4*)(in + i); sum += (int64_t)q.x + q.y + q.z + q.w; }
        else for (int j = i; j < n && j < i + 4; ++j) sum += in[j];
    }
    int64_t tot;
    (void)block_excl_scan_256(sum, w4, tot);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(64) k_scan_top(int64_t* part, int nb, int64_t* total_out)
{   // in-place exclusive scan of many block sums (arrays beyond 2048 blocks): one wavefront, which finds a free
    // slot next to other tiles' kernels where a 1024-thread workgroup waits for sixteen on one CU (measured: 4.5 ms per launch)
    const int lane = threadIdx.x;
    int64_t carry = 0;
    for (int base = 0; base < nb; base += 64) {
        const int i = base + lane;
        const int64_t v = i < nb ? part[i] : 0;
        int64_t incl = v;
        for (int o = 1; o < 64; o <<= 1) { int64_t u = __shfl_up(incl, o); if (lane >= o) incl += u; }
        if (i < nb) part[i] = carry + incl - v;
        carry += __shfl(incl, 63);
    }
    if (lane == 0) *total_out = carry;
}
template <bool FUSED>
__global__ void __launch_bounds__(256) k_scan_apply(const int32_t* in, int n, const int64_t* part, int64_t* out)
{   // FUSED: part[] holds the block sums themselves and every block adds up the ones before it (no launch in between)
    __shared__ int64_t w4[4];
    const int i0 = blockIdx.x * 4096 + threadIdx.x * 16;
    int v[16];
    int64_t sum = 0;
    for (int k = 0; k < 16; ++k) { v[k] = i0 + k < n ? in[i0 + k] : 0; sum += v[k]; }
    int64_t tot, base;
    if (FUSED) {
        int64_t mine = 0, before;
        for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) mine += part[b];
        (void)block_excl_scan_256(mine, w4, before);
        base = before;
    } else base = part[blockIdx.x];
    int64_t run = base + block_excl_scan_256(sum, w4, tot);
    for (int k = 0; k < 16; ++k) { if (i0 + k < n) out[i0 + k] = run; run += v[k]; }
    if (FUSED && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = base + tot;
}

// ---- reads by falling seed count (TileView::order): a counting sort over 32 logarithmic bins.  The order inside a bin is
// whatever the atomics give: it decides which lane works on which read, never a result.
DEV int order_bin(int n_seeds) { return n_seeds <= 0 ? 31 : __clz(n_seeds); }          // many seeds -> small bin -> early in the order
__global__ void __launch_bounds__(256) k_order_count(const int32_t* n_seeds, int n, int32_t* bins)
{
    __shared__ int32_t h[32];
    if (threadIdx.x < 32) h[threadIdx.x] = 0;
    __syncthreads();
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) atomicAdd(&h[order_bin(n_seeds[r])], 1);
    __syncthreads();
    if (threadIdx.x < 32 && h[threadIdx.x]) atomicAdd(&bins[threadIdx.x], h[threadIdx.x]);
}
__global__ void k_order_offsets(int32_t* bins)                  // bins[0..32): counts -> bins[32..64): running cursors
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { int acc = 0; for (int b = 0; b < 32; ++b) { bins[32 + b] = acc; acc += bins[b]; } }
}
__global__ void __launch_bounds__(256) k_order_scatter(const int32_t* n_seeds, int n, int32_t* bins, int32_t* order)
{   // a block ranks its reads within each bin in LDS and reserves the bin's range with one global atomic: ten million reads of
    // like weight would otherwise queue on a single cursor
    __shared__ int32_t h[32], base[32];
    if (threadIdx.x < 32) h[threadIdx.x] = 0;
    __syncthreads();
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    int b = 0, rank = 0;
    if (r < n) { b = order_bin(n_seeds[r]); rank = atomicAdd(&h[b], 1); }
    __syncthreads();
    if (threadIdx.x < 32) base[threadIdx.x] = h[threadIdx.x] ? atomicAdd(&bins[32 + threadIdx.x], h[threadIdx.x]) : 0;
    __syncthreads();
    if (r < n) order[base[b] + rank] = r;
}
void launch_order(hipStream_t st, const int32_t* n_seeds, int n, int32_t* bins64, int32_t* order)
{
    if (n <= 0) return;
    (void)hipMemsetAsync(bins64, 0, 64 * sizeof(int32_t), st);
    hipLaunchKernelGGL(k_order_count, dim3((n + 255) / 256), dim3(256), 0, st, n_seeds, n, bins64);
    hipLaunchKernelGGL(k_order_offsets, dim3(1), dim3(64), 0, st, bins64);
    hipLaunchKernelGGL(k_order_scatter, dim3((n + 255) / 256), dim3(256), 0, st, n_seeds, n, bins64, order);
}

// ---- read offsets of a stretch of the request, found on the device (launch_nul_offsets)
DEV uint32_t zero_byte_mask(uint32_t w) { return ~(((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w | 0x7f7f7f7fu); }   // 0x80 in every byte of w that is 0
DEV void nul_words(const uint8_t* seq, int64_t n_bytes, int64_t b0, uint32_t m[4])
{   // zero-byte masks of the 16 bytes at b0 (16-byte aligned; the buffer has >= 64 bytes of slack), bytes >= n_bytes ignored
    const uint4 q = *(const uint4*)(seq + b0);
    const uint32_t w[4] = { q.x, q.y, q.z, q.w };
    for (int k = 0; k < 4; ++k) {
        const int64_t left = n_bytes - (b0 + 4 * k);
        uint32_t z = zero_byte_mask(w[k]);
        if (left < 4) z = left <= 0 ? 0u : z & (0xffffffffu >> (8 * (4 - (int)left)));
        m[k] = z;
    }
}
__global__ void __launch_bounds__(256) k_nul_count(const uint8_t* seq, int64_t n_bytes, int32_t* cnt)
{
    __shared__ int64_t w4[4];
    const int64_t b0 = (int64_t)blockIdx.x * 4096 + threadIdx.x * 16;
    uint32_t m[4] = { 0, 0, 0, 0 };
    if (b0 < n_bytes) nul_words(seq, n_bytes, b0, m);
    int64_t tot;
    (void)block_excl_scan_256((int64_t)(__popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3])), w4, tot);
    if (threadIdx.x == 0) cnt[blockIdx.x] = (int32_t)tot;
}
__global__ void __launch_bounds__(256) k_nul_write(const uint8_t* seq, int64_t n_bytes, const int64_t* blk_off, int nb, int64_t* off, int64_t n_reads_max, int64_t* n_found)
{
    __shared__ int64_t w4[4];
    const int64_t b0 = (int64_t)blockIdx.x * 4096 + threadIdx.x * 16;
    uint32_t m[4] = { 0, 0, 0, 0 };
    if (b0 < n_bytes) nul_words(seq, n_bytes, b0, m);
    int64_t tot;
    int64_t k = blk_off[blockIdx.x] + block_excl_scan_256((int64_t)(__popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3])), w4, tot);
    for (int w = 0; w < 4; ++w)
        for (int j = 0; j < 4; ++j)
            if (m[w] >> (8 * j + 7) & 1) { if (k < n_reads_max) off[k + 1] = b0 + 4 * w + j + 1; ++k; }
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_found = blk_off[nb];
}

// one lane per seed occurrence: rank -> text position by LF-walk + sample (row a7), then the
// contig test of mem_chain (bns_intv2rid; occurrences bridging contigs or strands are dropped)
__global__ void k_sa(DevIndex ix, MemOpt opt, TileView tv, int64_t n_occ)
{
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n_lf = 0, n_sa = 0;
    if (g < n_occ) {
        int lo = 0, hi = tv.n_reads;                    // read r with seed_off[r] <= g < seed_off[r+1]
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (tv.seed_off[mid] <= g) lo = mid; else hi = mid; }
        int r = lo;
        int local = (int)(g - tv.seed_off[r]);
        const int32_t* iso = tv.intv_seed_off + (size_t)r * tv.intv_cap;
        int n = tv.n_intv[r];
        lo = 0; hi = n;
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (iso[mid] <= local) lo = mid; else hi = mid; }
        Intv p = tv.intv[(size_t)r * tv.intv_cap + lo];
        int64_t step = p.size > (uint64_t)(int64_t)opt.max_occ ? (int64_t)(p.size / (uint64_t)opt.max_occ) : 1;
        int64_t k = (int64_t)(local - iso[lo]) * step;
        Seed s;
        s.rbeg = (int64_t)sa_lookup(ix, p.x0 + (uint64_t)k, n_lf); ++n_sa;
        s.qbeg = (int32_t)(p.info >> 32);
        s.len = s.score = (int32_t)((uint32_t)p.info - (uint32_t)(p.info >> 32));
        s.next = -1;
        tv.seeds[g] = s;
        tv.seed_rid[g] = bns_intv2rid(ix, s.rbeg, s.rbeg + s.len);
    }
    unsigned long long a = n_lf, b = n_sa;
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); }
    if ((threadIdx.x & 63) == 0) { count_add(&tv.cnt->n_lf, a); count_add(&tv.cnt->n_sa, b); }
}

// Suffix-array densification at index load.  The image samples SA at every src_intv-th rank, which makes a lookup an
// LF-walk of geometrically distributed length (mean src_intv - 1 dependent occurrence-table gathers, the slowest lane of
// a wave several times that).  HBM has room for much more: walking LF from a sampled rank visits ranks whose text
// position decreases by one per step until the next sampled rank, and those walks partition the unsampled ranks (LF
// is a permutation), so one lane per sampled rank fills the whole array in seq_len steps in total.  Entries are kept
// for every ix.sa_intv-th rank (ix.sa_intv divides src_intv), 40 bits each.
__global__ void k_sa_densify(DevIndex ix, const uint64_t* sa_src, uint64_t n_src, int src_intv, uint32_t* lo, uint8_t* hi, int32_t* err)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_src) return;
    uint64_t k = t * (uint64_t)src_intv, v = sa_src[t];
    lo[k >> ix.sa_shift] = (uint32_t)v; hi[k >> ix.sa_shift] = (uint8_t)(v >> 32);
    if (ix.sa_intv == src_intv) return;
    if (t == 0) v = ix.seq_len;                                  // rank 0 is the empty suffix (stored as -1 upstream)
    const uint64_t smask = (uint64_t)src_intv - 1, dmask = (uint64_t)ix.sa_intv - 1;
    for (uint64_t step = 0; ; ++step) {
        if (step > ix.seq_len) { atomicOr(err, ERR_SCRATCH); return; }   // not a permutation: corrupt index
        k = lf_step(ix, k);
        if (!(k & smask)) return;
        --v;
        if (!(k & dmask)) { lo[k >> ix.sa_shift] = (uint32_t)v; hi[k >> ix.sa_shift] = (uint8_t)(v >> 32); }
    }
}

void launch_sa_densify(hipStream_t st, const DevIndex& ix, const uint64_t* sa_src, uint64_t n_src, int src_intv, uint32_t* lo, uint8_t* hi, int32_t* err)
{
    hipLaunchKernelGGL(k_sa_densify, dim3((unsigned)((n_src + 255) / 256)), dim3(256), 0, st, ix, sa_src, n_src, src_intv, lo, hi, err);
}

// image occ/bwt layout (128-symbol blocks: 4 x u64 counts + 8 x u32 symbols) -> device layout (see bwamem_types.h)
__global__ void k_build_occ64(const uint32_t* bwt, uint64_t n_blocks, uint4* occ)
{
    uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const uint32_t* blk = bwt + (b >> 1 << 4);                  // the 128-symbol source block
    const uint64_t* cnt = (const uint64_t*)blk;
    const uint32_t* sym = blk + 8 + ((b & 1) << 2);
    uint32_t c1 = 0, c2 = 0, c3 = 0;
    if (b & 1) for (int i = 0; i < 4; ++i) cnt_word(blk[8 + i], 16, c1, c2, c3);   // first half of the source block
    const uint64_t C = cnt[1] + c1, G = cnt[2] + c2, T = cnt[3] + c3;
    uint4 c, s;
    c.x = (uint32_t)C; c.y = (uint32_t)G; c.z = (uint32_t)T;
    c.w = (uint32_t)(C >> 32 & 0xff) | (uint32_t)(G >> 32 & 0xff) << 8 | (uint32_t)(T >> 32 & 0xff) << 16;
    s.x = sym[0]; s.y = sym[1]; s.z = sym[2]; s.w = sym[3];
    occ[2 * b] = c; occ[2 * b + 1] = s;
}

void launch_build_occ64(hipStream_t st, const uint32_t* bwt, uint64_t n_blocks, uint4* occ)
{
    hipLaunchKernelGGL(k_build_occ64, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, st, bwt, n_blocks, occ);
}

// packed reference -> one base code (0..3) per byte (bench / tooling: reads sampled from an existing image)
__global__ void k_unpack_pac(const uint8_t* pac, int64_t start, int64_t n, uint8_t* dst)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (uint8_t)pac_base(pac, start + i);
}
void launch_unpack_pac(hipStream_t st, const DevIndex& ix, int64_t start, int64_t n, uint8_t* dst)
{
    if (n <= 0) return;
    int64_t nb = (n + 255) / 256;
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(k_unpack_pac, dim3((unsigned)nb), dim3(256), 0, st, ix.pac, start, n, dst);
}

void launch_encode(hipStream_t st, uint8_t* seq, int64_t n_bytes)
{
    if (n_bytes <= 0) return;
    int64_t nb = (n_bytes + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(k_encode, dim3((unsigned)nb), dim3(256), 0, st, seq, n_bytes);
}
// k_seed is a persistent grid: as many one-wave workgroups as fit the CUs' LDS (the candidate stacks dominate), each
// pulling reads from the tile's queue (tv.err[8], zeroed by the caller with the other flags).
void launch_seed(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv)
{
    if (tv.n_reads <= 0) return;
    const int n_cu = ix.n_cu > 0 ? ix.n_cu : 256;
    int K = 16, wpc = 0, refill_min = 4;
    { const char* e = getenv("BWAMEM_HIP_SEED_REFILL"); if (e && atoi(e) > 0) refill_min = atoi(e); }
    { const char* e = getenv("BWAMEM_HIP_SEED_K"); if (e && atoi(e) > 0) K = atoi(e); }
    { const char* e = getenv("BWAMEM_HIP_SEED_WPC"); if (e && atoi(e) > 0) wpc = atoi(e); }
    while (K & (K - 1)) K &= K - 1;                                  // the candidate ring needs a power of two
    const size_t qbytes = (size_t)64 * 4 * (((size_t)tv.max_len + 7) / 8);
    const bool ldsq = qbytes <= 24576;
    int narrow = ix.seq_len < (1ull << 35) && tv.max_len < 1023;
    { const char* e = getenv("BWAMEM_HIP_SEED_NARROW"); if (e) narrow = narrow && atoi(e) != 0; }
    const size_t lds = (size_t)(narrow ? 5 * K * 32 : 3 * K * 64) * 4 + (size_t)SEED_EL_CAP * 64 * 4 + (ldsq ? qbytes : 0) + 16;
    if (!wpc) { wpc = (int)((size_t)(160 * 1024) / ((lds + 1023) & ~(size_t)1023)); wpc = wpc < 1 ? 1 : wpc > 16 ? 16 : wpc; }
    const int groups = (tv.n_reads + 63) / 64;
    int grid = groups < n_cu * wpc ? groups : n_cu * wpc;
    if (tv.smem_groups > 0 && grid > tv.smem_groups) grid = tv.smem_groups;
    if (ldsq) hipLaunchKernelGGL(k_seed<true>, dim3(grid), dim3(64), lds, st, ix, opt, tv, K, refill_min, narrow);
    else hipLaunchKernelGGL(k_seed<false>, dim3(grid), dim3(64), lds, st, ix, opt, tv, K, refill_min, narrow);
    hipLaunchKernelGGL(k_seed_fin, dim3((tv.n_reads + 255) / 256), dim3(256), 0, st, opt, tv);
}
size_t scan_tmp_bytes(int64_t n) { return (size_t)((n + 4095) / 4096 + 2) * 8; }
// exclusive scan int32 -> int64, out[n] = total.  Short arrays: the one-workgroup kernel.  Longer ones in three small
// launches over 4096-element blocks (block sums; beyond 2048 blocks a one-wave scan of the sums; block-local scan +
// offset), so that no stage of a tile waits on one large workgroup finding room next to other tiles' kernels.
// tmp: scan_tmp_bytes(n) bytes.
static int scan_knob(const char* name, int dflt) { const char* e = getenv(name); return e && atoi(e) > 0 ? atoi(e) : dflt; }
void launch_scan(hipStream_t st, const int32_t* in, int64_t* out, int n, int64_t* tmp)
{
    const int single_max = scan_knob("BWAMEM_HIP_SCAN_SINGLE_MAX", 8192), fused_max = scan_knob("BWAMEM_HIP_SCAN_FUSED_MAX", 2048);   // test knobs
    if (n <= single_max || !tmp) { hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, in, out, n); return; }
    const int nb = (n + 4095) / 4096;
    hipLaunchKernelGGL(k_scan_part, dim3(nb), dim3(256), 0, st, in, n, tmp);
    if (nb <= fused_max) { hipLaunchKernelGGL(k_scan_apply<true>, dim3(nb), dim3(256), 0, st, in, n, (const int64_t*)tmp, out); return; }
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(64), 0, st, tmp, nb, out + n);
    hipLaunchKernelGGL(k_scan_apply<false>, dim3(nb), dim3(256), 0, st, in, n, (const int64_t*)tmp, out);
}
size_t nul_tmp_bytes(int64_t n_bytes) { const size_t nb = (size_t)((n_bytes + 15) / 4096 + 2); return nb * 4 + 128 + (nb + 1) * 8 + scan_tmp_bytes((int64_t)nb); }
// Offsets of the NUL-terminated reads of a stretch of the request (the job of the strlen walk at jnibwa.c:204-212, done
// where the bytes already are): off[0] = 0, off[k + 1] = position after the k-th NUL.  n_reads_max bounds the writes;
// *n_found (device) receives the number of NULs seen.  tmp: nul_tmp_bytes(n_bytes) bytes.
void launch_nul_offsets(hipStream_t st, const uint8_t* seq, int64_t n_bytes, int64_t* off, int64_t n_reads_max, int64_t* n_found, void* tmp)
{
    const int nb = (int)((n_bytes + 4095) / 4096);
    int32_t* cnt = (int32_t*)tmp;
    int64_t* blk_off = (int64_t*)((char*)tmp + (((size_t)(nb + 2) * 4 + 63) & ~(size_t)63));
    int64_t* scan_tmp = blk_off + nb + 1;
    (void)hipMemsetAsync(off, 0, 8, st);
    if (nb <= 0) { (void)hipMemsetAsync(n_found, 0, 8, st); return; }
    hipLaunchKernelGGL(k_nul_count, dim3(nb), dim3(256), 0, st, seq, n_bytes, cnt);
    launch_scan(st, cnt, blk_off, nb, scan_tmp);
    hipLaunchKernelGGL(k_nul_write, dim3(nb), dim3(256), 0, st, seq, n_bytes, (const int64_t*)blk_off, nb, off, n_reads_max, n_found);
}
void launch_sa(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, int64_t n_occ)
{
    if (n_occ <= 0) return;
    hipLaunchKernelGGL(k_sa, dim3((unsigned)((n_occ + 255) / 256)), dim3(256), 0, st, ix, opt, tv, n_occ);
}
