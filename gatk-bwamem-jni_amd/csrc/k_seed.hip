// k_seed.hip -- seeding kernels: base encoding, three-pass SMEM collection over the
// HBM-resident occ table, seed-occurrence counting + scan, and sampled-SA lookup.
//
// Replaces, for the reference call at jnibwa.c:214, upstream bwamem.c mem_collect_intv and
// bwt.c bwt_smem1 / bwt_seed_strategy1 / bwt_extend / bwt_2occ4 / bwt_sa (SURVEY.md rows
// a2-a7).  One lane walks one read: every interval extension is two dependent random
// 64-byte gathers, so throughput comes from the number of independent reads in flight per
// CU, not from parallelism inside a read.
#include "dev_common.h"
#include "kernels.h"
#include "wave_ops.h"

// Per-read interval vectors live in global scratch laid out [entry][lane]: when the 64 reads of a wave push or
// read entry e together, the wave touches one contiguous 2 KB span instead of 64 scattered lines.
struct IntvVec {
    Intv* a; int n; int cap; int stride;
    __device__ Intv get(int i) const { return a[(size_t)i * stride]; }
    __device__ void set(int i, const Intv& v) { a[(size_t)i * stride] = v; }
    __device__ bool push(const Intv& v) { if (n >= cap) return false; a[(size_t)n * stride] = v; ++n; return true; }
};

DEV void vec_reverse(IntvVec& v, int from = 0)
{
    for (int i = from, j = v.n - 1; i < j; ++i, --j) { Intv t = v.get(i); v.set(i, v.get(j)); v.set(j, t); }
}

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

// mem_collect_intv (row a6): three seeding passes, written as a per-lane state machine.
//
// Upstream's control flow is four nested data-dependent loops per read.  Run naively one-read-per-lane, the
// lanes of a wave sit in different loops most of the time and the wave executes each lane's interval
// extensions almost serially (PMC: ~10 k extend steps per wave for ~860 per read).  Here every iteration of
// ONE wave-level loop performs exactly one interval extension per lane -- the expensive part: two dependent
// 64-byte occ gathers plus the popcounts -- and all bookkeeping between two extensions is cheap per-lane
// state transitions.  The sequence of extensions and pushes of each read is exactly upstream's
// (bwt_smem1 / bwt_seed_strategy1 / mem_collect_intv), so the resulting interval list is identical.
enum { ST_NEXT = 0, ST_FWD = 1, ST_BWD = 2, ST_P3 = 3, ST_DONE = 4 };

__global__ void k_seed(DevIndex ix, MemOpt opt, TileView tv)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = r < tv.n_reads;
    const int rr = in_range ? r : 0;
    uint32_t n_ext = 0;
    const uint8_t* q = tv.seq + tv.seq_off[rr];
    const int len = (int)(tv.seq_off[rr + 1] - tv.seq_off[rr] - 1);
    // scratch of wave-group g = r/64: [2 vectors][smem_cap entries][64 lanes]
    Intv* sc = tv.smem_scratch + ((size_t)(rr >> 6) * 2 * tv.smem_cap) * 64 + (rr & 63);
    // prev / curr candidate vectors: two lane-interleaved arrays whose roles swap; kept as plain registers
    // (an array indexed by the role bit would live in private scratch memory)
    Intv* const pa = sc; Intv* const pb = sc + (size_t)tv.smem_cap * 64;
    const int vcap = tv.smem_cap;
    int na = 0, nb = 0;
#define PREV_PTR (pv ? pb : pa)
#define CURR_PTR (pv ? pa : pb)
#define PREV_N (pv ? nb : na)
#define CURR_N (pv ? na : nb)
#define SET_CURR_N(v) do { if (pv) na = (v); else nb = (v); } while (0)
#define CURR_PUSH(val) do { int n_ = CURR_N; if (n_ >= vcap) ovf = true; else { CURR_PTR[(size_t)n_ * 64] = (val); SET_CURR_N(n_ + 1); } } while (0)
#define CURR_REVERSE() do { Intv* p_ = CURR_PTR; for (int a_ = 0, b_ = CURR_N - 1; a_ < b_; ++a_, --b_) { Intv t_ = p_[(size_t)a_ * 64]; p_[(size_t)a_ * 64] = p_[(size_t)b_ * 64]; p_[(size_t)b_ * 64] = t_; } } while (0)
    IntvVec mem = { tv.intv + (size_t)rr * tv.intv_cap, 0, tv.intv_cap, 1 };
    const int split_len = (int)(opt.min_seed_len * opt.split_factor + .499);
    const int min_seed_len = opt.min_seed_len;

    int st = (in_range && len >= min_seed_len) ? ST_NEXT : ST_DONE;
    int pass = 1, x = 0, i = 0, j = 0, c = 0, sx = 0, min_intv = 1, ret = 0, k2 = 0, old_n = 0, mem0 = 0;
    int pv = 0;                         // role bit: pv = 0 -> pa is prev, pb is curr
    bool any = false, ovf = false, row_start = false;
    uint64_t last_start = 0, last_curr_size = 0;
    Intv ik, req, ok;
    ik.x0 = ik.x1 = ik.size = ik.info = 0; req = ik; ok = ik;

    while (wave_any(st != ST_DONE)) {
        bool need = false;
        int is_back = 0;
        // ---- (a) per-lane transitions until this lane needs an interval extension
        while (st != ST_DONE && !need) {
            if (ovf) { st = ST_DONE; break; }
            if (st == ST_NEXT) {
                int nx = -1, nmin = 1;
                if (pass == 1) {
                    while (x < len && q[x] > 3) ++x;
                    if (x < len) { nx = x; nmin = 1; }
                    else { pass = 2; old_n = mem.n; k2 = 0; }
                } else if (pass == 2) {
                    while (k2 < old_n) {
                        Intv p = mem.get(k2);
                        int s0 = (int)(p.info >> 32), e0 = (int)(int32_t)p.info;
                        if (e0 - s0 < split_len || p.size > (uint64_t)(int64_t)opt.split_width) { ++k2; continue; }
                        nx = (s0 + e0) >> 1; nmin = (int)(p.size + 1);
                        break;
                    }
                    if (nx < 0) { pass = 3; x = 0; }
                    else if (q[nx] > 3) { ++k2; nx = -1; }          // bwt_smem1 returns at once on an ambiguous base
                } else {
                    if (!(opt.max_mem_intv > 0)) { st = ST_DONE; }
                    else {
                        while (x < len && q[x] > 3) ++x;
                        if (x >= len) st = ST_DONE;
                        else { set_intv(ix, q[x], ik); sx = x; i = x + 1; st = ST_P3; }
                    }
                }
                if (nx >= 0) {                                       // enter bwt_smem1(nx, nmin)
                    sx = nx; min_intv = nmin < 1 ? 1 : nmin;
                    set_intv(ix, q[sx], ik);
                    ik.info = (uint64_t)(sx + 1);
                    i = sx + 1; SET_CURR_N(0); mem0 = mem.n; any = false;
                    st = ST_FWD;
                }
            } else if (st == ST_FWD) {
                bool end_fwd = false;
                if (i < len) {
                    if (q[i] < 4) { req = ik; c = 3 - q[i]; is_back = 0; need = true; }
                    else { CURR_PUSH(ik); end_fwd = true; }
                } else { CURR_PUSH(ik); end_fwd = true; }
                if (end_fwd && !ovf) {                               // longest match first, then walk backwards
                    CURR_REVERSE();
                    ret = (int)CURR_PTR[0].info;
                    pv ^= 1;
                    i = sx - 1; j = 0; row_start = true;
                    st = ST_BWD;
                }
            } else if (st == ST_BWD) {
                if (row_start) { c = i < 0 ? -1 : (q[i] < 4 ? q[i] : -1); SET_CURR_N(0); row_start = false; }
                if (j < PREV_N) {
                    req = PREV_PTR[(size_t)j * 64];
                    if (c >= 0) { is_back = 1; need = true; }
                    else {                                           // start of the read or an ambiguous base: every candidate ends here
                        if (CURR_N == 0 && (!any || (uint64_t)(i + 1) < last_start)) {
                            any = true; last_start = (uint64_t)(i + 1);
                            Intv m = req; m.info |= (uint64_t)(i + 1) << 32;
                            if ((int)((uint32_t)m.info - (uint32_t)(m.info >> 32)) >= min_seed_len) { if (!mem.push(m)) ovf = true; }
                        }
                        ++j;
                    }
                } else if (CURR_N == 0) {                            // no candidate survived: this bwt_smem1 call is complete
                    vec_reverse(mem, mem0);
                    if (pass == 1) x = ret; else ++k2;
                    st = ST_NEXT;
                } else { pv ^= 1; --i; j = 0; row_start = true; }
            } else {                                                 // ST_P3: bwt_seed_strategy1
                if (i < len) {
                    if (q[i] < 4) { req = ik; c = 3 - q[i]; is_back = 0; need = true; }
                    else { x = i + 1; st = ST_NEXT; }
                } else { x = len; st = ST_NEXT; }
            }
        }
        // ---- (b) one interval extension per lane, executed by the whole wave together
        if (need) { ok = extend_one(ix, req, c, is_back); ++n_ext; }
        // ---- (c) consume the result
        if (need) {
            if (st == ST_FWD) {
                bool stop = false;
                if (ok.size != ik.size) {
                    CURR_PUSH(ik);
                    if (ok.size < (uint64_t)min_intv) stop = true;
                }
                if (stop) {                                          // upstream breaks with i < len: no final push
                    if (!ovf) {
                        CURR_REVERSE();
                        ret = (int)CURR_PTR[0].info;
                        pv ^= 1;
                        i = sx - 1; j = 0; row_start = true;
                        st = ST_BWD;
                    }
                } else { ik = ok; ik.info = (uint64_t)(i + 1); ++i; }
            } else if (st == ST_BWD) {
                if (ok.size < (uint64_t)min_intv) {
                    if (CURR_N == 0 && (!any || (uint64_t)(i + 1) < last_start)) {
                        any = true; last_start = (uint64_t)(i + 1);
                        Intv m = req; m.info |= (uint64_t)(i + 1) << 32;
                        if ((int)((uint32_t)m.info - (uint32_t)(m.info >> 32)) >= min_seed_len) { if (!mem.push(m)) ovf = true; }
                    }
                } else if (CURR_N == 0 || ok.size != last_curr_size) {
                    ok.info = req.info;
                    CURR_PUSH(ok);
                    last_curr_size = ok.size;
                }
                ++j;
            } else {                                                 // ST_P3
                if (ok.size < (uint64_t)(int64_t)(int)opt.max_mem_intv && i - sx >= min_seed_len) {
                    Intv m = ok;
                    m.info = (uint64_t)sx << 32 | (uint32_t)(i + 1);
                    if (m.size > 0) { if (!mem.push(m)) ovf = true; }
                    x = i + 1; st = ST_NEXT;
                } else { ik = ok; ++i; }
            }
        }
    }

    if (in_range) {
        int n_seeds = 0, l_rep = 0;
        if (!ovf) {
            // sort by info.  Intervals with equal info are the same substring, hence identical records, so the
            // order upstream's unstable sort leaves them in is unobservable: a plain insertion sort suffices.
            for (int a = 1; a < mem.n; ++a) {
                Intv t = mem.a[a];
                int b = a;
                while (b > 0 && mem.a[b - 1].info > t.info) { mem.a[b] = mem.a[b - 1]; --b; }
                mem.a[b] = t;
            }
            int b = 0, e = 0;
            int32_t* iso = tv.intv_seed_off + (size_t)r * tv.intv_cap;
            for (int a = 0; a < mem.n; ++a) {
                Intv p = mem.a[a];
                int sb = (int)(p.info >> 32), se = (int)(uint32_t)p.info;
                iso[a] = n_seeds;
                {
                    int64_t step = p.size > (uint64_t)(int64_t)opt.max_occ ? (int64_t)(p.size / (uint64_t)opt.max_occ) : 1;
                    int64_t cc = ((int64_t)p.size + step - 1) / step;
                    n_seeds += (int)(cc < opt.max_occ ? cc : opt.max_occ);
                }
                if (p.size <= (uint64_t)(int64_t)opt.max_occ) continue;
                if (sb > e) { l_rep += e - b; b = sb; e = se; }
                else e = e > se ? e : se;
            }
            l_rep += e - b;
        } else { atomicOr(tv.err, ERR_INTV_CAP); mem.n = 0; }
        tv.n_intv[r] = mem.n;
        tv.n_seeds[r] = n_seeds;
        tv.l_rep[r] = l_rep;
    }
    // one counter atomic per wave
    unsigned long long tot = n_ext;
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_down(tot, o);
    if ((threadIdx.x & 63) == 0) { count_add(&tv.cnt->n_ext, tot); }
}

// exclusive scan int32 -> int64 (single workgroup; n is a tile's read count, so this is tiny)
__global__ void k_scan(const int32_t* in, int64_t* out, int n)
{
    __shared__ int64_t part[1024];
    __shared__ int64_t carry;
    int t = threadIdx.x, nt = blockDim.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += nt) {
        int i = base + t;
        int64_t v = i < n ? in[i] : 0;
        part[t] = v;
        __syncthreads();
        for (int o = 1; o < nt; o <<= 1) {
            int64_t add = t >= o ? part[t - o] : 0;
            __syncthreads();
            part[t] += add;
            __syncthreads();
        }
        if (i < n) out[i] = carry + part[t] - v;
        __syncthreads();
        if (t == nt - 1) carry += part[t];
        __syncthreads();
    }
    if (t == 0) out[n] = carry;
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

void launch_encode(hipStream_t st, uint8_t* seq, int64_t n_bytes)
{
    if (n_bytes <= 0) return;
    int64_t nb = (n_bytes + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(k_encode, dim3((unsigned)nb), dim3(256), 0, st, seq, n_bytes);
}
void launch_seed(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv)
{
    if (tv.n_reads <= 0) return;
    hipLaunchKernelGGL(k_seed, dim3((tv.n_reads + 63) / 64), dim3(64), 0, st, ix, opt, tv);
}
void launch_scan(hipStream_t st, const int32_t* in, int64_t* out, int n)
{
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, in, out, n);
}
void launch_sa(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, int64_t n_occ)
{
    if (n_occ <= 0) return;
    hipLaunchKernelGGL(k_sa, dim3((unsigned)((n_occ + 255) / 256)), dim3(256), 0, st, ix, opt, tv, n_occ);
}
