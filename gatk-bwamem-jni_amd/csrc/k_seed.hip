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

// Per-read interval vectors live in global scratch laid out [entry][lane]: when the 64 reads of a wave push or
// read entry e together, the wave touches one contiguous 2 KB span instead of 64 scattered lines.
struct IntvVec {
    Intv* a; int n; int cap; int stride;
    __device__ Intv get(int i) const { return a[(size_t)i * stride]; }
    __device__ void set(int i, const Intv& v) { a[(size_t)i * stride] = v; }
    __device__ bool push(const Intv& v) { if (n >= cap) return false; a[(size_t)n * stride] = v; ++n; return true; }
};

// prev / curr candidates of one SMEM search: 16-byte packed entries (x0, x1, size: 37 bits each; end: 17 bits), so a
// push or a read is one 16-byte lane request.  Limits (checked on the host): text < 2^37 symbols, reads < 2^17 bases.
struct PackedVec {
    uint4* a; int n; int cap; int stride;
    static __device__ uint4 pack(const Intv& v) {
        uint64_t lo = v.x0 | (v.x1 << 37), hi = (v.x1 >> 27) | (v.size << 10) | ((v.info & 0x1ffff) << 47);
        uint4 r; r.x = (uint32_t)lo; r.y = (uint32_t)(lo >> 32); r.z = (uint32_t)hi; r.w = (uint32_t)(hi >> 32);
        return r;
    }
    static __device__ Intv unpack(const uint4& r) {
        uint64_t lo = (uint64_t)r.y << 32 | r.x, hi = (uint64_t)r.w << 32 | r.z;
        Intv v; v.x0 = lo & 0x1fffffffffull; v.x1 = (lo >> 37) | ((hi & 0x3ff) << 27); v.size = (hi >> 10) & 0x1fffffffffull; v.info = hi >> 47;
        return v;
    }
    __device__ Intv get(int i) const { return unpack(a[(size_t)i * stride]); }
    __device__ uint4 raw(int i) const { return a[(size_t)i * stride]; }
    __device__ void set_raw(int i, const uint4& v) { a[(size_t)i * stride] = v; }
    __device__ bool push(const Intv& v) { if (n >= cap) return false; a[(size_t)n * stride] = pack(v); ++n; return true; }
};

DEV void pvec_reverse(PackedVec& v)
{
    for (int i = 0, j = v.n - 1; i < j; ++i, --j) { uint4 t = v.raw(i); v.set_raw(i, v.raw(j)); v.set_raw(j, t); }
}

DEV void vec_reverse(IntvVec& v, int from = 0)
{
    for (int i = from, j = v.n - 1; i < j; ++i, --j) { Intv t = v.get(i); v.set(i, v.get(j)); v.set(j, t); }
}

// all SMEMs through position x with interval size >= min_intv, appended to mem when at least min_seed_len long
// (the caller's filter, fused so no intermediate vector is needed); returns the next x.
// Forward-extend recording each size change, then backward-extend every candidate in lock-step,
// longest first (App. B "SMEM(x, min_intv)").
DEV int smem1(const DevIndex& ix, int len, const uint8_t* q, int x, int min_intv, int min_seed_len,
              IntvVec& mem, PackedVec& v0, PackedVec& v1, uint32_t& n_ext, bool& ovf)
{
    Intv ik, ok;
    PackedVec *prev = &v0, *curr = &v1, *swap;
    int i, j, c;
    const int mem0 = mem.n;
    uint64_t last_start = ~0ull;                 // start of the most recently emitted match (before the length filter)
    bool any = false;
    if (q[x] > 3) return x + 1;
    if (min_intv < 1) min_intv = 1;
    set_intv(ix, q[x], ik);
    ik.info = (uint64_t)(x + 1);
    for (i = x + 1, curr->n = 0; i < len; ++i) {
        if (q[i] < 4) {
            c = 3 - q[i];
            ok = extend_one(ix, ik, c, 0); ++n_ext;
            if (ok.size != ik.size) {
                if (!curr->push(ik)) { ovf = true; return len; }
                if (ok.size < (uint64_t)min_intv) break;
            }
            ik = ok; ik.info = (uint64_t)(i + 1);
        } else {
            if (!curr->push(ik)) { ovf = true; return len; }
            break;
        }
    }
    if (i == len) { if (!curr->push(ik)) { ovf = true; return len; } }
    pvec_reverse(*curr);
    int ret = (int)curr->get(0).info;
    swap = curr; curr = prev; prev = swap;
    for (i = x - 1; i >= -1; --i) {
        c = i < 0 ? -1 : q[i] < 4 ? q[i] : -1;
        for (j = 0, curr->n = 0; j < prev->n; ++j) {
            Intv p = prev->get(j);
            if (c >= 0) { ok = extend_one(ix, p, c, 1); ++n_ext; }
            if (c < 0 || ok.size < (uint64_t)min_intv) {
                if (curr->n == 0) {
                    if (!any || (uint64_t)(i + 1) < last_start) {
                        any = true; last_start = (uint64_t)(i + 1);
                        ik = p; ik.info |= (uint64_t)(i + 1) << 32;
                        if ((int)((uint32_t)ik.info - (uint32_t)(ik.info >> 32)) >= min_seed_len) { if (!mem.push(ik)) { ovf = true; return len; } }
                    }
                }
            } else if (curr->n == 0 || ok.size != curr->get(curr->n - 1).size) {
                ok.info = p.info;
                if (!curr->push(ok)) { ovf = true; return len; }
            }
        }
        if (curr->n == 0) break;
        swap = curr; curr = prev; prev = swap;
    }
    vec_reverse(mem, mem0);                       // this call's matches in order of start
    return ret;
}

// pass 3: greedy forward seed (row a5)
DEV int seed_strategy1(const DevIndex& ix, int len, const uint8_t* q, int x, int min_len, int max_intv, Intv& mem, uint32_t& n_ext)
{
    Intv ik, ok;
    mem.x0 = mem.x1 = mem.size = mem.info = 0;
    if (q[x] > 3) return x + 1;
    set_intv(ix, q[x], ik);
    for (int i = x + 1; i < len; ++i) {
        if (q[i] < 4) {
            int c = 3 - q[i];
            ok = extend_one(ix, ik, c, 0); ++n_ext;
            if (ok.size < (uint64_t)(int64_t)max_intv && i - x >= min_len) {
                mem = ok;
                mem.info = (uint64_t)x << 32 | (uint32_t)(i + 1);
                return i + 1;
            }
            ik = ok;
        } else return i + 1;
    }
    return len;
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

// mem_collect_intv (row a6) + the per-read bookkeeping mem_chain does before looking up the SA:
// l_rep (repetitive fraction numerator) and the number of occurrences each interval contributes.
// LDSQ: the 64 reads of the block are staged in LDS with one coalesced copy, so the per-step base look-ups of the
// search do not compete with the occ gathers for vector-memory requests (used when 64 reads fit 24 KB).
template <bool LDSQ>
__global__ void __launch_bounds__(64) k_seed(DevIndex ix, MemOpt opt, TileView tv)
{
    HIP_DYNAMIC_SHARED(uint8_t, sq)
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n_ext = 0;
    int64_t base_off = 0;
    if (LDSQ) {
        const int r0 = blockIdx.x * blockDim.x;
        const int r1 = r0 + 64 < tv.n_reads ? r0 + 64 : tv.n_reads;
        base_off = tv.seq_off[r0];
        const int nbytes = (int)(tv.seq_off[r1] - base_off);
        for (int k = threadIdx.x; k < nbytes; k += 64) sq[k] = tv.seq[base_off + k];
        __syncthreads();
    }
    if (r < tv.n_reads) {
        const uint8_t* q = LDSQ ? (const uint8_t*)sq + (tv.seq_off[r] - base_off) : tv.seq + tv.seq_off[r];
        int len = (int)(tv.seq_off[r + 1] - tv.seq_off[r] - 1);
        // scratch of wave-group g = r/64: [2 vectors][smem_cap entries][64 lanes]
        uint4* sc = (uint4*)tv.smem_scratch + ((size_t)(r >> 6) * 2 * tv.smem_cap) * 64 + (r & 63);
        PackedVec v0 = { sc, 0, tv.smem_cap, 64 };
        PackedVec v1 = { sc + (size_t)tv.smem_cap * 64, 0, tv.smem_cap, 64 };
        IntvVec mem = { tv.intv + (size_t)r * tv.intv_cap, 0, tv.intv_cap, 1 };
        bool ovf = false;
        int n_seeds = 0, l_rep = 0;
        if (len >= opt.min_seed_len) {
            int x = 0;
            int split_len = (int)(opt.min_seed_len * opt.split_factor + .499);
            while (x < len && !ovf) {                       // pass 1: all SMEMs
                if (q[x] < 4) x = smem1(ix, len, q, x, 1, opt.min_seed_len, mem, v0, v1, n_ext, ovf);
                else ++x;
            }
            int old_n = mem.n;
            for (int k = 0; k < old_n && !ovf; ++k) {       // pass 2: re-seed long, rare SMEMs
                Intv p = mem.get(k);
                int start = (int)(p.info >> 32), end = (int)(int32_t)p.info;
                if (end - start < split_len || p.size > (uint64_t)(int64_t)opt.split_width) continue;
                smem1(ix, len, q, (start + end) >> 1, (int)(p.size + 1), opt.min_seed_len, mem, v0, v1, n_ext, ovf);
            }
            if (opt.max_mem_intv > 0) {                     // pass 3: greedy forward seeds
                x = 0;
                while (x < len && !ovf) {
                    if (q[x] < 4) {
                        Intv m;
                        x = seed_strategy1(ix, len, q, x, opt.min_seed_len, (int)opt.max_mem_intv, m, n_ext);
                        if (m.size > 0) { if (!mem.push(m)) ovf = true; }
                    } else ++x;
                }
            }
            if (!ovf) {
                // sort by info.  Intervals with equal info are the same substring, hence identical records, so the
                // order upstream's unstable sort leaves them in is unobservable: a plain insertion sort suffices.
                for (int i = 1; i < mem.n; ++i) {
                    Intv t = mem.a[i];
                    int j = i;
                    while (j > 0 && mem.a[j - 1].info > t.info) { mem.a[j] = mem.a[j - 1]; --j; }
                    mem.a[j] = t;
                }
                int b = 0, e = 0;
                int32_t* iso = tv.intv_seed_off + (size_t)r * tv.intv_cap;
                for (int i = 0; i < mem.n; ++i) {
                    Intv p = mem.a[i];
                    int sb = (int)(p.info >> 32), se = (int)(uint32_t)p.info;
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
            }
        }
        if (ovf) { atomicOr(tv.err, ERR_INTV_CAP); mem.n = 0; n_seeds = 0; l_rep = 0; }
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
    const uint64_t lo = (C & 0xffffffffffull) | (G << 40), hi = ((G >> 24) & 0xffff) | ((T & 0xffffffffffull) << 16);
    uint4 c, s;
    c.x = (uint32_t)lo; c.y = (uint32_t)(lo >> 32); c.z = (uint32_t)hi; c.w = (uint32_t)(hi >> 32);
    s.x = sym[0]; s.y = sym[1]; s.z = sym[2]; s.w = sym[3];
    occ[2 * b] = c; occ[2 * b + 1] = s;
}

void launch_build_occ64(hipStream_t st, const uint32_t* bwt, uint64_t n_blocks, uint4* occ)
{
    hipLaunchKernelGGL(k_build_occ64, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, st, bwt, n_blocks, occ);
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
    const size_t lds = (size_t)64 * ((size_t)tv.max_len + 1) + 64;
    if (lds <= 24576) hipLaunchKernelGGL(k_seed<true>, dim3((tv.n_reads + 63) / 64), dim3(64), lds, st, ix, opt, tv);
    else hipLaunchKernelGGL(k_seed<false>, dim3((tv.n_reads + 63) / 64), dim3(64), 0, st, ix, opt, tv);
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
