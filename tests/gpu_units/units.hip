// tests/gpu_units/units.hip -- TEST INFRASTRUCTURE: thin host-callable wrappers that run single device
// functions of the product (introsort, the wave-parallel extension DP, rank queries) on caller-supplied
// inputs, so tests can compare them with the oracle one function at a time.  Built by the test fixtures
// (hipcc on the GPU box, the emulation build on CPU); never part of libbwamem_hip.so.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../gatk-bwamem-jni_amd/csrc/k_extend.hip"
#include "../../gatk-bwamem-jni_amd/csrc/k_pe.hip"
#include "../../gatk-bwamem-jni_amd/csrc/k_cigar.hip"
#include "../../gatk-bwamem-jni_amd/csrc/chain_flt.h"
#include "../../gatk-bwamem-jni_amd/csrc/post_common.h"

struct PairX { uint64_t x, y; };
struct PairXLt { __device__ bool operator()(const PairX& a, const PairX& b) const { return a.x < b.x; } };

__global__ void k_unit_sort_pairs(PairX* a, int n) { if (threadIdx.x == 0 && blockIdx.x == 0) ks_introsort((size_t)n, a, PairXLt()); }

__global__ void __launch_bounds__(64) k_unit_extend(DevIndex ix, MemOpt opt, const uint8_t* query, int qlen, int tlen, int w, int end_bonus, int zdrop, int h0, int* out, int force_lds)
{
    HIP_DYNAMIC_SHARED(int32_t, smem)
    const int cap = qlen + 2, lane = threadIdx.x;
    ExtLds L;
    const int ring = extend_ring(opt, qlen);                    // the row rings, sized as production sizes them
    L.eh_h = smem; L.eh_e = smem + ring; L.tmpM = smem + 2 * ring; L.rm = ring - 1;
    uint8_t* sq = (uint8_t*)(smem + 3 * ring);
    (void)cap;
    L.query = sq;
    for (int j = lane; j < qlen; j += WAVE) sq[j] = query[j];
    __syncthreads();
    unsigned long long n_cells = 0;
    // 0: the 32-bit register forms; 1: the general LDS form; 2: the production entry (diagonal certificate first, packed 16-bit form
    // for two-chunk queries); 3: the production entry without the certificate (so that the packed form sees every case it accepts)
    // 4: the packed form with its rows in LDS (what queries beyond 191 bases take) on whatever query it accepts, the general form otherwise
    ExtRes r = force_lds == 1 ? extend_wave(ix, opt, L, lane, qlen, 0, 1, tlen, 0, 1, w, end_bonus, zdrop, h0, n_cells)
             : force_lds == 4 ? (extend_pkl_ok(opt, L.rm + 1, qlen, w, h0, score_max(opt)) ? extend_wave_pkl(ix, opt, L, lane, qlen, 0, 1, tlen, 0, 1, w, end_bonus, zdrop, h0, n_cells)
                                                                                             : extend_wave(ix, opt, L, lane, qlen, 0, 1, tlen, 0, 1, w, end_bonus, zdrop, h0, n_cells))
                         : extend_any(ix, opt, L, lane, qlen, 0, 1, tlen, 0, 1, w, end_bonus, zdrop, h0, n_cells, force_lds == 2, force_lds >= 2);
    if (lane == 0) { out[0] = r.score; out[1] = r.qle; out[2] = r.tle; out[3] = r.gtle; out[4] = r.gscore; out[5] = r.max_off; }
}

__global__ void k_unit_occ(DevIndex ix, const uint64_t* ks, int n, uint64_t* out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { uint64_t c[4]; occ4(ix, ks[i], c); for (int j = 0; j < 4; ++j) out[4 * i + j] = c[j]; }
}

extern "C" int unit_sort_pairs(int n, uint64_t* xy)
{
    PairX* d;
    if (hipMalloc((void**)&d, (size_t)n * 16 + 16) != hipSuccess) return -1;
    hipMemcpy(d, xy, (size_t)n * 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_unit_sort_pairs, dim3(1), dim3(64), 0, 0, d, n);
    int rc = hipDeviceSynchronize() == hipSuccess ? 0 : -1;
    hipMemcpy(xy, d, (size_t)n * 16, hipMemcpyDeviceToHost);
    hipFree(d);
    return rc;
}

// target is packed 2 bit/base into a throw-away "reference" so extend_wave reads it exactly as in production
extern "C" int unit_extend(const uint8_t* query, int qlen, const uint8_t* target, int tlen, const MemOpt* opt,
                           int w, int end_bonus, int zdrop, int h0, int* out6, int force_lds)
{
    std::vector<uint8_t> pac((size_t)tlen / 4 + 2, 0);
    for (int i = 0; i < tlen; ++i) pac[i >> 2] |= (uint8_t)(target[i] << ((~i & 3) << 1));
    uint8_t *d_pac, *d_q; int* d_out;
    hipMalloc((void**)&d_pac, pac.size()); hipMalloc((void**)&d_q, (size_t)qlen + 16); hipMalloc((void**)&d_out, 64);
    hipMemcpy(d_pac, pac.data(), pac.size(), hipMemcpyHostToDevice);
    hipMemcpy(d_q, query, (size_t)qlen, hipMemcpyHostToDevice);
    DevIndex ix; memset(&ix, 0, sizeof ix);
    ix.pac = d_pac; ix.l_pac = tlen;
    size_t cap = (size_t)qlen + 2, shmem = 3 * (size_t)extend_ring(*opt, qlen) * 4 + ((cap + 15) & ~(size_t)15);
    hipLaunchKernelGGL(k_unit_extend, dim3(1), dim3(64), shmem, 0, ix, *opt, d_q, qlen, tlen, w, end_bonus, zdrop, h0, d_out, force_lds);
    int rc = hipDeviceSynchronize() == hipSuccess ? 0 : -1;
    hipMemcpy(out6, d_out, 24, hipMemcpyDeviceToHost);
    hipFree(d_pac); hipFree(d_q); hipFree(d_out);
    return rc;
}

// mem_chain_flt's overlap loop in its two forms (k_chain.hip): wave != 0 -> chain_flt_wave by all 64 lanes, else chain_flt_lane on lane 0
__global__ void __launch_bounds__(64) k_unit_chain_flt(MemOpt opt, const Seed* seeds, Chain* a, int n_chn, int4* kept, int wave, int* n_kept)
{
    int got = 0;
    if (wave) got = chain_flt_wave(opt, seeds, a, n_chn, kept);
    else if (threadIdx.x == 0) got = chain_flt_lane(opt, seeds, a, n_chn, kept);
    if (threadIdx.x == 0) *n_kept = got;
}

// chains given as (query begin, query end, weight, is_alt), sorted by falling weight by the caller.
// out_kept[i] = Chain::kept after the loop, out_first[k] = first shadowed chain of the k-th kept chain; returns n_kept
extern "C" int unit_chain_flt(const MemOpt* opt, int n, const int32_t* qb, const int32_t* qe, const int32_t* w, const int32_t* alt, int wave, int32_t* out_kept, int32_t* out_first)
{
    std::vector<Seed> seeds((size_t)n); std::vector<Chain> a((size_t)n);
    memset(seeds.data(), 0, seeds.size() * sizeof(Seed)); memset(a.data(), 0, a.size() * sizeof(Chain));
    for (int i = 0; i < n; ++i) {
        seeds[i].qbeg = qb[i]; seeds[i].len = qe[i] - qb[i]; seeds[i].next = -1;
        a[i].seed0 = a[i].last = i; a[i].n = 1; a[i].w = (uint32_t)w[i]; a[i].is_alt = alt[i]; a[i].first = -1; a[i].kept = 0;
    }
    Seed* d_s; Chain* d_a; int4* d_k; int* d_n;
    hipMalloc((void**)&d_s, (size_t)n * sizeof(Seed) + 16); hipMalloc((void**)&d_a, (size_t)n * sizeof(Chain) + 16);
    hipMalloc((void**)&d_k, (size_t)n * 16 + 16); hipMalloc((void**)&d_n, 16);
    hipMemcpy(d_s, seeds.data(), (size_t)n * sizeof(Seed), hipMemcpyHostToDevice);
    hipMemcpy(d_a, a.data(), (size_t)n * sizeof(Chain), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_unit_chain_flt, dim3(1), dim3(64), 0, 0, *opt, (const Seed*)d_s, d_a, n, d_k, wave, d_n);
    int rc = hipDeviceSynchronize() == hipSuccess ? 0 : -1, n_kept = 0;
    hipMemcpy(&n_kept, d_n, 4, hipMemcpyDeviceToHost);
    hipMemcpy(a.data(), d_a, (size_t)n * sizeof(Chain), hipMemcpyDeviceToHost);
    std::vector<int4> k((size_t)n);
    hipMemcpy(k.data(), d_k, (size_t)n * 16, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) out_kept[i] = a[i].kept;
    for (int i = 0; i < n_kept && i < n; ++i) out_first[i] = k[i].w;
    hipFree(d_s); hipFree(d_a); hipFree(d_k); hipFree(d_n);
    return rc ? rc : n_kept;
}

// sort_regs (post_common.h): the four region sorts, on the regions themselves (by_key == 0) or through key records
__global__ void k_unit_sort_regs(AlnReg* a, int n, int which, SortKey* keys)
{
    if (threadIdx.x || blockIdx.x) return;
    if (which == 0) sort_regs(n, a, keys, RegReLt());
    else if (which == 1) sort_regs(n, a, keys, RegSLt());
    else if (which == 2) sort_regs(n, a, keys, RegHLt());
    else sort_regs(n, a, keys, RegHLt2());
}
extern "C" int unit_sizeof_alnreg() { return (int)sizeof(AlnReg); }
extern "C" int unit_sort_regs(int n, void* regs, int which, int by_key)
{
    AlnReg* d; SortKey* k;
    hipMalloc((void**)&d, (size_t)n * sizeof(AlnReg) + 16); hipMalloc((void**)&k, (size_t)n * sizeof(SortKey) + 16);
    hipMemcpy(d, regs, (size_t)n * sizeof(AlnReg), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_unit_sort_regs, dim3(1), dim3(64), 0, 0, d, n, which, by_key ? k : (SortKey*)0);
    int rc = hipDeviceSynchronize() == hipSuccess ? 0 : -1;
    hipMemcpy(regs, d, (size_t)n * sizeof(AlnReg), hipMemcpyDeviceToHost);
    hipFree(d); hipFree(k);
    return rc;
}

// mem_matesw's maintenance of the mate's hit list (k_pe.hip: matesw), one rescued region after the other: incr == 0 is
// upstream's sequence (insert behind the regions that score at least as high, then mem_sort_dedup_patch without a query);
// incr == 1 goes through matesw_insert once the list has been through one such call.  stat: final n, declined, full calls
// incr == 2: the wavefront forms (matesw_insert_wave, regs_insert_wave, sort_dedup_nq_wave), all 64 lanes
__global__ void __launch_bounds__(64) k_unit_matesw_list_wave(MemOpt opt, AlnReg* ma, int n0, const AlnReg* add, int n_add, SortKey* keys, int* stat)
{
    const int lane = threadIdx.x;
    int n = n0, declined = 0, full = 0;
    bool settled = false;
    for (int k = 0; k < n_add; ++k) {
        const AlnReg b = add[k];
        if (!(settled && n >= 1 && matesw_insert_wave(opt, b, n, ma, lane))) {
            if (settled && n >= 1) ++declined;
            int pos = n;
            for (int base = 0; base < n; base += 64) {
                const int kk = base + lane;
                const uint64_t lower = __ballot(kk < n && ma[kk].score < b.score);
                if (lower) { pos = base + __ffsll((long long)lower) - 1; break; }
            }
            regs_insert_wave(ma, n, pos, b, lane);
            ++n;
            settled = n >= 2;
            ++full;
            n = sort_dedup_nq_wave(opt, n, ma, keys, lane);
        }
    }
    if (lane == 0) { stat[0] = n; stat[1] = declined; stat[2] = full; }
}
__global__ void k_unit_matesw_list(MemOpt opt, AlnReg* ma, int n0, const AlnReg* add, int n_add, int incr, SortKey* keys, int* stat)
{
    if (threadIdx.x || blockIdx.x) return;
    DevIndex ix; memset(&ix, 0, sizeof ix);
    PostScratch S; memset(&S, 0, sizeof S);
    int n = n0, declined = 0, full = 0;
    bool settled = false;
    for (int k = 0; k < n_add; ++k) {
        AlnReg b = add[k];
        if (!(incr && settled && n >= 1 && matesw_insert(opt, b, n, ma))) {
            if (incr && settled && n >= 1) ++declined;
            ++n;
            int i;
            for (i = 0; i < n - 1; ++i) if (ma[i].score < b.score) break;
            for (int j = n - 1; j > i; --j) ma[j] = ma[j - 1];
            ma[i] = b;
            settled = n >= 2;
            ++full;
            n = sort_dedup_patch(ix, opt, S, 0, n, ma, 0, keys);
        }
    }
    stat[0] = n; stat[1] = declined; stat[2] = full;
}
extern "C" int unit_matesw_list(const MemOpt* opt, void* regs, int n0, const void* add, int n_add, int incr, int* stat3)
{
    AlnReg *d, *da; SortKey* k; int* ds;
    hipMalloc((void**)&d, (size_t)(n0 + n_add) * sizeof(AlnReg) + 16); hipMalloc((void**)&da, (size_t)n_add * sizeof(AlnReg) + 16);
    hipMalloc((void**)&k, (size_t)(n0 + n_add) * sizeof(SortKey) + 16); hipMalloc((void**)&ds, 16);
    hipMemcpy(d, regs, (size_t)n0 * sizeof(AlnReg), hipMemcpyHostToDevice);
    hipMemcpy(da, add, (size_t)n_add * sizeof(AlnReg), hipMemcpyHostToDevice);
    if (incr == 2) hipLaunchKernelGGL(k_unit_matesw_list_wave, dim3(1), dim3(64), 0, 0, *opt, d, n0, (const AlnReg*)da, n_add, k, ds);
    else hipLaunchKernelGGL(k_unit_matesw_list, dim3(1), dim3(64), 0, 0, *opt, d, n0, (const AlnReg*)da, n_add, incr, k, ds);
    int rc = hipDeviceSynchronize() == hipSuccess ? 0 : -1;
    hipMemcpy(stat3, ds, 12, hipMemcpyDeviceToHost);
    hipMemcpy(regs, d, (size_t)(n0 + n_add) * sizeof(AlnReg), hipMemcpyDeviceToHost);
    hipFree(d); hipFree(da); hipFree(k); hipFree(ds);
    return rc;
}


// ksw_align2 as mate rescue and seed re-scoring run it (k_pe.hip: launch_sw_jobs): n jobs, job i = query i (codes 0..4, qoff[i] ..
// qoff[i+1]) against tlen[i] reference bases from toff[i] of a throw-away packed "reference"; xtra as mem_matesw builds it.
// out: 7 ints per job (score, te, qe, score2, te2, tb, qb)
extern "C" int unit_sw_jobs(const MemOpt* opt, int n, const uint8_t* queries, const int64_t* qoff, const uint8_t* target, int64_t l_target,
                            const int64_t* toff, const int32_t* tlen, const int32_t* xtra, int32_t* out)
{
    std::vector<uint8_t> pac((size_t)l_target / 4 + 2, 0);
    for (int64_t i = 0; i < l_target; ++i) pac[i >> 2] |= (uint8_t)((target[i] & 3) << ((~i & 3) << 1));
    std::vector<SwJob> jobs((size_t)n);
    int max_q = 0, max_t = 0;
    for (int i = 0; i < n; ++i) {
        SwJob& j = jobs[i];
        j.rb = toff[i]; j.read = i; j.tag = i; j.l_ms = (int)(qoff[i + 1] - qoff[i]); j.is_rev = 0; j.tlen = tlen[i]; j.xtra = xtra[i]; j.q_off = 0; j.pad_ = 0;
        max_q = j.l_ms > max_q ? j.l_ms : max_q; max_t = tlen[i] > max_t ? tlen[i] : max_t;
    }
    uint8_t *d_pac, *d_seq; int64_t* d_off; int32_t *d_err, *d_cnt; SwJob* d_jobs; KswR* d_res;
    hipMalloc((void**)&d_pac, pac.size()); hipMalloc((void**)&d_seq, (size_t)qoff[n] + 64); hipMalloc((void**)&d_off, ((size_t)n + 1) * 8);
    hipMalloc((void**)&d_err, 64); hipMalloc((void**)&d_cnt, 64); hipMalloc((void**)&d_jobs, (size_t)n * sizeof(SwJob) + 64); hipMalloc((void**)&d_res, pe_rescue_bytes(1, n) + 64);
    hipMemcpy(d_pac, pac.data(), pac.size(), hipMemcpyHostToDevice);
    hipMemcpy(d_seq, queries, (size_t)qoff[n], hipMemcpyHostToDevice);
    hipMemcpy(d_off, qoff, ((size_t)n + 1) * 8, hipMemcpyHostToDevice);
    hipMemset(d_err, 0, 64);
    hipMemcpy(d_cnt, &n, 4, hipMemcpyHostToDevice);
    hipMemcpy(d_jobs, jobs.data(), (size_t)n * sizeof(SwJob), hipMemcpyHostToDevice);
    DevIndex ix; memset(&ix, 0, sizeof ix);
    ix.pac = d_pac; ix.l_pac = l_target;
    TileView tv; memset(&tv, 0, sizeof tv);
    tv.n_reads = n; tv.max_len = max_q; tv.seq = d_seq; tv.seq_off = d_off; tv.err = d_err;
    launch_sw_jobs(0, ix, *opt, tv, d_jobs, d_cnt, n, d_res, (max_t + max_q) / 2 + 16 /* as pipeline.cpp sizes it: (span + 2 L) / 2 + 16 */, max_q);
    int rc = hipDeviceSynchronize() == hipSuccess ? 0 : -1;
    int32_t e = 0;
    hipMemcpy(&e, d_err, 4, hipMemcpyDeviceToHost);
    hipMemcpy(out, d_res, (size_t)n * sizeof(KswR), hipMemcpyDeviceToHost);
    hipFree(d_pac); hipFree(d_seq); hipFree(d_off); hipFree(d_err); hipFree(d_cnt); hipFree(d_jobs); hipFree(d_res);
    return rc ? rc : e;
}


// ksw_global2 as k_gcigar's wave forms run it: job i = query i (codes 0..4) against tlen[i] reference bases from toff[i] of a
// throw-away packed "reference" with band w[i]; the direction matrix of a job in its own stretch of global memory, walked through
// tiles staged in LDS (what long reads do).  mode 0: the 32-bit diagonal forms; mode 1: the packed 16-bit form where its fit test
// accepts the job (out_ok = 1), the 32-bit form otherwise (out_ok = 0: refused up front, 2: gave up at a range check).
// range_override > 0 replaces the fit test's range (to drive the give-up path).  out: score, n_cigar per job; cigars [n][cig_cap]
__global__ void __launch_bounds__(64) k_unit_global(DevIndex ix, MemOpt opt, const uint8_t* queries, const int64_t* qoff, const int64_t* toff, const int32_t* tlen, const int32_t* wv,
                                                    int mode, int range_override, uint8_t* zall, const int64_t* zoff, uint32_t* cig, int cig_cap, int32_t* out, int max_q)
{
    HIP_DYNAMIC_SHARED(int32_t, smem)
    const int lane = threadIdx.x, job = blockIdx.x;
    uint8_t* sq = (uint8_t*)smem;
    uint8_t* tile = sq + ((max_q + 2 + 15) & ~15);
    SeqAcc A; A.q = queries + qoff[job]; A.qlen = (int)(qoff[job + 1] - qoff[job]); A.rev = 0; A.t0 = toff[job]; A.tlen = tlen[job];
    const int w = wv[job];
    for (int j = lane; j < A.qlen; j += WAVE) sq[j] = (uint8_t)acc_q(A, j);
    __syncthreads();
    const int n_col = A.qlen < 2 * w + 1 ? A.qlen : 2 * w + 1, nch = (2 * w + 1 + 63) >> 6;
    uint8_t* z = zall + zoff[job];
    int score = 0, okc = 0, err = 0, n_cigar = 0;
    bool done = false;
    GpkFit fit;
    if (mode == 1 && nch >= 2 && nch <= 14 && gpk_fit(opt, w, (nch + 1) >> 1, fit)) {
        if (range_override > 0) fit.range = range_override;
        if (nch <= 2) score = global_wave_diag_pk<1>(ix, opt, sq, lane, A, w, fit, z, false, n_col, done);
        else if (nch <= 4) score = global_wave_diag_pk<2>(ix, opt, sq, lane, A, w, fit, z, false, n_col, done);
        else if (nch <= 8) score = global_wave_diag_pk<4>(ix, opt, sq, lane, A, w, fit, z, false, n_col, done);
        else score = global_wave_diag_pk<7>(ix, opt, sq, lane, A, w, fit, z, false, n_col, done);
        okc = done ? 1 : 2;
    }
    if (!done) {
        if (nch <= 1) score = global_wave_diag(ix, opt, sq, lane, A, w, z, false, n_col);
        else if (nch <= 2) score = global_wave_diag_n<2>(ix, opt, sq, lane, A, w, z, false, n_col);
        else if (nch <= 4) score = global_wave_diag_n<4>(ix, opt, sq, lane, A, w, z, false, n_col);
        else if (nch <= 7) score = global_wave_diag_n<7>(ix, opt, sq, lane, A, w, z, false, n_col);
        else score = global_wave_diag_n<13>(ix, opt, sq, lane, A, w, z, false, n_col);
    }
    GClk K; K.c = nullptr; K.t = 0;
    n_cigar = traceback(z, false, n_col, w, A.tlen, A.qlen, lane, cig + (size_t)job * cig_cap, cig_cap, err, tile, K);
    if (lane == 0) { out[4 * job] = score; out[4 * job + 1] = n_cigar; out[4 * job + 2] = okc; out[4 * job + 3] = err; }
}

extern "C" int unit_global(const MemOpt* opt, int n, const uint8_t* queries, const int64_t* qoff, const uint8_t* target, int64_t l_target,
                           const int64_t* toff, const int32_t* tlen, const int32_t* w, int mode, int range_override, int cig_cap, int32_t* out, uint32_t* cigars)
{
    std::vector<uint8_t> pac((size_t)l_target / 4 + 2, 0);
    for (int64_t i = 0; i < l_target; ++i) pac[i >> 2] |= (uint8_t)((target[i] & 3) << ((~i & 3) << 1));
    std::vector<int64_t> zoff((size_t)n + 1, 0);
    int max_q = 0;
    for (int i = 0; i < n; ++i) {
        const int ql = (int)(qoff[i + 1] - qoff[i]);
        max_q = ql > max_q ? ql : max_q;
        zoff[i + 1] = zoff[i] + (((int64_t)(2 * w[i] + 1) * tlen[i] + 255) & ~(int64_t)255);
    }
    uint8_t *d_pac, *d_seq, *d_z; int64_t *d_qoff, *d_toff, *d_zoff; int32_t *d_tlen, *d_w, *d_out; uint32_t* d_cig;
    hipMalloc((void**)&d_pac, pac.size()); hipMalloc((void**)&d_seq, (size_t)qoff[n] + 64); hipMalloc((void**)&d_z, (size_t)zoff[n] + 256);
    hipMalloc((void**)&d_qoff, ((size_t)n + 1) * 8); hipMalloc((void**)&d_toff, (size_t)n * 8 + 8); hipMalloc((void**)&d_zoff, ((size_t)n + 1) * 8);
    hipMalloc((void**)&d_tlen, (size_t)n * 4 + 4); hipMalloc((void**)&d_w, (size_t)n * 4 + 4); hipMalloc((void**)&d_out, (size_t)n * 16 + 16); hipMalloc((void**)&d_cig, (size_t)n * cig_cap * 4 + 16);
    hipMemcpy(d_pac, pac.data(), pac.size(), hipMemcpyHostToDevice);
    hipMemcpy(d_seq, queries, (size_t)qoff[n], hipMemcpyHostToDevice);
    hipMemcpy(d_qoff, qoff, ((size_t)n + 1) * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_toff, toff, (size_t)n * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_zoff, zoff.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_tlen, tlen, (size_t)n * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_w, w, (size_t)n * 4, hipMemcpyHostToDevice);
    hipMemset(d_out, 0, (size_t)n * 16);
    DevIndex ix; memset(&ix, 0, sizeof ix);
    ix.pac = d_pac; ix.l_pac = l_target;
    const size_t shmem = (((size_t)max_q + 2 + 15) & ~(size_t)15) + 4096 + 64;
    hipLaunchKernelGGL(k_unit_global, dim3(n), dim3(64), shmem, 0, ix, *opt, d_seq, d_qoff, d_toff, d_tlen, d_w, mode, range_override, d_z, d_zoff, d_cig, cig_cap, d_out, max_q);
    int rc = hipDeviceSynchronize() == hipSuccess ? 0 : -1;
    hipMemcpy(out, d_out, (size_t)n * 16, hipMemcpyDeviceToHost);
    hipMemcpy(cigars, d_cig, (size_t)n * cig_cap * 4, hipMemcpyDeviceToHost);
    hipFree(d_pac); hipFree(d_seq); hipFree(d_z); hipFree(d_qoff); hipFree(d_toff); hipFree(d_zoff); hipFree(d_tlen); hipFree(d_w); hipFree(d_out); hipFree(d_cig);
    return rc;
}
