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

struct GLds { int32_t* eh_h; int32_t* eh_e; int32_t* tmpM; };

// one ksw_global2 call; the raw CIGAR (before clip / deletion squeezing) goes to cigar[0..*n_cigar)
static __device__ int global_wave(const DevIndex& ix, const MemOpt& opt, const GLds& L, int lane, const SeqAcc& A, int w,
                                  uint8_t* z, int n_col, uint32_t* cigar, int cig_cap, int* n_cigar, int& err)
{
    const int qlen = A.qlen, tlen = A.tlen;
    const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    for (int j = lane; j <= qlen; j += WAVE) {
        L.eh_h[j] = j == 0 ? 0 : (j <= w ? -(o_ins + e_ins * j) : MINUS_INF);
        L.eh_e[j] = MINUS_INF;
    }
    int tch = 4;
    __syncthreads();
    for (int i = 0; i < tlen; ++i) {
        if ((i & 63) == 0) { int ii = i + lane; tch = ii < tlen ? acc_t(ix, A, ii) : 4; }
        const int tb = wave_bcast(tch, i & 63);
        const int ms0 = opt.mat[tb * 5], ms1 = opt.mat[tb * 5 + 1], ms2 = opt.mat[tb * 5 + 2], ms3 = opt.mat[tb * 5 + 3], ms4 = opt.mat[tb * 5 + 4];
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        const int h1i = beg == 0 ? -(o_del + e_del * (i + 1)) : MINUS_INF;
        for (int c = beg; c < end; c += WAVE) {                     // phase A: M(i,j) from row i-1
            int j = c + lane;
            if (j < end) {
                int qc = acc_q(A, j);
                int sc = qc == 0 ? ms0 : qc == 1 ? ms1 : qc == 2 ? ms2 : qc == 3 ? ms3 : ms4;
                L.tmpM[j] = L.eh_h[j] + sc;
            }
        }
        __syncthreads();
        int fcarry = MINUS_INF, hlast = h1i;
        uint8_t* zi = z + (int64_t)i * n_col;
        for (int c = beg; c < end; c += WAVE) {                     // phase B: F by scan, H, E, direction bits
            int j = c + lane;
            bool act = j < end;
            int m = act ? L.tmpM[j] : 0;
            int e = act ? L.eh_e[j] : 0;
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
            if (act) { zi[j - beg] = (uint8_t)d; L.eh_e[j] = e2; L.eh_h[j + 1] = h; }
        }
        if (lane == 0) {
            if (end > beg) L.eh_h[beg] = h1i;
            else L.eh_h[end] = h1i;
            L.eh_e[end] = MINUS_INF;
        }
        (void)hlast;
        __syncthreads();
    }
    const int score = L.eh_h[qlen];
    int n = 0;
    if (lane == 0) {                                                 // backtrack; ops come out end-to-start
        int which = 0, i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
        bool ovf = false;
        while (i >= 0 && k >= 0 && !ovf) {
            int op;
            which = z[(int64_t)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
            if (which == 0) { op = 0; --i; --k; }
            else if (which == 1) { op = 2; --i; }
            else { op = 1; --k; }
            if (n == 0 || op != (int)(cigar[n - 1] & 0xf)) { if (n >= cig_cap) ovf = true; else cigar[n++] = 1u << 4 | (uint32_t)op; }
            else cigar[n - 1] += 1u << 4;
        }
        if (!ovf && i >= 0) { if (n == 0 || 2 != (int)(cigar[n - 1] & 0xf)) { if (n >= cig_cap) ovf = true; else cigar[n++] = (uint32_t)(i + 1) << 4 | 2; } else cigar[n - 1] += (uint32_t)(i + 1) << 4; }
        if (!ovf && k >= 0) { if (n == 0 || 1 != (int)(cigar[n - 1] & 0xf)) { if (n >= cig_cap) ovf = true; else cigar[n++] = (uint32_t)(k + 1) << 4 | 1; } else cigar[n - 1] += (uint32_t)(k + 1) << 4; }
        for (int a = 0; a < n >> 1; ++a) { uint32_t tmp = cigar[a]; cigar[a] = cigar[n - 1 - a]; cigar[n - 1 - a] = tmp; }
        if (ovf) { err |= ERR_CIGAR_CAP; n = 0; }
    }
    n = wave_bcast(n, 0);
    *n_cigar = n;
    __syncthreads();
    return score;
}

__global__ void __launch_bounds__(64) k_gcigar(DevIndex ix, MemOpt opt, TileView tv, const DpJob* jobs, DpOut* outs, uint32_t* cig_pool, int cig_cap,
                                               uint8_t* zpool, unsigned long long zpool_cap, unsigned long long* zpool_cur, int z_lds_cap)
{
    HIP_DYNAMIC_SHARED(int32_t, smem)
    const int job = blockIdx.x, lane = threadIdx.x;
    const DpJob jb = jobs[job];
    const AlnReg ar = tv.regs[tv.seed_off[jb.read] + jb.reg];
    const uint8_t* query = tv.seq + tv.seq_off[jb.read];
    const int cap = tv.max_len + 2;
    GLds L; L.eh_h = smem; L.eh_e = smem + cap; L.tmpM = smem + 2 * cap;
    uint8_t* z_lds = (uint8_t*)(smem + 3 * cap);
    int err = 0;
    SeqAcc A; A.q = query + ar.qb; A.qlen = ar.qe - ar.qb; A.rev = ar.rb >= ix.l_pac; A.t0 = ar.rb; A.tlen = (int)(ar.re - ar.rb);
    uint32_t* cigar = cig_pool + (size_t)job * cig_cap;
    // the retry loop of mem_reg2aln around bwa_gen_cigar2 (all lanes take the same decisions)
    int w2 = first_w2(opt, ar), i = 0, last_sc = -(1 << 30), score = 0, n_cigar = 0;
    const int l_query = A.qlen, rlen = A.tlen;
    const bool usable = !(l_query <= 0 || ar.rb >= ar.re || (ar.rb < ix.l_pac && ar.re > ix.l_pac) || ar.re > ix.l_pac << 1 || ar.rb < 0);
    if (usable) {
        do {
            w2 = w2 < opt.w << 2 ? w2 : opt.w << 2;
            int w, max_gap, max_ins, max_del, min_w, d;                // band of bwa_gen_cigar2
            max_ins = (int)((double)(((l_query + 1) >> 1) * opt.mat[0] - opt.o_ins) / opt.e_ins + 1.);
            max_del = (int)((double)(((l_query + 1) >> 1) * opt.mat[0] - opt.o_del) / opt.e_del + 1.);
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
            if (need > (unsigned long long)z_lds_cap) {                  // traceback matrix too big for LDS: bump-allocate HBM
                unsigned long long at = 0;
                if (lane == 0) at = atomicAdd(zpool_cur, (need + 63ull) & ~63ull);
                at = __shfl(at, 0);
                if (at + need > zpool_cap) { err |= ERR_ZPOOL; break; }
                z = zpool + at;
            }
            score = global_wave(ix, opt, L, lane, A, w, z, n_col, cigar, cig_cap, &n_cigar, err);
            if (score == last_sc || w2 == opt.w << 2) break;
            last_sc = score;
            w2 <<= 1;
        } while (++i < 3 && score < ar.truesc - opt.a);
    }
    if (lane == 0) {
        DpOut o; o.score = score; o.n_cigar = n_cigar;
        outs[job] = o;
        if (err) atomicOr(tv.err, err);
    }
}

void launch_gcigar(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, int n_jobs, const void* jobs, void* outs, uint32_t* cig_pool, int cig_cap,
                   uint8_t* zpool, unsigned long long zpool_cap, unsigned long long* zpool_cur)
{
    if (n_jobs <= 0) return;
    const int z_lds_cap = 16384;
    size_t cap = (size_t)tv.max_len + 2;
    size_t shmem = 3 * cap * sizeof(int32_t) + (size_t)z_lds_cap + 64;
    hipLaunchKernelGGL(k_gcigar, dim3(n_jobs), dim3(64), shmem, st, ix, opt, tv, (const DpJob*)jobs, (DpOut*)outs, cig_pool, cig_cap, zpool, zpool_cap, zpool_cur, z_lds_cap);
}
