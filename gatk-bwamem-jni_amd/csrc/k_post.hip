// k_post.hip -- region post-processing and record generation, one lane per read.
//
// Replaces, for the reference call at jnibwa.c:214, upstream bwamem.c mem_sort_dedup_patch /
// mem_patch_reg (row a13), mem_mark_primary_se (a14), mem_reg2aln + bwa.c bwa_gen_cigar2 +
// ksw.c ksw_global2 (a15), mem_approx_mapq_se (a16), mem_reg2sam + bwamem_extra.c mem_gen_alt
// (a17), and the reference's own record hook fmt_BAMish / bufLen (jnibwa.c:43-124, a18).
// These stages are branchy and tiny next to seeding and extension; they stay on the device
// so a batch never round-trips through the host between kernels.
#include "dev_common.h"
#include "kernels.h"
#include "post_common.h"

// ------------------------------------------------------------------ sort comparators
struct RegReLt { __device__ bool operator()(const AlnReg& a, const AlnReg& b) const { return a.re < b.re; } };
struct RegSLt  { __device__ bool operator()(const AlnReg& a, const AlnReg& b) const {
    return a.score > b.score || (a.score == b.score && (a.rb < b.rb || (a.rb == b.rb && a.qb < b.qb))); } };

// mem_patch_reg: can two colinear regions be merged into one global alignment?
DEV int patch_reg(const DevIndex& ix, const MemOpt& opt, PostScratch& S, const uint8_t* query, const AlnReg& a, const AlnReg& b, int* _w)
{
    if (a.rb < ix.l_pac && b.rb >= ix.l_pac) return 0;
    if (a.qb >= b.qb || a.qe >= b.qe || a.re >= b.re) return 0;
    int w = (int)((a.re - b.rb) - (a.qe - b.qb));
    w = w > 0 ? w : -w;
    double r = (double)(a.re - b.rb) / (b.re - a.rb) - (double)(a.qe - b.qb) / (b.qe - a.qb);
    r = r > 0. ? r : -r;
    if (a.re < b.rb || a.qe < b.qb) {
        if (w > opt.w << 1 || r >= 0.05f) return 0;
    } else if (w > opt.w << 2 || r >= 0.05f * 2) return 0;
    w += a.w + b.w;
    w = w < opt.w << 2 ? w : opt.w << 2;
    int score = 0;
    gen_cigar2(ix, opt, S, w, b.qe - a.qb, query + a.qb, a.rb, b.re, &score, false, 0, 0);
    int q_s = (int)((double)(b.qe - a.qb) / ((b.qe - b.qb) + (a.qe - a.qb)) * (b.score + a.score) + .499);
    int r_s = (int)((double)(b.re - a.rb) / ((b.re - b.rb) + (a.re - a.rb)) * (b.score + a.score) + .499);
    if ((double)score / (q_s > r_s ? q_s : r_s) < 0.90f) return 0;
    *_w = w;
    return score;
}

// mem_sort_dedup_patch; query == 0 disables patching (the mate-rescue caller)
__device__ int sort_dedup_patch(const DevIndex& ix, const MemOpt& opt, PostScratch& S, const uint8_t* query, int n, AlnReg* a)
{
    int m, i, j;
    if (n <= 1) return n;
    ks_introsort((size_t)n, a, RegReLt());
    for (i = 0; i < n; ++i) a[i].n_comp = 1;
    for (i = 1; i < n; ++i) {
        AlnReg* p = &a[i];
        if (p->rid != a[i - 1].rid || p->rb >= a[i - 1].re + opt.max_chain_gap) continue;
        for (j = i - 1; j >= 0 && p->rid == a[j].rid && p->rb < a[j].re + opt.max_chain_gap; --j) {
            AlnReg* q = &a[j];
            int64_t orr, oq, mr, mq;
            int score, w;
            if (q->qe == q->qb) continue;
            orr = q->re - p->rb;
            oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
            mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
            mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
            if ((float)orr > opt.mask_level_redun * (float)mr && (float)oq > opt.mask_level_redun * (float)mq) {
                if (p->score < q->score) { p->qe = p->qb; break; }
                else q->qe = q->qb;
            } else if (query && q->rb < p->rb && (score = patch_reg(ix, opt, S, query, *q, *p, &w)) > 0) {
                p->n_comp += q->n_comp + 1;
                p->seedcov = p->seedcov > q->seedcov ? p->seedcov : q->seedcov;
                p->sub = p->sub > q->sub ? p->sub : q->sub;
                p->csub = p->csub > q->csub ? p->csub : q->csub;
                p->qb = q->qb; p->rb = q->rb;
                p->truesc = p->score = score;
                p->w = w;
                q->qb = q->qe;
            }
        }
    }
    for (i = 0, m = 0; i < n; ++i)
        if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
    n = m;
    ks_introsort((size_t)n, a, RegSLt());
    for (i = 1; i < n; ++i)
        if (a[i].score == a[i - 1].score && a[i].rb == a[i - 1].rb && a[i].qb == a[i - 1].qb)
            a[i].qe = a[i].qb;
    for (i = 1, m = 1; i < n; ++i)
        if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
    return m;
}

__global__ void k_post1(DevIndex ix, MemOpt opt, TileView tv)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= tv.n_reads) return;
    PostScratch S = post_scratch_for(tv, r);
    const uint8_t* query = tv.seq + tv.seq_off[r];
    AlnReg* a = tv.regs + tv.seed_off[r];
    int n = sort_dedup_patch(ix, opt, S, query, tv.n_regs[r], a);
    for (int i = 0; i < n; ++i)
        if (a[i].rid >= 0 && ix.ann_is_alt[a[i].rid]) a[i].is_alt = 1;
    tv.n_regs[r] = n;
    if (S.err) atomicOr(tv.err, S.err);
}

// ------------------------------------------------------------------ single-end finalisation
__global__ void k_final_se(DevIndex ix, MemOpt opt, TileView tv)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= tv.n_reads) return;
    PostScratch S = post_scratch_for(tv, r);
    const uint8_t* query = tv.seq + tv.seq_off[r];
    int l_query = (int)(tv.seq_off[r + 1] - tv.seq_off[r] - 1);
    AlnReg* a = tv.regs + tv.seed_off[r];
    int n = tv.n_regs[r];
    int32_t* zbuf = (int32_t*)(tv.srt + tv.seed_off[r]);      // >= 2 ints per region
    OutBuf ob; ob.p = tv.out + (size_t)r * tv.out_cap; ob.cap = tv.out_cap; ob.len = 0; ob.ovf = false;

    mark_primary_se(opt, n, a, tv.read_id0 + r, zbuf);
    if (opt.flag & MEM_F_PRIMARY5) reorder_primary5(opt.T, n, a);
    reg2sam(ix, opt, S, ob, l_query, query, n, a, zbuf, 0, (const MateInfo*)0);

    tv.out_len[r] = ob.ovf ? 0 : ob.len;
    if (ob.ovf) atomicOr(tv.err, ERR_OUT_CAP);
    if (S.err) atomicOr(tv.err, S.err);
}

// gather the per-read staging slots into one contiguous result buffer
__global__ void k_pack(TileView tv, uint8_t* dst)
{
    int r = blockIdx.x * (blockDim.x >> 3) + (threadIdx.x >> 3);   // 8 lanes per read
    int sub = threadIdx.x & 7;
    if (r >= tv.n_reads) return;
    const uint32_t* src = (const uint32_t*)(tv.out + (size_t)r * tv.out_cap);
    uint32_t* d = (uint32_t*)(dst + tv.out_off[r]);
    int nw = tv.out_len[r] >> 2;
    for (int i = sub; i < nw; i += 8) d[i] = src[i];
}

void launch_post1(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv)
{
    if (tv.n_reads <= 0) return;
    hipLaunchKernelGGL(k_post1, dim3((tv.n_reads + 63) / 64), dim3(64), 0, st, ix, opt, tv);
}
void launch_final_se(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv)
{
    if (tv.n_reads <= 0) return;
    hipLaunchKernelGGL(k_final_se, dim3((tv.n_reads + 63) / 64), dim3(64), 0, st, ix, opt, tv);
}
void launch_pack(hipStream_t st, const TileView& tv, uint8_t* dst)
{
    if (tv.n_reads <= 0) return;
    hipLaunchKernelGGL(k_pack, dim3((tv.n_reads + 31) / 32), dim3(256), 0, st, tv, dst);
}
