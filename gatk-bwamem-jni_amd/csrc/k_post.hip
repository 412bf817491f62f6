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

// WAVE_PER_READ = false: one lane per read.  true (tiles of long reads): one wavefront per read -- sixty-four times the waves
// in flight for this latency-bound stage -- with lane 0 doing the updates and the banded global alignments of region
// patching spread across the lanes (sort_dedup_patch_wave / global_score_wave; rings in dynamic LDS).
template <bool WAVE_PER_READ>
__global__ void __launch_bounds__(64, WAVE_PER_READ ? 3 : 6) k_post1(DevIndex ix, MemOpt opt, TileView tv, int ring)
{
    HIP_DYNAMIC_SHARED(int32_t, smem)
    const int r = WAVE_PER_READ ? (int)blockIdx.x : (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (r >= tv.n_reads) return;
    const bool writer = !WAVE_PER_READ || threadIdx.x == 0;
    WaveDp wd; wd.eh_h = smem; wd.eh_e = smem + ring; wd.tmpM = smem + 2 * ring; wd.rm = ring - 1; wd.lane = (int)threadIdx.x; wd.no_pk = (tv.debug & 0x10000) != 0;
    PostScratch S = post_scratch_for(tv, r);
    const uint8_t* query = tv.seq + tv.seq_off[r];
    AlnReg* a = tv.regs + tv.seed_off[r];
    {   // regions must be sane before any loop is sized from them: fail the call loudly instead of spinning
        const int l_query = (int)(tv.seq_off[r + 1] - tv.seq_off[r] - 1);
        const int n0 = tv.n_regs[r];
        bool bad = n0 < 0 || n0 > tv.seed_off[r + 1] - tv.seed_off[r];
        for (int i = 0; !bad && i < n0; ++i)
            bad = a[i].qb < 0 || a[i].qe < a[i].qb || a[i].qe > l_query || a[i].rb < 0 || a[i].re < a[i].rb || a[i].re > ix.l_pac << 1
               || a[i].re - a[i].rb > 4 * (int64_t)l_query + 1024 || a[i].rid < 0 || a[i].rid >= ix.n_seqs;
        if (bad) {
            if (writer && !(atomicOr(tv.err, ERR_BAD_REG) & ERR_BAD_REG)) { tv.err[1] = r; tv.err[2] = n0; if (n0 > 0) { tv.err[3] = a[0].qb; tv.err[4] = a[0].qe; tv.err[5] = (int)a[0].rb; tv.err[6] = (int)a[0].re; tv.err[7] = a[0].score; } }
            if (WAVE_PER_READ) __syncthreads();
            if (writer) tv.n_regs[r] = 0;
            return;
        }
    }
    if (tv.debug & 0xff) printf("[k] post1 read %d n=%d\n", r, tv.n_regs[r]);
    const int n_in = tv.n_regs[r];
    if (WAVE_PER_READ) __syncthreads();                              // every lane has read the count before lane 0 replaces it
    SortKey* keys = (tv.debug & 0x400) ? nullptr : sort_keys_for(tv, r);          // BWAMEM_HIP_DEBUGK=1024: sort the regions themselves (tests)
    int n = WAVE_PER_READ ? sort_dedup_patch_wave(ix, opt, S, query, n_in, a, wd, keys) : sort_dedup_patch(ix, opt, S, query, n_in, a, tv.debug & 0xff, keys);
    if (tv.debug & 0xff) printf("[k] post1 read %d done n=%d\n", r, n);
    if (writer)
        for (int i = 0; i < n; ++i)
            if (a[i].rid >= 0 && ix.ann_is_alt[a[i].rid]) a[i].is_alt = 1;
    if (writer) tv.n_regs[r] = n;
    if (S.err && writer) atomicOr(tv.err, S.err);
}

// ------------------------------------------------------------------ single-end finalisation
// step 1: primary marking, and a job for every region whose record (or XA entry) needs a banded global alignment
__global__ void __launch_bounds__(64, 6) k_final_prep(DevIndex ix, MemOpt opt, TileView tv)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= tv.n_reads) return;
    AlnReg* a = tv.regs + tv.seed_off[r];
    const int n = tv.n_regs[r];
    int32_t* zbuf = (int32_t*)(tv.srt + tv.seed_off[r]);      // >= 2 ints per region
    mark_primary_se(opt, n, a, tv.read_id0 + r, zbuf, (tv.debug & 0x400) ? nullptr : sort_keys_for(tv, r));
    if (opt.flag & MEM_F_PRIMARY5) reorder_primary5(opt.T, n, a);
    int32_t *cnt = 0, *has_alt = 0;
    if (!(opt.flag & MEM_F_ALL) && n > 0) {
        cnt = zbuf; has_alt = zbuf + n;
        if (xa_prepare(opt, n, a, cnt, has_alt) == 0) cnt = has_alt = 0;
    }
    DpJob* jobs = (DpJob*)tv.jobs;
    for (int k = 0; k < n; ++k) {
        AlnReg* p = &a[k];
        p->pad_ = 0;
        // will mem_reg2sam emit a record for it?
        bool rec = !(p->score < opt.T) && !(p->secondary >= 0 && (p->is_alt || !(opt.flag & MEM_F_ALL)))
                && !(p->secondary >= 0 && p->secondary < INT_MAX_ && (float)p->score < (float)a[p->secondary].score * opt.drop_ratio);
        // will mem_gen_alt list it in an XA tag?
        bool xa = false;
        if (cnt) {
            int pr = get_pri_idx(opt.XA_drop_ratio, a, k);
            xa = pr >= 0 && !(cnt[pr] > opt.max_XA_hits_alt || (!has_alt[pr] && cnt[pr] > opt.max_XA_hits));
        }
        if ((rec || xa) && region_needs_dp(opt, *p)) {
            // (one slot per reservation: every slot below job_cap that was handed out gets written, so an overflow leaves no
            // unwritten slot behind -- unlike the several-slot reservations of k_pe_rescue_plan / k_rescore_plan)
            int job = atomicAdd(tv.job_cnt, 1);
            if (job < tv.job_cap) { DpJob jb; jb.read = r; jb.reg = k; jobs[job] = jb; p->pad_ = job + 1; }
            else atomicOr(tv.err, ERR_JOB_CAP);
        }
    }
}

// step 3 (after k_gcigar): mem_reg2sam record selection and the reference's fmt_BAMish record writer
// (64, 6): a tile's 6 144 waves are six per SIMD; with the default register budget only four would be resident, and this stage
// waits on dependent loads most of the time
__global__ void __launch_bounds__(64, 6) k_final_se(DevIndex ix, MemOpt opt, TileView tv, JobView jv)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= tv.n_reads) return;
    PostScratch S = post_scratch_for(tv, r);
    const uint8_t* query = tv.seq + tv.seq_off[r];
    int l_query = (int)(tv.seq_off[r + 1] - tv.seq_off[r] - 1);
    AlnReg* a = tv.regs + tv.seed_off[r];
    int n = tv.n_regs[r];
    int32_t* zbuf = (int32_t*)(tv.srt + tv.seed_off[r]);
    OutBuf ob; ob.p = tv.out + (size_t)r * tv.out_cap; ob.cap = tv.out_cap; ob.len = 0; ob.ovf = false;

    reg2sam(ix, opt, S, ob, l_query, query, n, a, zbuf, 0, (const MateInfo*)0, &jv);

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
    if (tv.max_len > 1000) {                                         // long reads: a wavefront per read
        int ring = 64;
        long long need = 8ll * (opt.w > 0 ? opt.w : 0) + 16;
        if (need > (long long)tv.max_len + 4) need = (long long)tv.max_len + 4;
        if (need > 4096) need = 4096;
        while (ring < need) ring <<= 1;
        hipLaunchKernelGGL(k_post1<true>, dim3(tv.n_reads), dim3(64), 3 * (size_t)ring * sizeof(int32_t), st, ix, opt, tv, ring);
        return;
    }
    hipLaunchKernelGGL(k_post1<false>, dim3((tv.n_reads + 63) / 64), dim3(64), 0, st, ix, opt, tv, 64);
}
void launch_final_prep(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv)
{
    if (tv.n_reads <= 0) return;
    hipLaunchKernelGGL(k_final_prep, dim3((tv.n_reads + 63) / 64), dim3(64), 0, st, ix, opt, tv);
}
void launch_final_se(hipStream_t st, const DevIndex& ix, const MemOpt& opt, const TileView& tv, const void* job_out, const uint32_t* job_cig, int cig_cap)
{
    if (tv.n_reads <= 0) return;
    JobView jv; jv.out = (const DpOut*)job_out; jv.cig = job_cig; jv.cig_cap = cig_cap;
    hipLaunchKernelGGL(k_final_se, dim3((tv.n_reads + 63) / 64), dim3(64), 0, st, ix, opt, tv, jv);
}
void launch_pack(hipStream_t st, const TileView& tv, uint8_t* dst)
{
    if (tv.n_reads <= 0) return;
    hipLaunchKernelGGL(k_pack, dim3((tv.n_reads + 31) / 32), dim3(256), 0, st, tv, dst);
}
