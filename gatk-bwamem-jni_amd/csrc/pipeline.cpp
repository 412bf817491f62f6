// pipeline.cpp -- host side of the library: index residency in HBM, request upload, the
// tile loop that drives the kernels, result download, and the jnibwa_* C ABI.
//
// Drop-in boundary: reference src/main/c/jnibwa.c:126-235 (every exported function cites its
// counterpart in include/bwamem_hip.h).  The device pipeline replaces the single upstream call
// mem_process_seqs at jnibwa.c:214.  There is no host fallback: every stage of the hot path
// runs as a HIP kernel and any device error makes the call return NULL.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <unistd.h>
#include <fcntl.h>
#include <errno.h>
#include <time.h>
#include <sys/stat.h>
#include <sys/mman.h>
#include <dlfcn.h>
#include <mutex>
#include <condition_variable>
#include <thread>
#include <atomic>
#include <functional>
#include <string>
#include <vector>
#include <deque>
#include <stdexcept>
#include <algorithm>
#include <cmath>
#include "bwamem_types.h"
#include "kernels.h"
#include "index_io.h"
#include "../../include/bwamem_hip.h"

#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    fprintf(stderr, "[bwamem_hip] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return false; } } while (0)

// ------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    bool ensure(size_t n) {
        if (n <= bytes) return true;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        n += n / 8 + 256;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) { fprintf(stderr, "[bwamem_hip] hipMalloc(%zu) failed: %s\n", n, hipGetErrorString(e)); p = nullptr; return false; }
        bytes = n;
        return true;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <typename T> T* as() const { return (T*)p; }
};

struct Stats {
    std::mutex mu;
    bool enabled = false;
    bwamem_stats_t s;
    Stats() { memset(&s, 0, sizeof s); }
};
static Stats g_stats;
static int g_device = 0;
static bool g_device_explicit = false;     // bwamem_hip_set_device was called: indexes opened afterwards live on that device only

enum KernelId { K_ENCODE, K_SEED, K_SA, K_CHAIN, K_EXTEND, K_POST, K_FINAL, K_PACK, K_OTHER, K_N };

struct Timed { KernelId id; hipEvent_t a, b; };

// DP rows in global memory (tv.dp_rows).  The wave forms of k_extend and k_gcigar keep their rows in LDS as rings around
// the band, so this is the exception: k_gcigar jobs whose band is wider than its rings (regions with a long net indel,
// only possible when the reads are longer than the rings), k_extend for reads that do not fit LDS next to the rings, and
// BWAMEM_HIP_DP_ROWS=hbm, which sends everything there (tests).
static void dp_rows_policy(int lds, const MemOpt& opt, int L, bool& need_rows, bool& ext_hbm, bool& gcigar_hbm_only)
{
    if (lds <= 0) lds = 64 << 10;
    const char* e = getenv("BWAMEM_HIP_DP_ROWS");
    const bool forced = e && !strcmp(e, "hbm");
    ext_hbm = forced || extend_lds_bytes(opt, L) + 64 > (size_t)lds;
    gcigar_hbm_only = forced || (size_t)L + 4 + 3 * 1024 * 4 + 6144 + 64 > (size_t)lds;    // (the read itself next to the smallest useful rings)
    need_rows = forced || ext_hbm || gcigar_hbm_only || (long long)L + 4 > 8ll * (opt.w > 0 ? opt.w : 0) + 16 || getenv("BWAMEM_HIP_GCIGAR_RING") != nullptr;
}

struct Workspace {
    int T = 0, L = 0, intv_cap = 0, smem_cap = 0, out_cap = 0;
    int64_t seed_cap = 0, post_per_read = 0;
    DevBuf intv, n_intv, smem, l_rep, n_seeds, seed_off, intv_seed_off;
    DevBuf seeds, seed_rid, cseeds, chains, chain_store, n_chains, bt_nodes, srt, regs, n_regs;
    DevBuf out, out_len, out_off, post, err, cnt;
    DevBuf dp_rows; int dp_rows_blocks = 0;          // DP rows of k_extend / k_gcigar in global memory (dp_rows_policy)
    bool ext_hbm = false, gcigar_hbm_only = false;
    DevBuf jobs, job_out, job_cig, job_cnt, zpool, zslabs;   // global-alignment jobs; zslabs: traceback slabs of k_gcigar's resident grid (long reads)
    size_t zslab_bytes = 0;
    int job_cap = 0, job_cig_cap = 0; size_t zpool_cap = 0;
    int out_cap_hint = 512;                           // bytes per read of the output staging slots (grown on overflow, kept across tiles)
    int dev_lds = 0;                                  // LDS per workgroup of the device this workspace lives on
    DevBuf scan_tmp, packed, order;                        // block sums of the multi-block scan; the tile's packed records on their way to the host
    DevBuf pe_dir, pe_is, pe_caps, pe_reg_off2, pe_regs2, pe_ints2, pe_vpool, pe_scratch, pe_states, pe_rescue[3];   // paired-end stages
    hipStream_t stream = nullptr;
    std::vector<Timed> timed;

    // with_seed = false: the tile takes its interval lists from a SeedStore (single-end path) and needs no seeding arrays
    bool ensure_reads(const MemOpt& opt, int T_, int L_, int intv_cap_, int out_cap_, int64_t post_per_read_, bool with_seed = true) {
        T = T_; L = L_; intv_cap = intv_cap_; smem_cap = L_ + 2; out_cap = out_cap_; post_per_read = post_per_read_;
        size_t t = (size_t)T;
        if (with_seed && !(intv.ensure(t * intv_cap * sizeof(Intv)) && n_intv.ensure(t * 4) && smem.ensure(((t + 63) / 64 * 64) * 2 * smem_cap * 16)
            && l_rep.ensure(t * 4) && n_seeds.ensure(t * 4) && intv_seed_off.ensure(t * intv_cap * 4))) return false;
        dp_rows_blocks = 0; ext_hbm = gcigar_hbm_only = false;
        bool need_rows = false;
        if (!with_seed) dp_rows_policy(dev_lds, opt, L, need_rows, ext_hbm, gcigar_hbm_only);
        if (need_rows) {                                                            // a bounded grid of workgroups, each with its own three rows (about 2 GB in all)
            const size_t per_block = 3 * ((size_t)L + 2) * 4;
            dp_rows_blocks = (int)std::min<size_t>(4096, std::max<size_t>(256, ((size_t)2 << 30) / per_block));
            if (!dp_rows.ensure((size_t)dp_rows_blocks * per_block)) return false;
        }
        return seed_off.ensure((t + 1) * 8) && scan_tmp.ensure(scan_tmp_bytes((int64_t)t + 1)) && order.ensure(t * 4 + 64 * 4)
            && n_chains.ensure(t * 4) && n_regs.ensure(t * 4) && out.ensure(t * out_cap) && out_len.ensure(t * 4)
            && out_off.ensure((t + 1) * 8) && post.ensure(t * (size_t)post_per_read) && err.ensure(64) && cnt.ensure(sizeof(DevCounters));
    }
    bool ensure_jobs(int n_jobs, int cig_cap_, size_t zpool_bytes) {
        job_cap = std::max(job_cap, n_jobs); job_cig_cap = cig_cap_; zpool_cap = std::max(zpool_cap, zpool_bytes);
        return jobs.ensure((size_t)job_cap * 8) && job_out.ensure((size_t)job_cap * 8) && job_cig.ensure((size_t)job_cap * job_cig_cap * 4)
            && job_cnt.ensure(64) && zpool.ensure(zpool_cap);
    }
    bool ensure_seeds(int64_t n) {
        seed_cap = n;
        size_t s = (size_t)n + 16;
        return seeds.ensure(s * sizeof(Seed)) && seed_rid.ensure(s * 4) && cseeds.ensure(s * sizeof(Seed)) && chains.ensure(s * sizeof(Chain))
            && chain_store.ensure(s * sizeof(Chain)) && bt_nodes.ensure((s / 4 + 3 * (size_t)T + 16) * BT_NODE_INTS * 4) && srt.ensure(s * 8)
            && regs.ensure(s * sizeof(AlnReg));
    }
    void release() {
        DevBuf* all[] = { &intv, &n_intv, &smem, &l_rep, &n_seeds, &seed_off, &intv_seed_off, &seeds, &seed_rid, &cseeds, &chains,
                          &chain_store, &n_chains, &bt_nodes, &srt, &regs, &n_regs, &out, &out_len, &out_off, &post, &err, &cnt, &dp_rows,
                          &jobs, &job_out, &job_cig, &job_cnt, &zpool, &zslabs, &pe_dir, &pe_is, &pe_caps, &pe_reg_off2, &pe_regs2, &pe_ints2, &pe_vpool, &pe_scratch, &pe_states,
                          &pe_rescue[0], &pe_rescue[1], &pe_rescue[2], &scan_tmp, &packed, &order };
        for (DevBuf* b : all) b->release();
        if (stream) { (void)hipStreamDestroy(stream); stream = nullptr; }
    }
    TileView view() const {
        TileView tv; memset(&tv, 0, sizeof tv);
        tv.intv_cap = intv_cap; tv.intv = intv.as<Intv>(); tv.n_intv = n_intv.as<int32_t>();
        tv.smem_scratch = smem.as<Intv>(); tv.smem_cap = smem_cap; tv.l_rep = l_rep.as<int32_t>();
        tv.n_seeds = n_seeds.as<int32_t>(); tv.seed_off = seed_off.as<int64_t>(); tv.intv_seed_off = intv_seed_off.as<int32_t>();
        tv.seeds = seeds.as<Seed>(); tv.seed_rid = seed_rid.as<int32_t>(); tv.cseeds = cseeds.as<Seed>();
        tv.chains = chains.as<Chain>(); tv.n_chains = n_chains.as<int32_t>(); tv.bt_nodes = bt_nodes.as<int32_t>();
        tv.srt = srt.as<uint64_t>(); tv.regs = regs.as<AlnReg>(); tv.n_regs = n_regs.as<int32_t>();
        tv.out_cap = out_cap; tv.out = out.as<uint8_t>(); tv.out_len = out_len.as<int32_t>(); tv.out_off = out_off.as<int64_t>();
        tv.post_scratch = post.as<uint8_t>(); tv.post_scratch_per_read = post_per_read;
        tv.err = err.as<int32_t>(); tv.cnt = cnt.as<DevCounters>();
        tv.dp_rows = dp_rows_blocks ? dp_rows.as<int32_t>() : nullptr; tv.dp_rows_blocks = dp_rows_blocks;
        tv.ext_hbm = ext_hbm; tv.gcigar_hbm_only = gcigar_hbm_only;
        tv.order = nullptr;
        tv.job_cnt = job_cnt.as<int32_t>(); tv.jobs = jobs.p; tv.job_cap = job_cap;
        tv.smem_groups = (T + 63) / 64;
        { const char* e = getenv("BWAMEM_HIP_DEBUGK"); tv.debug = e ? atoi(e) : 0; }
        return tv;
    }
};

// Interval lists of one seeding chunk (single-end path): k_seed runs over chunks of a few million reads -- several
// tiles -- because its running time is bounded below by the slowest read of a launch; the tiles of the chunk then
// point their TileView at slices of these arrays.
#define SEED_STORES_MAX 4
struct SeedStore {
    DevBuf intv, n_intv, intv_seed_off, n_seeds, l_rep;
    int cap = 0;                      // intervals per read
    bool ensure(size_t reads, int cap_) {
        cap = cap_;
        return intv.ensure(reads * cap * sizeof(Intv)) && n_intv.ensure(reads * 4) && intv_seed_off.ensure(reads * cap * 4)
            && n_seeds.ensure(reads * 4) && l_rep.ensure(reads * 4);
    }
    void release() { intv.release(); n_intv.release(); intv_seed_off.release(); n_seeds.release(); l_rep.release(); }
};

// A stretch of a request that jnibwa_createAlignments streams to the device while the first stretches are already being
// aligned: the bases (ASCII as sent, encoded in place), the read offsets found on the device and their host copy (pinned).
// The buffers belong to the index and are reused by the next call (calls on one index are serialised).
struct ReqBuf {
    DevBuf seq, off, tmp;
    int64_t* h_off = nullptr; size_t h_cap = 0;       // hipHostMalloc
    bool ensure_host(size_t n) {
        if (n <= h_cap) return true;
        if (h_off) (void)hipHostFree(h_off);
        h_off = nullptr; h_cap = 0;
        n += n / 8 + 64;
        if (hipHostMalloc((void**)&h_off, n * 8, hipHostMallocDefault) != hipSuccess) { h_off = nullptr; return false; }
        h_cap = n;
        return true;
    }
    void release() { seq.release(); off.release(); tmp.release(); if (h_off) (void)hipHostFree(h_off); h_off = nullptr; h_cap = 0; }
};

// What the tiles of earlier calls needed, per read, shared by all workspaces of an index: a tile is re-run when one of its
// buffers turns out too small (deterministic kernels, so a retry is exact), and whichever worker picks up the next tile
// should not have to learn the same sizes again.
struct CapHints {
    std::mutex mu;
    double jobs_per_read = 0, zpool_mult = 1;       // zpool_mult: how many times the default traceback pool a tile ended up needing
    double seeds_per_read = 0;                      // seed occurrences per read of the tiles seen so far (running mean): how repetitive the data is
    int out_cap = 512;
    void saw_seeds(int T, int64_t n_occ) { std::lock_guard<std::mutex> lk(mu); if (T >= 1024) seeds_per_read = seeds_per_read == 0 ? (double)n_occ / T : 0.75 * seeds_per_read + 0.25 * (double)n_occ / T; }
    // Reads in repeats carry a hundred times the work of unique ones, one read per lane or wave, and a kernel lasts as long as
    // its heaviest read: on repeat-rich data twice the reads per tile amortise those tails over twice the work (human-like
    // genome: +30 %), on easy data larger tiles only coarsen the overlap between tiles (-3 %).
    int tile_scale() { std::lock_guard<std::mutex> lk(mu); return seeds_per_read > 32 ? 2 : 1; }
    void learn(int T, int n_jobs, double zpool_mult_, int out_cap_) {
        std::lock_guard<std::mutex> lk(mu);
        if (T >= 64) jobs_per_read = std::max(jobs_per_read, 1.25 * n_jobs / T);
        zpool_mult = std::min(64.0, std::max(zpool_mult, zpool_mult_));
        out_cap = std::max(out_cap, out_cap_);
    }
    void get(int T, int& jobs, double& zpool_mult_, int& out_cap_) {
        std::lock_guard<std::mutex> lk(mu);
        jobs = (int)std::min(1.0e9, jobs_per_read * T) + 1; zpool_mult_ = zpool_mult; out_cap_ = out_cap;
    }
    // paired-end phase 2: mate-rescue alignments and global-alignment jobs per read of the tiles seen so far (pairs in repeats ask
    // for hundreds of rescue alignments; a tile sized for the easy case runs its stage twice)
    double pe_rescue_per_read = 0, pe_jobs_per_read = 0;
    int pe_cap_u = 256;                             // candidate pairs per pair in mem_pair's list (a pair inside a tandem array has thousands)
    void learn_pe(int T, int n_rescue, int n_jobs, int cap_u) {
        std::lock_guard<std::mutex> lk(mu);
        if (T >= 64) { pe_rescue_per_read = std::max(pe_rescue_per_read, 1.25 * n_rescue / T); pe_jobs_per_read = std::max(pe_jobs_per_read, 1.25 * n_jobs / T); }
        pe_cap_u = std::min(2048, std::max(pe_cap_u, cap_u));      // (16 bytes each for every pair of a tile: larger lists stay a per-tile retry)
    }
    void get_pe(int T, int& rescue, int& jobs, int& cap_u) {
        std::lock_guard<std::mutex> lk(mu);
        rescue = (int)std::min(1.0e9, pe_rescue_per_read * T) + 1; jobs = (int)std::min(1.0e9, pe_jobs_per_read * T) + 1; cap_u = pe_cap_u;
    }
};

struct bwaidx_s {
    uint8_t* mem = nullptr; size_t l_mem = 0; bool mmapped = false;
    HostIndex h;
    int device = 0;
    DevIndex d;
    DevBuf d_occ, d_sa_lo, d_sa_hi, d_pac, d_ann_off, d_ann_len, d_ann_alt, d_name_off, d_names, d_log, d_ptab;
    PairTab pair_tab;               // insert-size score terms of the running paired-end call (build_pair_tab)
    std::mutex mu;                  // one call at a time per index/device
    Workspace ws;
    std::vector<Workspace*> extra_ws;   // further tiles in flight (one stream + host thread each)
    Workspace seed_ws;                  // stream, flags and spill area of the seeding stage (single-end path)
    SeedStore seed_store[SEED_STORES_MAX];   // a ring: chunk c + n is seeded while the tiles of chunk c run (n = seed_stores() - 1)
    CapHints hints;
    std::vector<ReqBuf*> req_bufs;      // request stretches of a streamed call (jnibwa_createAlignments)
    hipStream_t up_stream = nullptr;    // their uploads
    // One handle, several devices (jnibwa_openIndex, BWAMEM_HIP_DEVICES): the handle is the replica on the first device and
    // owns the others, each a bwaidx_s of its own (index, workspaces, hints) over the same mapped image.
    std::vector<bwaidx_s*> peers;
    std::atomic<uint32_t> next_replica{0};   // small calls take the replicas in turn
    std::mutex split_mu;                     // one call at a time is cut across all replicas (its shards wait for each other while holding their replicas)
};

struct TileOut { uint8_t* d = nullptr; size_t bytes = 0; bool owned = false; };

// Where the packed records of the tiles go.  Tiles finish out of order on their workers but the response is the
// concatenation in read order, so a tile learns its offset once every earlier tile has announced its size.
//   to_host (jnibwa_createAlignments): straight into the malloc'ed block Java will free, tile by tile as they finish
//     (grown by realloc, only while no copy is in flight);
//   otherwise (bwamem_hip_batch_*): into one device slab sized from the previous call on this batch; a tile that does
//     not fit (or the first call, when no size is known) gets an allocation of its own.
struct OutSink {
    bool to_host = false;
    std::mutex mu; std::condition_variable cv;
    std::deque<int64_t> bytes, start;               // per tile: size (-1 = not known yet), offset in the response
    size_t n_placed = 0; int64_t placed_end = 0;    // tiles [0, n_placed) have their offsets
    uint64_t reads_total = 0, reads_known = 0;
    uint8_t* h_buf = nullptr; size_t h_cap = 0; int inflight = 0;
    DevBuf pool; size_t last_total = 0;
    bool aborted = false;
    void begin_call(uint64_t n_reads) {
        std::lock_guard<std::mutex> lk(mu);
        bytes.clear(); start.clear(); n_placed = 0; placed_end = 0; reads_total = n_reads; reads_known = 0; inflight = 0; aborted = false;
        if (h_buf) { free(h_buf); h_buf = nullptr; } h_cap = 0;
    }
    void add_tiles(size_t n) { std::lock_guard<std::mutex> lk(mu); bytes.resize(bytes.size() + n, -1); start.resize(start.size() + n, 0); }
    void abort() { { std::lock_guard<std::mutex> lk(mu); aborted = true; } cv.notify_all(); }
    // tile i has nbytes of records: its place in the response.  Host mode: the destination, to be released with copy_done().
    // Device mode: a slice of the slab, or null (allocate).  false: the call failed elsewhere, or out of memory.
    bool place(size_t i, int64_t nbytes, int n_reads, uint8_t** dst) {
        std::unique_lock<std::mutex> lk(mu);
        bytes[i] = nbytes; reads_known += (uint64_t)n_reads;
        cv.notify_all();
        cv.wait(lk, [&] {
            while (n_placed < bytes.size() && bytes[n_placed] >= 0) { start[n_placed] = placed_end; placed_end += bytes[n_placed]; ++n_placed; }
            return aborted || n_placed > i;
        });
        if (aborted) return false;
        const size_t need = (size_t)(start[i] + nbytes);
        if (!to_host) { *dst = pool.p && need <= pool.bytes ? pool.as<uint8_t>() + start[i] : nullptr; return true; }
        if (need > h_cap) {
            cv.wait(lk, [&] { return aborted || inflight == 0; });
            if (aborted) return false;
            if (need > h_cap) {
                size_t est = reads_known ? (size_t)((double)placed_end / (double)reads_known * (double)reads_total * 1.25) : 0;
                const char* es = getenv("BWAMEM_HIP_OUT_SLACK");                  // (tests: 0 makes the block grow tile by tile)
                size_t cap = std::max(std::max(need, est), h_cap + h_cap / 2) + (es ? (size_t)atoll(es) : (size_t)1 << 20);
                uint8_t* nb = (uint8_t*)realloc(h_buf, cap);
                if (!nb) { aborted = true; cv.notify_all(); return false; }
                h_buf = nb; h_cap = cap;
#ifdef MADV_HUGEPAGE
                {   // half a gigabyte of first-touch page faults sits on the workers' copy path: ask for huge pages (a hint; ignored where unavailable)
                    const uintptr_t lo = ((uintptr_t)nb + 4095) & ~(uintptr_t)4095, hi = ((uintptr_t)nb + cap) & ~(uintptr_t)4095;
                    if (hi > lo + ((size_t)2 << 20)) (void)madvise((void*)lo, hi - lo, MADV_HUGEPAGE);
                }
#endif
            }
        }
        ++inflight;
        *dst = h_buf + start[i];
        return true;
    }
    void copy_done() { { std::lock_guard<std::mutex> lk(mu); --inflight; } cv.notify_all(); }
    // the finished response (host mode): shrunk to its size; the caller owns it
    void* take(size_t total) {
        std::lock_guard<std::mutex> lk(mu);
        void* r = h_buf ? realloc(h_buf, total ? total : 1) : malloc(total ? total : 1);
        if (!r) r = h_buf;
        h_buf = nullptr; h_cap = 0;
        return r;
    }
    ~OutSink() { if (h_buf) free(h_buf); }
};

struct bwamem_batch_s {
    bwaidx_s* idx = nullptr;
    uint32_t n_reads = 0;
    size_t n_bytes = 0;
    DevBuf d_raw, d_seq, d_off;     // d_raw: the request as uploaded (ASCII); d_seq: working copy, encoded per call
    std::vector<int64_t> h_off;
    const char* h_payload = nullptr; // streamed form (jnibwa_createAlignments): the caller's strings, uploaded stretch by stretch during the call
    std::deque<TileOut> tiles;
    OutSink sink;
    size_t result_bytes = 0;
    struct PeCall* pe = nullptr;    // paired-end call split in two steps (bwamem_hip_batch_pe_begin / _finish): phase-1 products kept in HBM
};

static const int LOG_TAB_N = 1 << 20;

// ------------------------------------------------------------------------------------------ index
static bool upload_index(bwaidx_s* ix)
{
    HIP_OK(hipSetDevice(ix->device));
    const HostIndex& h = ix->h;
    const int n = (int)h.contigs.size();
    {   // upload the image's occ/bwt array to a scratch buffer and re-block it into the device layout
        DevBuf tmp;
        size_t bwt_bytes = ((size_t)h.bwt_size * 4 + 255) & ~(size_t)63;
        const uint64_t n_blocks = (h.seq_len + 63) / 64 + 1;
        if (!tmp.ensure(bwt_bytes) || !ix->d_occ.ensure((size_t)n_blocks * 32 + 64)) { tmp.release(); return false; }
        HIP_OK(hipMemset(tmp.p, 0, tmp.bytes));
        HIP_OK(hipMemcpy(tmp.p, h.bwt, (size_t)h.bwt_size * 4, hipMemcpyHostToDevice));
        launch_build_occ64(0, tmp.as<uint32_t>(), n_blocks, ix->d_occ.as<uint4>());
        HIP_OK(hipDeviceSynchronize());
        tmp.release();
    }
    if (!ix->d_pac.ensure((size_t)(h.l_pac / 4 + 1) + 16)) return false;
    HIP_OK(hipMemcpy(ix->d_pac.p, h.pac, (size_t)(h.l_pac / 4 + 1), hipMemcpyHostToDevice));
    std::vector<int64_t> off(n); std::vector<int32_t> len(n), alt(n), noff(n + 1);
    std::string names;
    for (int i = 0; i < n; ++i) {
        off[i] = h.contigs[i].offset; len[i] = h.contigs[i].len; alt[i] = h.contigs[i].is_alt;
        noff[i] = (int32_t)names.size();
        names += h.contigs[i].name; names.push_back('\0');
    }
    noff[n] = (int32_t)names.size();
    if (!ix->d_ann_off.ensure((size_t)n * 8 + 8) || !ix->d_ann_len.ensure((size_t)n * 4 + 4) || !ix->d_ann_alt.ensure((size_t)n * 4 + 4)
        || !ix->d_name_off.ensure((size_t)(n + 1) * 4) || !ix->d_names.ensure(names.size() + 1) || !ix->d_log.ensure((size_t)LOG_TAB_N * 8)) return false;
    HIP_OK(hipMemcpy(ix->d_ann_off.p, off.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(ix->d_ann_len.p, len.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(ix->d_ann_alt.p, alt.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(ix->d_name_off.p, noff.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(ix->d_names.p, names.data(), names.size(), hipMemcpyHostToDevice));
    {   // log() stays a host (glibc) function: decisions that depend on it read this table
        std::vector<double> lt(LOG_TAB_N);
        for (int i = 0; i < LOG_TAB_N; ++i) lt[i] = log((double)i);
        HIP_OK(hipMemcpy(ix->d_log.p, lt.data(), (size_t)LOG_TAB_N * 8, hipMemcpyHostToDevice));
    }
    DevIndex& d = ix->d;
    memset(&d, 0, sizeof d);
    d.occ = ix->d_occ.as<uint4>(); d.pac = ix->d_pac.as<uint8_t>();
    d.ann_offset = ix->d_ann_off.as<int64_t>(); d.ann_len = ix->d_ann_len.as<int32_t>(); d.ann_is_alt = ix->d_ann_alt.as<int32_t>();
    d.ann_name_off = ix->d_name_off.as<int32_t>(); d.names = ix->d_names.as<char>(); d.log_tab = ix->d_log.as<double>();
    d.primary = h.primary; for (int i = 0; i < 5; ++i) d.L2[i] = h.L2[i];
    d.seq_len = h.seq_len; d.l_pac = h.l_pac; d.n_seqs = n; d.log_tab_n = LOG_TAB_N;
    {   // launch geometry of THIS device (a process may open indexes on different GPUs)
        int v = 0;
        d.n_cu = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ix->device) == hipSuccess && v > 0 ? v : 256;
        d.lds_bytes = hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, ix->device) == hipSuccess && v > 0 ? v : 64 << 10;
    }
    {   // suffix array: the image samples every h.sa_intv-th rank; keep it as dense as HBM comfortably allows (5 bytes per
        // kept rank, at most a quarter of what is free now), so that a lookup is a short LF-walk or none at all
        int dense = 1;
        if (const char* e = getenv("BWAMEM_HIP_SA_INTV")) { dense = atoi(e); if (dense < 1 || (dense & (dense - 1))) dense = 1; }
        size_t free_b = 0, total_b = 0;
        HIP_OK(hipMemGetInfo(&free_b, &total_b));
        if (dense > h.sa_intv) dense = h.sa_intv;
        while (dense < h.sa_intv && ((size_t)(h.seq_len / dense) + 1) * 5 + (size_t)h.n_sa * 8 > free_b / 4) dense <<= 1;
        d.sa_intv = dense; d.sa_shift = 0;
        while ((1 << d.sa_shift) < dense) ++d.sa_shift;
        const size_t n_kept = (size_t)(h.seq_len >> d.sa_shift) + 1;
        DevBuf src, flag;
        if (!ix->d_sa_lo.ensure(n_kept * 4 + 64) || !ix->d_sa_hi.ensure(n_kept + 64) || !src.ensure((size_t)h.n_sa * 8) || !flag.ensure(64)) { src.release(); flag.release(); return false; }
        d.sa_lo = ix->d_sa_lo.as<uint32_t>(); d.sa_hi = ix->d_sa_hi.as<uint8_t>();
        int32_t bad = 1;
        if (hipMemcpy(src.p, h.sa, (size_t)h.n_sa * 8, hipMemcpyHostToDevice) == hipSuccess && hipMemset(flag.p, 0, 64) == hipSuccess) {
            launch_sa_densify(0, d, src.as<uint64_t>(), h.n_sa, h.sa_intv, ix->d_sa_lo.as<uint32_t>(), ix->d_sa_hi.as<uint8_t>(), flag.as<int32_t>());
            if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&bad, flag.p, 4, hipMemcpyDeviceToHost) != hipSuccess) bad = 1;
        }
        src.release(); flag.release();
        if (bad) { fprintf(stderr, "[bwamem_hip] suffix array densification failed\n"); return false; }
    }
    return true;
}

static void free_index(bwaidx_s* ix)
{
    (void)hipSetDevice(ix->device);
    DevBuf* all[] = { &ix->d_occ, &ix->d_sa_lo, &ix->d_sa_hi, &ix->d_pac, &ix->d_ann_off, &ix->d_ann_len, &ix->d_ann_alt, &ix->d_name_off, &ix->d_names, &ix->d_log, &ix->d_ptab };
    for (DevBuf* b : all) b->release();
    ix->ws.release();
    ix->seed_ws.release();
    for (SeedStore& st : ix->seed_store) st.release();
    for (Workspace* w : ix->extra_ws) { w->release(); delete w; }
    ix->extra_ws.clear();
    for (ReqBuf* r : ix->req_bufs) { r->release(); delete r; }
    ix->req_bufs.clear();
    if (ix->up_stream) { (void)hipStreamDestroy(ix->up_stream); ix->up_stream = nullptr; }
}

// ------------------------------------------------------------------------------------------ roctx ranges
// Named ranges around the stages of a call (request upload, seeding chunk, each tile, tile download, pestat reduction) for
// rocprofv3 --marker-trace.  The marker library is looked up at run time (rocprofiler-sdk's roctx, else roctracer's): no
// link dependency, and without a profiler's library on the path the ranges are no-ops.
struct Roctx {
    int (*push)(const char*) = nullptr; int (*pop)() = nullptr;
    Roctx() {
        if (const char* e = getenv("BWAMEM_HIP_ROCTX")) if (atoi(e) == 0) return;
        for (const char* lib : { "librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4" }) {
            void* h = dlopen(lib, RTLD_LAZY | RTLD_LOCAL);
            if (!h) continue;
            push = (int (*)(const char*))dlsym(h, "roctxRangePushA"); pop = (int (*)())dlsym(h, "roctxRangePop");
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
static Roctx& roctx() { static Roctx r; return r; }
struct RoctxRange { bool on; RoctxRange(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); } ~RoctxRange() { if (on) roctx().pop(); } };

// ------------------------------------------------------------------------------------------ timing
static void timed_begin(Workspace& ws, KernelId id)
{
    if (!g_stats.enabled) return;
    Timed t; t.id = id;
    if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) return;
    (void)hipEventRecord(t.a, ws.stream);
    ws.timed.push_back(t);
}
static void timed_end(Workspace& ws)
{
    if (!g_stats.enabled || ws.timed.empty()) return;
    (void)hipEventRecord(ws.timed.back().b, ws.stream);
}
static void timed_collect(Workspace& ws)
{
    if (ws.timed.empty()) return;
    std::lock_guard<std::mutex> lk(g_stats.mu);
    for (Timed& t : ws.timed) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            double* slot[K_N] = { &g_stats.s.ms_encode, &g_stats.s.ms_seed, &g_stats.s.ms_sa, &g_stats.s.ms_chain, &g_stats.s.ms_extend,
                                  &g_stats.s.ms_post, &g_stats.s.ms_final, &g_stats.s.ms_pack, &g_stats.s.ms_other };
            *slot[t.id] += ms;
            if (t.id == K_SEED) ++g_stats.s.n_launch_seed;
            if (t.id == K_SA) ++g_stats.s.n_launch_sa;
            if (t.id == K_EXTEND) ++g_stats.s.n_launch_extend;
        }
        (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b);
    }
    ws.timed.clear();
}
static int g_verbose = -1;
static bool verbose() { if (g_verbose < 0) { const char* e = getenv("BWAMEM_HIP_VERBOSE"); g_verbose = e && atoi(e) > 0; } return g_verbose > 0; }
// BWAMEM_HIP_VERBOSE=1: synchronise after every launch and log its wall time (debugging aid only)
static void verbose_sync(Workspace& ws, const char* what)
{
    static double t_last = 0;
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    double t0 = ts.tv_sec + ts.tv_nsec * 1e-9;
    hipError_t e = hipStreamSynchronize(ws.stream);
    clock_gettime(CLOCK_MONOTONIC, &ts);
    double t1 = ts.tv_sec + ts.tv_nsec * 1e-9;
    fprintf(stderr, "[bwamem_hip] %-28s %9.3f ms  (%s)\n", what, (t1 - t0) * 1e3, e == hipSuccess ? "ok" : hipGetErrorString(e));
    fflush(stderr);
    t_last = t1; (void)t_last;
}
// every launch is checked: a launch the device refuses (e.g. more LDS than a CU has) must fail the call, not leave stale buffers behind
#define TIMED(ws, id, call) do { timed_begin(ws, id); call; timed_end(ws); \
    { hipError_t le_ = hipGetLastError(); if (le_ != hipSuccess) { fprintf(stderr, "[bwamem_hip] launch failed: %s (%s)\n", hipGetErrorString(le_), #call); return false; } } \
    if (verbose()) verbose_sync(ws, #call); } while (0)

// ------------------------------------------------------------------------------------------ tile loop
// per-read scratch of the one-lane-per-read post stages: (h,e) row, CIGAR, MD and -- only where the scalar global
// alignment with traceback still runs there (paired-end) -- the traceback matrix
static int64_t post_bytes_per_read(int L, const MemOpt& opt, bool with_traceback = true)
{
    int64_t ncol = std::min<int64_t>(L, 2 * ((int64_t)opt.w << 2) + 1);
    int64_t tl = 3 * (int64_t)L + 64;
    int64_t fixed = (int64_t)2 * (L + 2) * 4 + (int64_t)(4 * L + 16) * 4 + (8 * L + 32);
    return ((fixed + (with_traceback ? ncol * tl : 64)) + 63) & ~(int64_t)63;
}


// BWAMEM_HIP_DUMP=1 (tiles of <= 64 reads): print chains and regions after extension (debugging aid only)
static void debug_dump(Workspace& ws, const TileView& tv, int T)
{
    (void)hipStreamSynchronize(ws.stream);
    std::vector<int64_t> so(T + 1); std::vector<int32_t> nc(T), nr(T);
    (void)hipMemcpy(so.data(), tv.seed_off, (T + 1) * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(nc.data(), tv.n_chains, T * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(nr.data(), tv.n_regs, T * 4, hipMemcpyDeviceToHost);
    for (int r = 0; r < T; ++r) {
        int64_t ns = so[r + 1] - so[r];
        fprintf(stderr, "[dump] read %d: seeds=%lld chains=%d regs=%d\n", r, (long long)ns, nc[r], nr[r]);
        if (ns <= 0 || ns > 4096) continue;
        std::vector<Chain> ch(ns); std::vector<Seed> cs(ns); std::vector<AlnReg> rg(ns);
        (void)hipMemcpy(ch.data(), tv.chains + so[r], ns * sizeof(Chain), hipMemcpyDeviceToHost);
        (void)hipMemcpy(cs.data(), tv.cseeds + so[r], ns * sizeof(Seed), hipMemcpyDeviceToHost);
        (void)hipMemcpy(rg.data(), tv.regs + so[r], ns * sizeof(AlnReg), hipMemcpyDeviceToHost);
        for (int i = 0; i < nc[r] && i < ns; ++i) {
            fprintf(stderr, "[dump]   chain %d: pos=%lld n=%d rid=%d w=%u kept=%d seed0=%d\n", i, (long long)ch[i].pos, ch[i].n, ch[i].rid, ch[i].w, ch[i].kept, ch[i].seed0);
            for (int j = 0; j < ch[i].n && ch[i].seed0 + j < ns; ++j) { const Seed& s = cs[ch[i].seed0 + j]; fprintf(stderr, "[dump]     seed rbeg=%lld qbeg=%d len=%d score=%d\n", (long long)s.rbeg, s.qbeg, s.len, s.score); }
        }
        for (int i = 0; i < nr[r] && i < ns; ++i)
            fprintf(stderr, "[dump]   reg %d: rb=%lld re=%lld qb=%d qe=%d rid=%d score=%d truesc=%d w=%d seedcov=%d seedlen0=%d\n", i, (long long)rg[i].rb, (long long)rg[i].re,
                    rg[i].qb, rg[i].qe, rg[i].rid, rg[i].score, rg[i].truesc, rg[i].w, rg[i].seedcov, rg[i].seedlen0);
    }
    fflush(stderr);
}

// Longest read the device path takes: the packed candidate format of the seeding kernel holds 17-bit positions.  (Reads
// beyond about 12 000 bases keep their DP rows in global memory instead of LDS: dp_rows_in_hbm.)
static bool read_length_ok(const bwaidx_s* ix, int L)
{
    (void)ix;
    if (L >= (1 << 17) - 1) { fprintf(stderr, "[bwamem_hip] reads of 131071 bases or more are not supported\n"); return false; }
    return true;
}

// ------------------------------------------------------------------------------------------ paired-end flow
// Phase 1 (per tile): seeding .. region de-duplication, exactly as single-end; the regions of every tile are
// kept compact in HBM.  Then the batch-global insert-size statistics (mem_pestat) are reduced on the host
// from per-pair (orientation, insert size) candidates -- the one cross-read dependency of the path
//  -- and phase 2 (per tile) does mate rescue, pairing and record generation.

static void host_pestat(const MemOpt& opt, const std::vector<int8_t>& dir, const std::vector<int64_t>& is, MemPestat pes[4])
{   // upstream mem_pestat after candidate collection (bwamem_pair.c).  Upstream sorts each orientation's insert sizes and
    // walks the sorted array; candidates are bounded by opt.max_ins, so the sorted array is represented by a histogram
    // (value -> multiplicity) and walked in the same order -- the same percentiles and the same sequence of double
    // additions, without an O(n log n) host sort of millions of values between the two phases of a call
    uint64_t vmax = 0;
    for (size_t i = 0; i < dir.size(); ++i) if (dir[i] >= 0 && (uint64_t)is[i] > vmax) vmax = (uint64_t)is[i];
    const bool use_hist = vmax < ((uint64_t)1 << 22);
    std::vector<uint64_t> isize[4];                 // sorted values (fallback) ...
    std::vector<uint32_t> hist[4];                  // ... or multiplicities
    size_t n_of[4] = { 0, 0, 0, 0 };
    if (use_hist) {
        for (int d = 0; d < 4; ++d) hist[d].assign((size_t)vmax + 1, 0);
        for (size_t i = 0; i < dir.size(); ++i) if (dir[i] >= 0) { ++hist[(int)dir[i]][(size_t)is[i]]; ++n_of[(int)dir[i]]; }
    } else {
        for (size_t i = 0; i < dir.size(); ++i) if (dir[i] >= 0) isize[(int)dir[i]].push_back((uint64_t)is[i]);
        for (int d = 0; d < 4; ++d) { std::sort(isize[d].begin(), isize[d].end()); n_of[d] = isize[d].size(); }
    }
    memset(pes, 0, 4 * sizeof(MemPestat));
    for (int d = 0; d < 4; ++d) {
        MemPestat* r = &pes[d];
        const size_t n = n_of[d];
        if (n < 10) { r->failed = 1; continue; }
        auto at = [&](size_t k) -> uint64_t {        // k-th smallest
            if (!use_hist) return isize[d][k];
            size_t c = 0;
            for (size_t v = 0; v < hist[d].size(); ++v) { c += hist[d][v]; if (c > k) return (uint64_t)v; }
            return vmax;
        };
        auto walk = [&](auto&& f) {                  // every value, ascending, with multiplicity
            if (!use_hist) { for (uint64_t v : isize[d]) f(v); return; }
            for (size_t v = 0; v < hist[d].size(); ++v) for (uint32_t c = hist[d][v]; c; --c) f((uint64_t)v);
        };
        int p25 = (int)at((size_t)(int)(.25 * n + .499));
        int p75 = (int)at((size_t)(int)(.75 * n + .499));
        r->low = (int)(p25 - 2.0 * (p75 - p25) + .499);
        if (r->low < 1) r->low = 1;
        r->high = (int)(p75 + 2.0 * (p75 - p25) + .499);
        size_t x = 0;
        r->avg = 0;
        walk([&](uint64_t v) { if (v >= (uint64_t)r->low && v <= (uint64_t)r->high) { r->avg += v; ++x; } });
        r->avg /= x;
        r->std = 0;
        walk([&](uint64_t v) { if (v >= (uint64_t)r->low && v <= (uint64_t)r->high) r->std += (v - r->avg) * (v - r->avg); });
        r->std = sqrt(r->std / x);
        r->low  = (int)(p25 - 3.0 * (p75 - p25) + .499);
        r->high = (int)(p75 + 3.0 * (p75 - p25) + .499);
        if (r->low  > r->avg - 4.0 * r->std) r->low  = (int)(r->avg - 4.0 * r->std + .499);
        if (r->high < r->avg + 4.0 * r->std) r->high = (int)(r->avg + 4.0 * r->std + .499);
        if (r->low < 1) r->low = 1;
    }
    size_t mx = 0;
    for (int d = 0; d < 4; ++d) mx = std::max(mx, n_of[d]);
    for (int d = 0; d < 4; ++d) if (pes[d].failed == 0 && n_of[d] < mx * 0.05) pes[d].failed = 1;
}

static bool align_batch_pe(bwaidx_s* ix, const MemOpt& opt, const MemPestat* pes0, bwamem_batch_s* b, int64_t read_id0);

// mem_pair scores a candidate pair with .721 * log(2 erfc(|dist - avg| / std / sqrt 2)) * a (upstream bwamem_pair.c), and the
// sum is truncated to an int: a decision that depends on libm.  dist is an integer in [low, high], so the host (glibc, the
// library the reference itself calls) evaluates the term for every distance the call can see and the device indexes the
// table (k_pe.hip: mem_pair).  Only distances within 40 standard deviations are stored: beyond |ns| / sqrt 2 >= 28 glibc's
// erfc is exactly 0 and the term is -inf (checked at the clipped ends).
static bool build_pair_tab(bwaidx_s* ix, const MemPestat* pes)
{
    PairTab& pt = ix->pair_tab;
    memset(&pt, 0, sizeof pt);
    std::vector<double> tab;
    const int64_t cap = (int64_t)1 << 24;
    for (int d = 0; d < 4; ++d) {
        if (pes[d].failed || pes[d].high < pes[d].low) continue;
        int64_t lo = pes[d].low, hi = pes[d].high;
        const double avg = pes[d].avg, sd = pes[d].std;
        auto term = [&](int64_t dist) { const double ns = (dist - avg) / sd; return log(2. * erfc(fabs(ns) * M_SQRT1_2)); };
        if (std::isfinite(avg) && std::isfinite(sd) && sd >= 0 && fabs(avg) < 1e15 && sd < 1e13) {
            const int64_t clo = (int64_t)floor(avg - 40. * sd) - 1, chi = (int64_t)ceil(avg + 40. * sd) + 1;
            const bool lo_ok = clo <= lo || term(clo) == -INFINITY, hi_ok = chi >= hi || term(chi) == -INFINITY;
            if (lo_ok && clo > lo) lo = clo;
            if (hi_ok && chi < hi) hi = chi;
        }
        if (hi < lo) continue;
        if (hi - lo + 1 > cap || (int64_t)tab.size() + (hi - lo + 1) > cap) {
            fprintf(stderr, "[bwamem_hip] insert-size range [%d, %d] of orientation %d is too wide to tabulate\n", pes[d].low, pes[d].high, d);
            return false;
        }
        pt.lo[d] = lo; pt.n[d] = (int32_t)(hi - lo + 1); pt.off[d] = (int32_t)tab.size();
        for (int64_t dist = lo; dist <= hi; ++dist) tab.push_back(term(dist));
    }
    if (!ix->d_ptab.ensure(tab.size() * 8 + 8)) return false;
    if (!tab.empty()) HIP_OK(hipMemcpy(ix->d_ptab.p, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
    pt.t = ix->d_ptab.as<double>();
    return true;
}

// a tile: reads [r0, r1) of the call; seq / seq_off: device pointers to the bases of the stretch the tile lies in and to
// the tile's r1 - r0 + 1 read offsets into it
struct TileSpec { uint32_t r0, r1; int L; uint8_t* seq; const int64_t* seq_off; };

// a stretch of the request resident in HBM: whole reads [r0, r1), offsets relative to seq (host and device copies)
struct ReqChunk { uint32_t r0 = 0, r1 = 0; uint8_t* seq = nullptr; const int64_t* d_off = nullptr; const int64_t* h_off = nullptr; };

static uint32_t max_tile_reads(bool even, int scale)
{
    const char* env_t = getenv("BWAMEM_HIP_TILE");
    uint32_t max_T = env_t && atoi(env_t) > 0 ? (uint32_t)atoi(env_t) : 393216u * (uint32_t)(scale > 0 ? scale : 1);
    if (even) max_T = std::max(2u, max_T & ~1u);
    return max_T;
}

// cut a stretch of the request into tiles from a per-workspace device-memory budget (pairs are never split when even = true)
static void plan_tiles(const ReqChunk& rc, const MemOpt& opt, bool even, bool with_traceback, int scale, std::vector<TileSpec>& tiles)
{
    const char* env_gb = getenv("BWAMEM_HIP_TILE_GB");
    const int64_t budget = (int64_t)(env_gb ? atoi(env_gb) : 24 * (scale > 0 ? scale : 1)) << 30;
    const uint32_t max_T = max_tile_reads(even, scale);
    const uint32_t n = rc.r1 - rc.r0;
    uint32_t r0 = 0;
    while (r0 < n) {
        int L0 = 1;
        uint32_t r1 = r0;
        while (r1 < n && r1 - r0 < max_T) {
            int len = (int)(rc.h_off[r1 + 1] - rc.h_off[r1] - 1);
            int L1 = std::max(L0, len);
            int64_t pr = (int64_t)std::max(64, L1 + 8) * (sizeof(Intv) + 4) + 2 * (int64_t)(L1 + 2) * 16 + 512 + post_bytes_per_read(L1, opt, with_traceback) + 64 * 300;
            if (r1 > r0 + (even ? 1u : 0u) && pr * (int64_t)(r1 - r0 + 1) > budget && (!even || ((r1 - r0) & 1) == 0)) break;
            L0 = L1; ++r1;
        }
        TileSpec t; t.r0 = rc.r0 + r0; t.r1 = rc.r0 + r1; t.L = L0; t.seq = rc.seq; t.seq_off = rc.d_off + r0;
        tiles.push_back(t);
        r0 = r1;
    }
}

// The seeding stage: k_seed + k_seed_fin over one chunk of reads (several tiles), on the seeding workspace's stream, into
// one of the two interval stores.
struct SeedChunk { uint32_t r0 = 0, r1 = 0; int L = 1; uint8_t* seq = nullptr; const int64_t* seq_off = nullptr; size_t tile0 = 0, tile1 = 0; int state = 0; size_t tiles_left = 0; };   // state: 0 pending, 1 ready, -1 failed

static bool seed_chunk(bwaidx_s* ix, const MemOpt& opt, const SeedChunk& ch, SeedStore& store, int& intv_cap_scale)
{
    Workspace& sw = ix->seed_ws;
    if (!sw.stream) {
        // the persistent seeding grid is gather-bound and needs only a few waves per CU: let the dispatcher place it first
        // (highest stream priority) and fill the remaining wave slots with the tiles' arithmetic-bound kernels
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        const char* e = getenv("BWAMEM_HIP_SEED_PRIO");
        const int prio = e ? (atoi(e) > 0 ? hi : atoi(e) < 0 ? lo : 0) : hi;
        HIP_OK(hipStreamCreateWithPriority(&sw.stream, hipStreamNonBlocking, prio));
    }
    RoctxRange rr("bwamem_hip:seed_chunk");
    const int T = (int)(ch.r1 - ch.r0), L = ch.L;
    const int max_groups = 8192;                                 // upper bound on the persistent k_seed grid
    const int groups = std::min(max_groups, (T + 63) / 64);
    for (int attempts = 0; attempts < 8; ++attempts) {
        const int cap = std::max(64, L + 8) * intv_cap_scale;
        if (!store.ensure((size_t)T, cap)) return false;
        if (!sw.err.ensure(64) || !sw.cnt.ensure(sizeof(DevCounters)) || !sw.smem.ensure((size_t)groups * (L + 2) * 64 * 16)) return false;
        TileView tv; memset(&tv, 0, sizeof tv);
        tv.n_reads = T; tv.max_len = L;
        tv.seq = ch.seq; tv.seq_off = ch.seq_off;
        tv.intv_cap = cap; tv.intv = store.intv.as<Intv>(); tv.n_intv = store.n_intv.as<int32_t>();
        tv.intv_seed_off = store.intv_seed_off.as<int32_t>(); tv.n_seeds = store.n_seeds.as<int32_t>(); tv.l_rep = store.l_rep.as<int32_t>();
        tv.smem_scratch = sw.smem.as<Intv>(); tv.smem_cap = L + 2; tv.smem_groups = groups;
        tv.err = sw.err.as<int32_t>(); tv.cnt = sw.cnt.as<DevCounters>();
        HIP_OK(hipMemsetAsync(sw.err.p, 0, 64, sw.stream));
        HIP_OK(hipMemsetAsync(sw.cnt.p, 0, sizeof(DevCounters), sw.stream));
        TIMED(sw, K_SEED, launch_seed(sw.stream, ix->d, opt, tv));
        int32_t h[16]; DevCounters hc;
        HIP_OK(hipMemcpyAsync(h, tv.err, sizeof h, hipMemcpyDeviceToHost, sw.stream));
        HIP_OK(hipMemcpyAsync(&hc, tv.cnt, sizeof hc, hipMemcpyDeviceToHost, sw.stream));
        HIP_OK(hipStreamSynchronize(sw.stream));
        timed_collect(sw);
        if (h[0] & ERR_INTV_CAP) { intv_cap_scale *= 2; { std::lock_guard<std::mutex> lk(g_stats.mu); ++g_stats.s.n_retries; } continue; }
        { std::lock_guard<std::mutex> lk(g_stats.mu); g_stats.s.n_ext += hc.n_ext; }
        return true;
    }
    fprintf(stderr, "[bwamem_hip] seeding chunk could not be sized after 8 attempts\n");
    return false;
}

// a tile's view of its reads and of its slice of the chunk's interval lists
static TileView tile_view(const Workspace& ws, const TileSpec& spec, int64_t read_id0, const SeedStore& seeds_of_chunk, uint32_t chunk_r0)
{
    TileView v = ws.view();
    v.n_reads = (int)(spec.r1 - spec.r0); v.max_len = spec.L; v.read_id0 = read_id0 + spec.r0;
    v.seq = spec.seq; v.seq_off = spec.seq_off;
    const size_t at = (size_t)(spec.r0 - chunk_r0);
    v.intv_cap = seeds_of_chunk.cap;
    v.intv = seeds_of_chunk.intv.as<Intv>() + at * seeds_of_chunk.cap;
    v.intv_seed_off = seeds_of_chunk.intv_seed_off.as<int32_t>() + at * seeds_of_chunk.cap;
    v.n_intv = seeds_of_chunk.n_intv.as<int32_t>() + at;
    v.n_seeds = seeds_of_chunk.n_seeds.as<int32_t>() + at;
    v.l_rep = seeds_of_chunk.l_rep.as<int32_t>() + at;
    v.smem_scratch = nullptr; v.smem_groups = 0;
    { const char* e = getenv("BWAMEM_HIP_ORDER"); if (!(e && atoi(e) == 0)) v.order = ws.order.as<int32_t>() + 64; }   // filled by launch_order below the seed-count scan
    return v;
}

// the packed records of a finished tile: into the response (host mode: copied out now, from the workspace's packing
// buffer) or into the batch's device slab
static bool emit_tile(Workspace& ws, bwamem_batch_s* b, size_t tile_index, const TileView& tv, int64_t out_total, TileOut& to)
{
    RoctxRange rr("bwamem_hip:tile_emit");
    OutSink& sink = b->sink;
    uint8_t* dst = nullptr;
    to.bytes = (size_t)out_total; to.d = nullptr; to.owned = false;
    if (!sink.place(tile_index, out_total, tv.n_reads, &dst)) return false;
    if (sink.to_host) {
        bool ok = true;
        if (out_total > 0) {
            ok = ws.packed.ensure((size_t)out_total);
            if (ok) {
                timed_begin(ws, K_PACK); launch_pack(ws.stream, tv, ws.packed.as<uint8_t>()); timed_end(ws);
                ok = hipGetLastError() == hipSuccess
                  && hipMemcpyAsync(dst, ws.packed.p, (size_t)out_total, hipMemcpyDeviceToHost, ws.stream) == hipSuccess
                  && hipStreamSynchronize(ws.stream) == hipSuccess;
            }
        }
        sink.copy_done();
        if (!ok) fprintf(stderr, "[bwamem_hip] response download failed\n");
        return ok;
    }
    if (out_total > 0) {
        if (dst) to.d = dst;
        else { HIP_OK(hipMalloc((void**)&to.d, (size_t)out_total)); to.owned = true; }
        TIMED(ws, K_PACK, launch_pack(ws.stream, tv, to.d));
        HIP_OK(hipStreamSynchronize(ws.stream));
    }
    return true;
}

// one single-end tile, start to packed response, on the workspace's own stream
static bool run_tile_se(bwaidx_s* ix, Workspace& ws, const MemOpt& opt, bwamem_batch_s* b, int64_t read_id0, size_t tile_index, const TileSpec& spec,
                        TileOut& to, const SeedStore& seeds_of_chunk, uint32_t chunk_r0, int& out_cap)
{
    RoctxRange rr("bwamem_hip:tile_se");
    const uint32_t r0 = spec.r0;
    const int T = (int)(spec.r1 - spec.r0), L = spec.L;
    int attempts = 0, job_cap_hint = 0;
    bool rescore_full = false;
    size_t zpool_hint = (size_t)64 << 20;
    double zmult = 1;
    { int oc = 0; ix->hints.get(T, job_cap_hint, zmult, oc); out_cap = std::max(out_cap, oc); }
    for (;;) {
        if (++attempts > 8) { fprintf(stderr, "[bwamem_hip] tile could not be sized after 8 attempts\n"); return false; }
        if (!ws.ensure_reads(opt, T, L, seeds_of_chunk.cap, out_cap, post_bytes_per_read(L, opt, false), false)) return false;
        if (!ws.ensure_seeds(std::max<int64_t>(ws.seed_cap, (int64_t)T * 16))) return false;
        {   // the HBM pool also holds the direction nibbles of k_gcigar_lane: 20 bytes per target row of every job
            const int jc = std::max(job_cap_hint, std::max(1024, T / 4));
            size_t zdef = (size_t)jc * (size_t)(2 * L + 64) * 20 + ((size_t)64 << 20);
            if (gcigar_slab_bytes(opt, L)) zdef = std::min(zdef, (size_t)1 << 30);          // long reads: the wave form's matrices live in its slabs
            if (!ws.ensure_jobs(jc, 4 * L + 16, std::max(zpool_hint, (size_t)((double)zdef * zmult)))) return false;
        }
        TileView tv = tile_view(ws, spec, read_id0, seeds_of_chunk, chunk_r0);
        HIP_OK(hipMemsetAsync(ws.err.p, 0, 64, ws.stream));
        HIP_OK(hipMemsetAsync(ws.cnt.p, 0, sizeof(DevCounters), ws.stream));
        HIP_OK(hipMemsetAsync(ws.job_cnt.p, 0, 64, ws.stream));
        TIMED(ws, K_OTHER, launch_scan(ws.stream, tv.n_seeds, tv.seed_off, T, ws.scan_tmp.as<int64_t>()));
        if (tv.order) TIMED(ws, K_OTHER, launch_order(ws.stream, tv.n_seeds, T, ws.order.as<int32_t>(), ws.order.as<int32_t>() + 64));
        int64_t n_occ = 0; int32_t err = 0; int32_t errv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        HIP_OK(hipMemcpyAsync(&n_occ, tv.seed_off + T, 8, hipMemcpyDeviceToHost, ws.stream));
        HIP_OK(hipStreamSynchronize(ws.stream));
        ix->hints.saw_seeds(T, n_occ);
        if (n_occ > ws.seed_cap) {
            if (!ws.ensure_seeds(n_occ + n_occ / 4)) return false;
            tv = tile_view(ws, spec, read_id0, seeds_of_chunk, chunk_r0);
        }
        TIMED(ws, K_SA, launch_sa(ws.stream, ix->d, opt, tv, n_occ));
        TIMED(ws, K_CHAIN, launch_chain(ws.stream, ix->d, opt, tv, ws.chain_store.as<Chain>()));
        if (rescore_needed(opt, tv)) {                     // long reads: seed re-scoring jobs (at most one per seed occurrence)
            if (n_occ + 16 > 0x7fffffff) { fprintf(stderr, "[bwamem_hip] tile with more than 2^31 seed occurrences\n"); return false; }
            // one job per seed occurrence at most, so n_occ slots always fit; BWAMEM_HIP_RESCORE_CAP0 (tests) makes the first
            // attempt too small: the plan kernel then voids the list and the tile is run again with the full size
            static const int cap0 = []{ const char* e = getenv("BWAMEM_HIP_RESCORE_CAP0"); return e && atoi(e) > 0 ? atoi(e) : 0; }();
            const int rc = cap0 && !rescore_full ? cap0 : (int)(n_occ + 16);
            if (!(ws.pe_rescue[0].ensure(pe_rescue_bytes(0, rc)) && ws.pe_rescue[1].ensure(pe_rescue_bytes(1, rc)) && ws.pe_rescue[2].ensure((size_t)T * 8 + 64))) return false;
            TIMED(ws, K_CHAIN, launch_rescore(ws.stream, ix->d, opt, tv, ws.pe_rescue[0].p, ws.pe_rescue[1].p, ws.pe_rescue[2].as<int32_t>() + 16, ws.pe_rescue[2].as<int32_t>(), rc));
        }
        TIMED(ws, K_EXTEND, launch_extend(ws.stream, ix->d, opt, tv));
        if (getenv("BWAMEM_HIP_DUMP") && T <= 64) debug_dump(ws, tv, T);
        TIMED(ws, K_POST, launch_post1(ws.stream, ix->d, opt, tv));
        if (L > 1000) {
            // long reads: regions per read vary by orders of magnitude, and a tile re-run costs seconds.  Every job is a region,
            // so the region count (known now) bounds the job list: size it before the jobs are listed instead of retrying
            int64_t n_regs_total = 0;
            TIMED(ws, K_OTHER, launch_scan(ws.stream, tv.n_regs, tv.out_off, T, ws.scan_tmp.as<int64_t>()));
            HIP_OK(hipMemcpyAsync(&n_regs_total, tv.out_off + T, 8, hipMemcpyDeviceToHost, ws.stream));
            HIP_OK(hipStreamSynchronize(ws.stream));
            if (n_regs_total > ws.job_cap) {
                if (!ws.ensure_jobs((int)std::min<int64_t>(n_regs_total + 64, 0x7fffffff), 4 * L + 16, ws.zpool_cap)) return false;
                tv = tile_view(ws, spec, read_id0, seeds_of_chunk, chunk_r0);
                HIP_OK(hipMemsetAsync(ws.job_cnt.p, 0, 64, ws.stream));
            }
        }
        TIMED(ws, K_FINAL, launch_final_prep(ws.stream, ix->d, opt, tv));
        int32_t n_jobs = 0;
        HIP_OK(hipMemcpyAsync(&n_jobs, tv.job_cnt, 4, hipMemcpyDeviceToHost, ws.stream));
        HIP_OK(hipStreamSynchronize(ws.stream));
        if (n_jobs > ws.job_cap) { job_cap_hint = n_jobs + n_jobs / 4; { std::lock_guard<std::mutex> lk(g_stats.mu); ++g_stats.s.n_retries; } continue; }
        ws.zslab_bytes = gcigar_slab_bytes(opt, L);
        if (ws.zslab_bytes && !ws.zslabs.ensure(ws.zslab_bytes * (size_t)gcigar_slab_grid(ix->d, 1 << 30))) return false;
        TIMED(ws, K_FINAL, launch_gcigar(ws.stream, ix->d, opt, tv, n_jobs, ws.jobs.p, ws.job_out.p, ws.job_cig.as<uint32_t>(), ws.job_cig_cap,
                                         ws.zpool.as<uint8_t>(), (unsigned long long)ws.zpool_cap, (unsigned long long*)(ws.job_cnt.as<int32_t>() + 2),
                                         ws.zslabs.as<uint8_t>(), ws.zslab_bytes, ws.job_cnt.as<int32_t>() + 4));
        TIMED(ws, K_FINAL, launch_final_se(ws.stream, ix->d, opt, tv, ws.job_out.p, ws.job_cig.as<uint32_t>(), ws.job_cig_cap));
        TIMED(ws, K_OTHER, launch_scan(ws.stream, tv.out_len, tv.out_off, T, ws.scan_tmp.as<int64_t>()));
        int64_t out_total = 0;
        DevCounters hc;
        HIP_OK(hipMemcpyAsync(&out_total, tv.out_off + T, 8, hipMemcpyDeviceToHost, ws.stream));
        HIP_OK(hipMemcpyAsync(&hc, tv.cnt, sizeof hc, hipMemcpyDeviceToHost, ws.stream));
        HIP_OK(hipMemcpyAsync(errv, tv.err, sizeof errv, hipMemcpyDeviceToHost, ws.stream));
        HIP_OK(hipStreamSynchronize(ws.stream));
        err = errv[0];
        if (tv.debug & 0x2000) fprintf(stderr, "[bwamem_hip] global alignment, Mclk summed over waves: rows %.1f tile staging %.1f walk %.1f end %.1f | jobs %d\n", hc.dbg[5] / 1e6, hc.dbg[6] / 1e6, hc.dbg[7] / 1e6, hc.dbg[8] / 1e6, n_jobs);
        if (err) {
            { std::lock_guard<std::mutex> lk(g_stats.mu); ++g_stats.s.n_retries; }
            if (err & ERR_RESCUE_CAP) { if (rescore_full) { fprintf(stderr, "[bwamem_hip] internal error: seed re-scoring list beyond the seed count\n"); return false; } rescore_full = true; continue; }
            if (err & ERR_BAD_REG) { fprintf(stderr, "[bwamem_hip] internal error: extension produced an invalid region (tile read %d of call read %lld: n=%d qb=%d qe=%d rb=%d re=%d score=%d)\n", errv[1], (long long)(read_id0 + r0 + errv[1]), errv[2], errv[3], errv[4], errv[5], errv[6], errv[7]); return false; }
            if (err & ERR_LONG_READ) { fprintf(stderr, "[bwamem_hip] reads long enough to need seed re-scoring (mem_flt_chained_seeds) are not supported on the device path yet\n"); return false; }
            if (err & ERR_BTREE) { fprintf(stderr, "[bwamem_hip] internal error: chain B-tree pool exhausted\n"); return false; }
            if ((err & (ERR_SCRATCH | ERR_CIGAR_CAP)) && !(err & (ERR_ZPOOL | ERR_JOB_CAP))) { fprintf(stderr, "[bwamem_hip] internal error: post-processing scratch exhausted (err=%d)\n", err); return false; }
            if (err & ERR_OUT_CAP) { out_cap *= 4; continue; }
            if (err & ERR_ZPOOL) { zpool_hint = std::max(zpool_hint * 4, ws.zpool_cap * 4); continue; }
            if (err & ERR_JOB_CAP) { job_cap_hint = std::max(ws.job_cap * 2, 4096); continue; }
            fprintf(stderr, "[bwamem_hip] device error flags %d\n", err); return false;
        }
        if (!emit_tile(ws, b, tile_index, tv, out_total, to)) return false;
        ix->hints.learn(T, n_jobs, zpool_hint > ((size_t)64 << 20) ? (double)zpool_hint / ((double)std::max(job_cap_hint, std::max(1024, T / 4)) * (2.0 * L + 64) * 20 + (double)((size_t)64 << 20)) : 1.0, out_cap);
        {
            std::lock_guard<std::mutex> lk(g_stats.mu);
            g_stats.s.n_reads += T; g_stats.s.n_ext += hc.n_ext; g_stats.s.n_lf += hc.n_lf; g_stats.s.n_sa += hc.n_sa;
            g_stats.s.n_dp_cells += hc.n_dp_cells; ++g_stats.s.n_tiles;
        }
        timed_collect(ws);
        return true;
    }
}

// ---------------------------------------------------------------------------------------- the call pipeline
// Three kinds of host threads share one call: a PRODUCER makes stretches of the request resident (all at once for a
// batch that already lives in HBM; for jnibwa_createAlignments by streaming the caller's buffer to the device while the
// first stretches are being aligned) and cuts them into tiles and seeding chunks; the SEEDER runs k_seed one chunk ahead
// of the tiles; WORKERS (one stream + workspace each, BWAMEM_HIP_STREAMS of them) take the tiles in order.
struct CallPipe {
    std::mutex mu; std::condition_variable cv;
    std::deque<TileSpec> specs; std::deque<SeedChunk> chunks; std::deque<size_t> chunk_of;
    bool produced_all = false, failed = false;
    size_t next_tile = 0;
    std::function<void(size_t)> on_new_tiles;        // called (under mu) with the number of tiles just appended
    void fail() { { std::lock_guard<std::mutex> lk(mu); failed = true; } cv.notify_all(); }
};

// Seeding chunks: runs of consecutive tiles.  k_seed cannot finish before its slowest read has (one lane walks one read),
// so a launch over one tile spends much of its time in a thin tail; over a few million reads the queue keeps the lanes fed
// for most of the launch.  The first chunks are small (one tile, then two, ...) so that the tile workers start early.
static void append_stretch(CallPipe& pp, const ReqChunk& rc, const MemOpt& opt, bool even, int tile_scale)
{
    // the larger tile is for short reads in repeats; long reads carry thousands of seeds each whatever the reference, and twice
    // their workspaces does not fit four times next to the index
    { const uint32_t n = rc.r1 - rc.r0; if (n && (rc.h_off[n] - rc.h_off[0]) / n > 1000) tile_scale = 1; }
    // (paired-end calls took the standard tile while the pairing stage was one lane per pair and lasted as long as its heaviest pair;
    // with the incremental list maintenance and a wavefront per heavy pair the larger tile pays here too: human-like genome + 3 %)
                                            // hide that better than larger ones amortise it (human-like genome, 4 M reads: 0.45 M reads/s with twice the reads per tile, 0.69 M without)
    std::vector<TileSpec> tiles;
    plan_tiles(rc, opt, even, false, tile_scale, tiles);
    const char* e = getenv("BWAMEM_HIP_SEED_CHUNK");
    const uint32_t chunk_reads = e && atoi(e) > 0 ? (uint32_t)atoi(e) : 2097152u;
    const char* eg = getenv("BWAMEM_HIP_SEED_GB");
    const int64_t budget = (int64_t)(eg && atoi(eg) > 0 ? atoi(eg) : 16) << 30;     // per interval store
    std::lock_guard<std::mutex> lk(pp.mu);
    bool fresh = true;                                                               // a chunk never spans two stretches (separate buffers)
    for (const TileSpec& t : tiles) {
        const size_t i = pp.specs.size();
        const int64_t n_new = fresh ? 0 : (int64_t)(t.r1 - pp.chunks.back().r0);
        const int L_new = fresh ? 1 : std::max(pp.chunks.back().L, t.L);
        const int64_t first_T = (int64_t)(pp.specs.empty() ? t.r1 - t.r0 : pp.specs[0].r1 - pp.specs[0].r0);
        const int64_t ramp = std::min<int64_t>((int64_t)chunk_reads, first_T << std::min<size_t>(pp.chunks.empty() ? 0 : pp.chunks.size() - (fresh ? 0 : 1), 8));
        if (fresh || n_new > ramp || n_new * std::max(64, L_new + 8) * 36 > budget) {
            SeedChunk c; c.r0 = t.r0; c.tile0 = i; c.L = 1; c.seq = t.seq; c.seq_off = t.seq_off;
            pp.chunks.push_back(c);
            fresh = false;
        }
        SeedChunk& c = pp.chunks.back();
        c.r1 = t.r1; c.tile1 = i + 1; c.L = std::max(c.L, t.L); c.tiles_left = c.tile1 - c.tile0;
        pp.specs.push_back(t);
        pp.chunk_of.push_back(pp.chunks.size() - 1);
    }
    if (pp.on_new_tiles) pp.on_new_tiles(tiles.size());
    pp.cv.notify_all();
}

// number of NUL bytes in [p, p + n)
static size_t count_zero_bytes(const char* p, size_t n) { size_t c = 0; for (size_t i = 0; i < n; ++i) c += p[i] == 0; return c; }

// The end (exclusive, right after a NUL) of the stretch of the request starting at p that holds `want` reads, or fewer once
// it is max_bytes long; *got = the reads in it.  The request carries no length (jnibwa.c:204-212 walks it with strlen), so
// this walk is what bounds every access to the caller's buffer.  It counts NULs a page at a time: the bytes up to the end
// of the page p points into are readable whenever p itself is inside the request (the classic strlen argument).
static const char* scan_reads(const char* p0, uint64_t want, size_t max_bytes, uint64_t* got)
{
    const char* q = p0;
    uint64_t have = 0;
#if defined(__SANITIZE_ADDRESS__) || defined(BWAMEM_HIP_EXACT_WALK)
    while (have < want) { q += strlen(q) + 1; ++have; if ((size_t)(q - p0) >= max_bytes) break; }
#else
    while (have < want) {
        const size_t n = 4096 - ((uintptr_t)q & 4095);
        const size_t c = count_zero_bytes(q, n);
        if (have + c < want && (size_t)(q + n - p0) < max_bytes) { have += c; q += n; continue; }
        const char* e = q + n;
        bool done = false;
        while (q < e) {                                           // the stop lies in this page (or the page has no NUL at all)
            const char* z = (const char*)memchr(q, 0, (size_t)(e - q));
            if (!z) { q = e; break; }
            q = z + 1; ++have;
            if (have == want || (size_t)(q - p0) >= max_bytes) { done = true; break; }
        }
        if (done) break;
    }
#endif
    *got = have;
    return q;
}

// producer of a streamed call: walk, upload, find the read offsets on the device, encode -- stretch by stretch
static bool produce_streamed(bwaidx_s* ix, CallPipe& pp, bwamem_batch_s* b, const MemOpt& opt, bool even)
{
    HIP_OK(hipSetDevice(ix->device));
    if (!ix->up_stream) HIP_OK(hipStreamCreateWithFlags(&ix->up_stream, hipStreamNonBlocking));
    hipStream_t st = ix->up_stream;
    const uint32_t max_T = max_tile_reads(even, 1);
    const char* e = getenv("BWAMEM_HIP_SEED_CHUNK");
    const uint64_t chunk_reads = e && atoi(e) > 0 ? (uint64_t)atoi(e) : 2097152u;
    const char* eb = getenv("BWAMEM_HIP_UPLOAD_BYTES");
    const size_t max_bytes = eb && atoll(eb) > 0 ? (size_t)atoll(eb) : (size_t)384 << 20;
    const char* p = b->h_payload;
    uint64_t r = 0;
    size_t k = 0;
    while (r < b->n_reads) {
        { std::lock_guard<std::mutex> lk(pp.mu); if (pp.failed) return false; }
        uint64_t want = std::min<uint64_t>(chunk_reads, (uint64_t)max_T << std::min<size_t>(k, 8));
        want = std::min<uint64_t>(std::max<uint64_t>(want, even ? 2 : 1), b->n_reads - r);
        if (even && (want & 1) && r + want < b->n_reads) ++want;
        RoctxRange rr("bwamem_hip:request_stretch");
        uint64_t got = 0;
        const char* end = scan_reads(p, want, max_bytes, &got);
        if (even && (got & 1) && r + got < b->n_reads) { uint64_t one = 0; end = scan_reads(end, 1, (size_t)-1, &one); got += one; }   // a pair is never split
        const size_t nbytes = (size_t)(end - p);
        if (nbytes >= ((size_t)1 << 40)) { fprintf(stderr, "[bwamem_hip] request stretch too large\n"); return false; }
        if (k >= ix->req_bufs.size()) ix->req_bufs.push_back(new ReqBuf());
        ReqBuf& rb = *ix->req_bufs[k];
        if (!rb.seq.ensure(nbytes + 64) || !rb.off.ensure((got + 2) * 8 + 16) || !rb.tmp.ensure(nul_tmp_bytes((int64_t)nbytes)) || !rb.ensure_host(got + 2)) return false;
        int64_t* d_found = rb.off.as<int64_t>() + got + 1;           // one spare slot behind the offsets
        HIP_OK(hipMemcpyAsync(rb.seq.p, p, nbytes, hipMemcpyHostToDevice, st));
        launch_nul_offsets(st, rb.seq.as<uint8_t>(), (int64_t)nbytes, rb.off.as<int64_t>(), (int64_t)got, d_found, rb.tmp.p);
        launch_encode(st, rb.seq.as<uint8_t>(), (int64_t)nbytes);
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpyAsync(rb.h_off, rb.off.p, (got + 2) * 8, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        if (rb.h_off[got + 1] != (int64_t)got || rb.h_off[got] != (int64_t)nbytes) {
            fprintf(stderr, "[bwamem_hip] internal error: device read offsets disagree with the host walk (%lld reads / %lld bytes against %llu / %zu)\n",
                    (long long)rb.h_off[got + 1], (long long)rb.h_off[got], (unsigned long long)got, nbytes);
            return false;
        }
        ReqChunk rc; rc.r0 = (uint32_t)r; rc.r1 = (uint32_t)(r + got); rc.seq = rb.seq.as<uint8_t>(); rc.d_off = rb.off.as<int64_t>(); rc.h_off = rb.h_off;
        append_stretch(pp, rc, opt, even, ix->hints.tile_scale());
        b->n_bytes += nbytes;
        p = end; r += got; ++k;
    }
    return true;
}

// Runs fn for every tile of the call (see CallPipe).  fn finds the interval lists of its tile in the store it is handed
// (at offset r0 - chunk_r0).
typedef std::function<bool(Workspace&, size_t, const TileSpec&, const SeedStore&, uint32_t)> TileFn;
static bool run_pipeline(bwaidx_s* ix, const MemOpt& opt, bwamem_batch_s* b, bool even, CallPipe& pp, const TileFn& fn)
{
    const char* env_s = getenv("BWAMEM_HIP_STREAMS");
    const int n_workers = std::max(1, env_s ? atoi(env_s) : 4);
    while ((int)ix->extra_ws.size() < n_workers - 1) ix->extra_ws.push_back(new Workspace());
    // BWAMEM_HIP_SEED_AHEAD=0 serialises the seeding behind the tiles (isolated kernel timings)
    const bool seed_ahead = !(getenv("BWAMEM_HIP_SEED_AHEAD") && atoi(getenv("BWAMEM_HIP_SEED_AHEAD")) == 0);
    // interval stores in the ring (BWAMEM_HIP_SEED_STORES, 2 .. SEED_STORES_MAX): with two the seeding kernel waits for the tiles of the
    // chunk before last, and a 10 M-read call has no seeding on the device for a third of its time
    const char* env_ns = getenv("BWAMEM_HIP_SEED_STORES");
    const int n_stores = std::max(2, std::min<int>(SEED_STORES_MAX, env_ns ? atoi(env_ns) : 3));
    auto fail = [&]() { pp.fail(); b->sink.abort(); };
    auto producer = [&]() {
        bool ok = false;
        try {
            if (b->h_payload) ok = produce_streamed(ix, pp, b, opt, even);
            else {
                ReqChunk rc; rc.r0 = 0; rc.r1 = b->n_reads; rc.seq = b->d_seq.as<uint8_t>(); rc.d_off = b->d_off.as<int64_t>(); rc.h_off = b->h_off.data();
                append_stretch(pp, rc, opt, even, ix->hints.tile_scale());
                ok = true;
            }
        } catch (const std::exception& ex) { fprintf(stderr, "[bwamem_hip] %s\n", ex.what()); }
        if (!ok) fail();
        { std::lock_guard<std::mutex> lk(pp.mu); pp.produced_all = true; }
        pp.cv.notify_all();
    };
    auto seeder = [&]() {
        try {
            if (hipSetDevice(ix->device) != hipSuccess) { fail(); return; }
            int intv_cap_scale = 1;
            for (size_t c = 0; ; ++c) {
                SeedChunk ch;
                {
                    std::unique_lock<std::mutex> lk(pp.mu);
                    pp.cv.wait(lk, [&] { return pp.failed || c < pp.chunks.size() || pp.produced_all; });
                    if (pp.failed || c >= pp.chunks.size()) break;
                    pp.cv.wait(lk, [&] {                                       // the store of chunk c was last used by chunk c - n_stores
                        if (pp.failed) return true;
                        for (size_t k = 0; k < c; ++k) if ((k + (size_t)n_stores <= c || !seed_ahead) && pp.chunks[k].tiles_left != 0) return false;
                        return true;
                    });
                    if (pp.failed) break;
                    ch = pp.chunks[c];
                }
                const bool ok = seed_chunk(ix, opt, ch, ix->seed_store[c % (size_t)n_stores], intv_cap_scale);
                { std::lock_guard<std::mutex> lk(pp.mu); pp.chunks[c].state = ok ? 1 : -1; }
                if (!ok) { fail(); break; }
                pp.cv.notify_all();
            }
        } catch (const std::exception& ex) { fprintf(stderr, "[bwamem_hip] %s\n", ex.what()); fail(); }
    };
    auto worker = [&](int k) {
        try {
            Workspace& w = k == 0 ? ix->ws : *ix->extra_ws[k - 1];
            w.dev_lds = ix->d.lds_bytes;
            if (hipSetDevice(ix->device) != hipSuccess) { fail(); return; }
            if (!w.stream && hipStreamCreate(&w.stream) != hipSuccess) { fail(); return; }
            for (;;) {
                size_t i, c; TileSpec sp; uint32_t chunk_r0;
                {
                    std::unique_lock<std::mutex> lk(pp.mu);
                    pp.cv.wait(lk, [&] { return pp.failed || pp.next_tile < pp.specs.size() || pp.produced_all; });
                    if (pp.failed || pp.next_tile >= pp.specs.size()) break;
                    i = pp.next_tile++; sp = pp.specs[i]; c = pp.chunk_of[i];
                    pp.cv.wait(lk, [&] { return pp.failed || pp.chunks[c].state != 0; });
                    if (pp.failed || pp.chunks[c].state < 0) break;
                    chunk_r0 = pp.chunks[c].r0;
                }
                const bool ok = fn(w, i, sp, ix->seed_store[c % (size_t)n_stores], chunk_r0);
                { std::lock_guard<std::mutex> lk(pp.mu); --pp.chunks[c].tiles_left; }
                pp.cv.notify_all();
                if (!ok) { fail(); break; }
            }
        } catch (const std::exception& ex) { fprintf(stderr, "[bwamem_hip] %s\n", ex.what()); fail(); }
    };
    std::vector<std::thread> th;
    th.emplace_back(producer);
    th.emplace_back(seeder);
    for (int k = 1; k < n_workers; ++k) th.emplace_back(worker, k);
    worker(0);
    for (std::thread& t : th) t.join();
    return !pp.failed;
}

// ---------------------------------------------------------------------------------------- paired-end call
// Phase 1 (per tile, tiles in flight like single-end, seeding in chunks): seeds .. regions of every read, kept per tile,
// and the per-pair insert-size candidates.  Then the batch-global statistics on the host (mem_pestat), unless the caller
// supplied them.  Phase 2 (per tile, tiles in flight): mate rescue, pairing, DP jobs, records.
struct PeTile { uint32_t r0 = 0; int T = 0, L = 0; uint8_t* seq = nullptr; const int64_t* seq_off = nullptr; DevBuf n_regs, regs, reg_off; int64_t n_regs_total = 0; std::vector<int8_t> cand_dir; std::vector<int64_t> cand_is; };

#define PE_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fprintf(stderr, "[bwamem_hip] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return false; } } while (0)
#define PE_REQ(cond) do { if (!(cond)) return false; } while (0)

static bool pe_phase1_tile(bwaidx_s* ix, Workspace& ws, const MemOpt& opt, bwamem_batch_s* b, int64_t read_id0, const TileSpec& spec, PeTile* pt,
                           const SeedStore& seeds_of_chunk, uint32_t chunk_r0, bool want_cand)
{
    RoctxRange rr("bwamem_hip:tile_pe_phase1");
    const uint32_t r0 = spec.r0;
    const int T = (int)(spec.r1 - spec.r0), L = spec.L;
    PE_REQ(ws.ensure_reads(opt, T, L, seeds_of_chunk.cap, ws.out_cap_hint, post_bytes_per_read(L, opt, false), false));
    PE_REQ(ws.ensure_seeds(std::max<int64_t>(ws.seed_cap, (int64_t)T * 16)));
    auto make_view = [&]() { return tile_view(ws, spec, read_id0, seeds_of_chunk, chunk_r0); };
    TileView tv = make_view();
    PE_OK(hipMemsetAsync(ws.err.p, 0, 64, ws.stream));
    PE_OK(hipMemsetAsync(ws.cnt.p, 0, sizeof(DevCounters), ws.stream));
    TIMED(ws, K_OTHER, launch_scan(ws.stream, tv.n_seeds, tv.seed_off, T, ws.scan_tmp.as<int64_t>()));
    if (tv.order) TIMED(ws, K_OTHER, launch_order(ws.stream, tv.n_seeds, T, ws.order.as<int32_t>(), ws.order.as<int32_t>() + 64));
    int64_t n_occ = 0; int32_t err = 0;
    PE_OK(hipMemcpyAsync(&n_occ, tv.seed_off + T, 8, hipMemcpyDeviceToHost, ws.stream));
    PE_OK(hipStreamSynchronize(ws.stream));
    ix->hints.saw_seeds(T, n_occ);
    if (n_occ > ws.seed_cap) { PE_REQ(ws.ensure_seeds(n_occ + n_occ / 4)); tv = make_view(); }
    TIMED(ws, K_SA, launch_sa(ws.stream, ix->d, opt, tv, n_occ));
    TIMED(ws, K_CHAIN, launch_chain(ws.stream, ix->d, opt, tv, ws.chain_store.as<Chain>()));
    if (rescore_needed(opt, tv)) {                     // long reads: seed re-scoring jobs (at most one per seed occurrence)
        if (n_occ + 16 > 0x7fffffff) { fprintf(stderr, "[bwamem_hip] tile with more than 2^31 seed occurrences\n"); return false; }
        const int rc = (int)(n_occ + 16);                  // one job per seed occurrence at most: the list always fits (see run_tile_se)
        if (!(ws.pe_rescue[0].ensure(pe_rescue_bytes(0, rc)) && ws.pe_rescue[1].ensure(pe_rescue_bytes(1, rc)) && ws.pe_rescue[2].ensure((size_t)T * 8 + 64))) return false;
        TIMED(ws, K_CHAIN, launch_rescore(ws.stream, ix->d, opt, tv, ws.pe_rescue[0].p, ws.pe_rescue[1].p, ws.pe_rescue[2].as<int32_t>() + 16, ws.pe_rescue[2].as<int32_t>(), rc));
    }
    TIMED(ws, K_EXTEND, launch_extend(ws.stream, ix->d, opt, tv));
    TIMED(ws, K_POST, launch_post1(ws.stream, ix->d, opt, tv));
    pt->r0 = r0; pt->T = T; pt->L = L; pt->seq = spec.seq; pt->seq_off = spec.seq_off;
    PE_REQ(pt->n_regs.ensure((size_t)T * 4) && pt->reg_off.ensure(((size_t)T + 1) * 8));
    TIMED(ws, K_OTHER, launch_scan(ws.stream, tv.n_regs, pt->reg_off.as<int64_t>(), T, ws.scan_tmp.as<int64_t>()));
    DevCounters hc;
    PE_OK(hipMemcpyAsync(&pt->n_regs_total, pt->reg_off.as<int64_t>() + T, 8, hipMemcpyDeviceToHost, ws.stream));
    PE_OK(hipMemcpyAsync(&err, tv.err, 4, hipMemcpyDeviceToHost, ws.stream));
    PE_OK(hipMemcpyAsync(&hc, tv.cnt, sizeof hc, hipMemcpyDeviceToHost, ws.stream));
    PE_OK(hipStreamSynchronize(ws.stream));
    if (err) { fprintf(stderr, "[bwamem_hip] device error flags %d in the paired-end phase 1\n", err); return false; }
    PE_REQ(pt->regs.ensure((size_t)(pt->n_regs_total + 1) * sizeof(AlnReg)));
    PE_OK(hipMemcpyAsync(pt->n_regs.p, tv.n_regs, (size_t)T * 4, hipMemcpyDeviceToDevice, ws.stream));
    TIMED(ws, K_OTHER, launch_pe_copy_regs(ws.stream, tv, tv.regs, tv.seed_off, pt->regs.as<AlnReg>(), pt->reg_off.as<int64_t>(), tv.n_regs));
    if (want_cand) {
        const int np = T >> 1;
        PE_REQ(ws.pe_dir.ensure((size_t)np + 8) && ws.pe_is.ensure((size_t)np * 8 + 8));
        TIMED(ws, K_OTHER, launch_pestat_cand(ws.stream, ix->d, opt, tv, ws.pe_dir.as<int8_t>(), ws.pe_is.as<int64_t>()));
        pt->cand_dir.resize(np); pt->cand_is.resize(np);
        if (np) {
            PE_OK(hipMemcpyAsync(pt->cand_dir.data(), ws.pe_dir.p, (size_t)np, hipMemcpyDeviceToHost, ws.stream));
            PE_OK(hipMemcpyAsync(pt->cand_is.data(), ws.pe_is.p, (size_t)np * 8, hipMemcpyDeviceToHost, ws.stream));
        }
    }
    PE_OK(hipStreamSynchronize(ws.stream));
    {
        std::lock_guard<std::mutex> lk(g_stats.mu);
        g_stats.s.n_reads += T; g_stats.s.n_lf += hc.n_lf; g_stats.s.n_sa += hc.n_sa; g_stats.s.n_dp_cells += hc.n_dp_cells; ++g_stats.s.n_tiles;
    }
    timed_collect(ws);
    return true;
}

static bool pe_phase2_tile(bwaidx_s* ix, Workspace& ws, const MemOpt& opt, bwamem_batch_s* b, int64_t read_id0, size_t tile_index, PeTile* pt, const MemPestat* pes, TileOut& to)
{
    RoctxRange rr("bwamem_hip:tile_pe_phase2");
    const int T = pt->T, L = pt->L;
    int attempts = 0, cap_u = 256, pe_job_cap = 0, pe_rescue_cap = 0;
    ix->hints.get_pe(T, pe_rescue_cap, pe_job_cap, cap_u);
    size_t pe_zpool = 0;
    for (;;) {
        if (++attempts > 8) { fprintf(stderr, "[bwamem_hip] paired-end tile could not be sized\n"); return false; }
        PE_REQ(ws.ensure_reads(opt, T, L, 0, ws.out_cap_hint, post_bytes_per_read(L, opt, false), false));
        TileView tv = ws.view();
        tv.n_reads = T; tv.max_len = L; tv.read_id0 = read_id0 + pt->r0;
        tv.seq = pt->seq; tv.seq_off = pt->seq_off;
        PE_OK(hipMemsetAsync(ws.err.p, 0, 64, ws.stream));
        PE_OK(hipMemcpyAsync(tv.n_regs, pt->n_regs.p, (size_t)T * 4, hipMemcpyDeviceToDevice, ws.stream));
        PE_REQ(ws.pe_caps.ensure((size_t)T * 4) && ws.pe_reg_off2.ensure(((size_t)T + 1) * 8));
        TIMED(ws, K_OTHER, launch_pe_caps(ws.stream, opt, tv, ws.pe_caps.as<int32_t>()));
        TIMED(ws, K_OTHER, launch_scan(ws.stream, ws.pe_caps.as<int32_t>(), ws.pe_reg_off2.as<int64_t>(), T, ws.scan_tmp.as<int64_t>()));
        int64_t tot = 0;
        PE_OK(hipMemcpyAsync(&tot, ws.pe_reg_off2.as<int64_t>() + T, 8, hipMemcpyDeviceToHost, ws.stream));
        PE_OK(hipStreamSynchronize(ws.stream));
        int span = 0;
        for (int d = 0; d < 4; ++d) if (!pes[d].failed) span = std::max(span, pes[d].high - pes[d].low);
        const int cap_h = L + 32, cap_b = (span + 2 * L) / 2 + 16;
        const int64_t per_pair = (((int64_t)16 * cap_h + (int64_t)8 * cap_b + (int64_t)2 * opt.max_matesw * sizeof(AlnReg) + (int64_t)16 * cap_u) + 63) & ~(int64_t)63;
        PE_REQ(ws.pe_regs2.ensure((size_t)(tot + 1) * sizeof(AlnReg)) && ws.pe_ints2.ensure((size_t)(tot + 1) * 8) && ws.pe_vpool.ensure((size_t)(tot + 2) * 16 + 64 + (size_t)(tot + 2) * 24)
               && ws.pe_scratch.ensure((size_t)((T >> 1) + 1) * (size_t)per_pair));
        AlnReg* regs2 = ws.pe_regs2.as<AlnReg>(); int64_t* reg_off2 = ws.pe_reg_off2.as<int64_t>();
        TIMED(ws, K_OTHER, launch_pe_copy_regs(ws.stream, tv, pt->regs.as<AlnReg>(), pt->reg_off.as<int64_t>(), regs2, reg_off2, tv.n_regs));
        // pairing decisions and the list of regions whose CIGAR needs DP; the DP jobs; the records
        const int jc = std::max(pe_job_cap, std::max(1024, T / 2));
        PE_REQ(ws.ensure_jobs(jc, 4 * L + 16, std::max(pe_zpool, (size_t)jc * (size_t)(2 * L + 64) * 20 + ((size_t)64 << 20))));
        PE_REQ(ws.pe_states.ensure(pe_state_bytes(T)));
        {   // the job buffers may have moved
            TileView t2 = ws.view();
            t2.n_reads = T; t2.max_len = L; t2.read_id0 = tv.read_id0; t2.seq = tv.seq; t2.seq_off = tv.seq_off;
            tv = t2;
        }
        PE_OK(hipMemsetAsync(ws.job_cnt.p, 0, 64, ws.stream));
        if (tv.debug & 0x2000) PE_OK(hipMemsetAsync(ws.cnt.p, 0, sizeof(DevCounters), ws.stream));
        {   // rescue alignments: a few per cent of the pairs ask for one, pairs in repeats for many
            static const int floor0 = []{ const char* e = getenv("BWAMEM_HIP_PE_RESCUE_CAP0"); return e && atoi(e) > 0 ? atoi(e) : 4096; }();   // test knob: start small, take the resize path
            const int rc = std::max(pe_rescue_cap, floor0 < 4096 ? floor0 : std::max(4096, T / 8));
            PE_REQ(ws.pe_rescue[0].ensure(pe_rescue_bytes(0, rc)) && ws.pe_rescue[1].ensure(pe_rescue_bytes(1, rc)) && ws.pe_rescue[2].ensure((size_t)(T / 2 + 1) * 12 + 64));   // header, first / count of each pair's rescue jobs, list of the heavy pairs
            pe_rescue_cap = rc;
        }
        TIMED(ws, K_FINAL, launch_pe_pair(ws.stream, ix->d, opt, tv, regs2, reg_off2, tv.n_regs, ws.pe_ints2.as<int32_t>(), ws.pe_vpool.p, (char*)ws.pe_vpool.p + (((size_t)(tot + 2) * 16 + 63) & ~(size_t)63),
                                          ws.pe_scratch.as<uint8_t>(), per_pair, cap_h, cap_b, cap_u, pes, ix->pair_tab, ws.pe_states.p,
                                          ws.pe_rescue[0].p, ws.pe_rescue[1].p, ws.pe_rescue[2].as<int32_t>() + 16, ws.pe_rescue[2].as<int32_t>() + 16 + (T / 2 + 1),
                                          ws.pe_rescue[2].as<int32_t>(), pe_rescue_cap));
        int32_t n_jobs = 0, err = 0, n_rescue = 0;
        PE_OK(hipMemcpyAsync(&n_jobs, tv.job_cnt, 4, hipMemcpyDeviceToHost, ws.stream));
        PE_OK(hipMemcpyAsync(&err, tv.err, 4, hipMemcpyDeviceToHost, ws.stream));
        PE_OK(hipMemcpyAsync(&n_rescue, ws.pe_rescue[2].p, 4, hipMemcpyDeviceToHost, ws.stream));
        PE_OK(hipStreamSynchronize(ws.stream));
        if (tv.debug & 0x2000) {
            DevCounters hc;
            PE_OK(hipMemcpy(&hc, tv.cnt, sizeof hc, hipMemcpyDeviceToHost));
            fprintf(stderr, "[bwamem_hip] pairing stage, Mclk summed over waves: rescue %.1f mark_primary %.1f pair %.1f jobs %.1f | plan %.1f | rescue jobs %d\n",
                    hc.dbg[0] / 1e6, hc.dbg[1] / 1e6, hc.dbg[2] / 1e6, hc.dbg[3] / 1e6, hc.dbg[4] / 1e6, n_rescue);
        }
        if (n_rescue > pe_rescue_cap) { pe_rescue_cap = n_rescue + n_rescue / 4; continue; }     // (the kernels after the plan saw ERR_RESCUE_CAP and did nothing)
        ix->hints.learn_pe(T, n_rescue, n_jobs, cap_u);
        if ((err & ERR_JOB_CAP) || n_jobs > ws.job_cap) { pe_job_cap = std::max(n_jobs + n_jobs / 4, ws.job_cap * 2); continue; }
        if (err & ERR_SCRATCH) { cap_u *= 8; if (attempts < 4) continue; }
        if (err) { fprintf(stderr, "[bwamem_hip] device error flags %d in the paired-end pairing stage\n", err); return false; }
        {
            TileView tvj = tv;                              // the job kernels address a region as regs[seed_off[read] + index]
            tvj.regs = regs2; tvj.seed_off = reg_off2;
            ws.zslab_bytes = gcigar_slab_bytes(opt, L);
            PE_REQ(!ws.zslab_bytes || ws.zslabs.ensure(ws.zslab_bytes * (size_t)gcigar_slab_grid(ix->d, 1 << 30)));
            TIMED(ws, K_FINAL, launch_gcigar(ws.stream, ix->d, opt, tvj, n_jobs, ws.jobs.p, ws.job_out.p, ws.job_cig.as<uint32_t>(), ws.job_cig_cap,
                                             ws.zpool.as<uint8_t>(), (unsigned long long)ws.zpool_cap, (unsigned long long*)(ws.job_cnt.as<int32_t>() + 2),
                                             ws.zslabs.as<uint8_t>(), ws.zslab_bytes, ws.job_cnt.as<int32_t>() + 4));
        }
        TIMED(ws, K_FINAL, launch_pe_out(ws.stream, ix->d, opt, tv, regs2, reg_off2, tv.n_regs, ws.pe_ints2.as<int32_t>(), pes, ws.pe_states.p,
                                         ws.job_out.p, ws.job_cig.as<uint32_t>(), ws.job_cig_cap));
        TIMED(ws, K_OTHER, launch_scan(ws.stream, tv.out_len, tv.out_off, T, ws.scan_tmp.as<int64_t>()));
        int64_t out_total = 0;
        PE_OK(hipMemcpyAsync(&out_total, tv.out_off + T, 8, hipMemcpyDeviceToHost, ws.stream));
        PE_OK(hipMemcpyAsync(&err, tv.err, 4, hipMemcpyDeviceToHost, ws.stream));
        PE_OK(hipStreamSynchronize(ws.stream));
        if (err & ERR_OUT_CAP) { ws.out_cap_hint *= 4; continue; }
        if (err & ERR_ZPOOL) { pe_zpool = std::max(pe_zpool * 4, ws.zpool_cap * 4); continue; }
        if (err) { fprintf(stderr, "[bwamem_hip] device error flags %d in the paired-end phase 2\n", err); return false; }
        PE_REQ(emit_tile(ws, b, tile_index, tv, out_total, to));
        timed_collect(ws);
        return true;
    }
}
#undef PE_OK
#undef PE_REQ

// what phase 1 of a paired-end call leaves behind for phase 2
struct PeCall { std::deque<PeTile> tiles; int64_t read_id0 = 0; };
static void pe_call_free(bwamem_batch_s* b)
{
    if (!b->pe) return;
    for (PeTile& t : b->pe->tiles) { t.n_regs.release(); t.regs.release(); t.reg_off.release(); }
    delete b->pe; b->pe = nullptr;
}

// phase 1 over all tiles (seeding .. regions per read).  The insert-size statistics are a property of the whole call,
// so when they have to be inferred every tile must finish phase 1 before any can start phase 2; when the caller
// supplies them (pes0) a tile goes straight on to phase 2 on the same worker and the call is a single pass
static bool pe_begin(bwaidx_s* ix, const MemOpt& opt, const MemPestat* pes0, bwamem_batch_s* b, int64_t read_id0)
{
    pe_call_free(b);
    if (pes0 && !build_pair_tab(ix, pes0)) return false;
    PeCall* pc = b->pe = new PeCall();
    pc->read_id0 = read_id0;
    CallPipe pp;
    pp.on_new_tiles = [&](size_t n) { pc->tiles.resize(pc->tiles.size() + n); b->tiles.resize(b->tiles.size() + n); b->sink.add_tiles(n); };
    if (!run_pipeline(ix, opt, b, true, pp, [&](Workspace& w, size_t i, const TileSpec& spec, const SeedStore& store, uint32_t chunk_r0) {
            if (!read_length_ok(ix, spec.L)) return false;
            PeTile* pt; TileOut* to;
            { std::lock_guard<std::mutex> lk(pp.mu); pt = &pc->tiles[i]; to = &b->tiles[i]; }     // (the deques grow while tiles run; their elements stay put)
            if (!pe_phase1_tile(ix, w, opt, b, read_id0, spec, pt, store, chunk_r0, pes0 == nullptr)) return false;
            if (!pes0) return true;
            const bool ok = pe_phase2_tile(ix, w, opt, b, read_id0, i, pt, pes0, *to);
            pt->n_regs.release(); pt->regs.release(); pt->reg_off.release();
            return ok;
        })) { pe_call_free(b); return false; }
    if (pes0) {
        for (const TileOut& t : b->tiles) b->result_bytes += t.bytes;
        pe_call_free(b);
    }
    return true;
}

// phase 2 (mate rescue, pairing, records) with the statistics of the whole call: tiles in flight on the worker streams
static bool pe_finish(bwaidx_s* ix, const MemOpt& opt, const MemPestat* pes, bwamem_batch_s* b)
{
    if (!b->pe) return false;
    if (!build_pair_tab(ix, pes)) { pe_call_free(b); return false; }
    std::deque<PeTile>& tiles = b->pe->tiles;
    const int64_t read_id0 = b->pe->read_id0;
    const char* env_s = getenv("BWAMEM_HIP_STREAMS");
    const int n_workers = std::max(1, std::min<int>((int)tiles.size(), env_s ? atoi(env_s) : 4));
    while ((int)ix->extra_ws.size() < n_workers - 1) ix->extra_ws.push_back(new Workspace());
    std::atomic<size_t> next(0);
    std::atomic<bool> failed(false);
    auto worker = [&](int k) {
        try {
            Workspace& w = k == 0 ? ix->ws : *ix->extra_ws[k - 1];
            w.dev_lds = ix->d.lds_bytes;
            if (hipSetDevice(ix->device) != hipSuccess || (!w.stream && hipStreamCreate(&w.stream) != hipSuccess)) { failed = true; b->sink.abort(); return; }
            while (!failed) {
                const size_t i = next++;
                if (i >= tiles.size()) break;
                if (!pe_phase2_tile(ix, w, opt, b, read_id0, i, &tiles[i], pes, b->tiles[i])) { failed = true; b->sink.abort(); }
            }
        } catch (const std::exception& ex) { fprintf(stderr, "[bwamem_hip] %s\n", ex.what()); failed = true; b->sink.abort(); }
    };
    std::vector<std::thread> th;
    for (int k = 1; k < n_workers; ++k) th.emplace_back(worker, k);
    worker(0);
    for (std::thread& t : th) t.join();
    if (!failed) { for (const TileOut& t : b->tiles) b->result_bytes += t.bytes; b->sink.last_total = b->result_bytes; }
    pe_call_free(b);
    return !failed;
}

// the (orientation, insert size) candidates of the pairs of this call / shard, in pair order
static void pe_candidates(const bwamem_batch_s* b, std::vector<int8_t>& dir, std::vector<int64_t>& is)
{
    dir.clear(); is.clear();
    if (!b->pe) return;
    for (const PeTile& t : b->pe->tiles) { dir.insert(dir.end(), t.cand_dir.begin(), t.cand_dir.end()); is.insert(is.end(), t.cand_is.begin(), t.cand_is.end()); }
}

static bool align_batch_pe(bwaidx_s* ix, const MemOpt& opt, const MemPestat* pes0, bwamem_batch_s* b, int64_t read_id0)
{
    if (!pe_begin(ix, opt, pes0, b, read_id0)) return false;
    if (pes0) return true;
    MemPestat pes[4];
    {
        RoctxRange rr("bwamem_hip:pestat");
        std::vector<int8_t> cand_dir; std::vector<int64_t> cand_is;
        pe_candidates(b, cand_dir, cand_is);
        host_pestat(opt, cand_dir, cand_is, pes);
    }
    return pe_finish(ix, opt, pes, b);
}

// Tiles are independent, and every kernel of a tile ends in a tail of a few long-running reads; several tiles are
// therefore kept in flight on separate HIP streams (one host thread + workspace each) so that one tile's tail
// overlaps the next tile's bulk.
// pe_step = 1: only phase 1 of a paired-end call (bwamem_hip_batch_pe_begin)
static void release_tile_outputs(bwamem_batch_s* b)
{
    for (TileOut& t : b->tiles) if (t.d && t.owned) (void)hipFree(t.d);
    b->tiles.clear(); b->result_bytes = 0;
}

static bool align_batch(bwaidx_s* ix, const MemOpt& opt, const MemPestat* pes, bwamem_batch_s* b, int64_t read_id0, int pe_step = 0)
{
    HIP_OK(hipSetDevice(ix->device));
    Workspace& ws = ix->ws;
    if (!ws.stream) HIP_OK(hipStreamCreate(&ws.stream));
    release_tile_outputs(b);
    pe_call_free(b);
    b->sink.begin_call(b->n_reads);
    if (b->n_reads == 0) return true;
    if (!b->h_payload) {                                   // resident batch: fresh working copy of the bases
        HIP_OK(hipMemcpyAsync(b->d_seq.p, b->d_raw.p, b->n_bytes, hipMemcpyDeviceToDevice, ws.stream));
        TIMED(ws, K_ENCODE, launch_encode(ws.stream, b->d_seq.as<uint8_t>(), (int64_t)b->n_bytes));
        HIP_OK(hipStreamSynchronize(ws.stream));
        timed_collect(ws);
        // the response of the previous call on this batch sizes the device slab of this one (no allocation per tile)
        if (!b->sink.to_host && b->sink.last_total && !b->sink.pool.ensure(b->sink.last_total + b->sink.last_total / 16 + 4096)) return false;
    } else b->n_bytes = 0;
    bool ok;
    if (pe_step == 1) ok = pe_begin(ix, opt, nullptr, b, read_id0);
    else if (opt.flag & MEM_F_PE) ok = align_batch_pe(ix, opt, pes, b, read_id0);
    else {
        CallPipe pp;
        pp.on_new_tiles = [&](size_t n) { b->tiles.resize(b->tiles.size() + n); b->sink.add_tiles(n); };
        ok = run_pipeline(ix, opt, b, false, pp, [&](Workspace& w, size_t i, const TileSpec& spec, const SeedStore& store, uint32_t chunk_r0) {
            if (!read_length_ok(ix, spec.L)) return false;
            TileOut* to;
            { std::lock_guard<std::mutex> lk(pp.mu); to = &b->tiles[i]; }
            return run_tile_se(ix, w, opt, b, read_id0, i, spec, *to, store, chunk_r0, w.out_cap_hint);
        });
        if (ok) for (const TileOut& t : b->tiles) b->result_bytes += t.bytes;
    }
    if (ok && pe_step != 1) b->sink.last_total = b->result_bytes;
    return ok;
}

// ------------------------------------------------------------------------------------------ C ABI
// No C++ exception may unwind through a JNI / ctypes frame (it would take the JVM down): the exported functions report
// failure the way the reference's do (NULL / non-zero), whatever went wrong inside (std::bad_alloc from a request-sized
// vector, std::length_error from an absurd nSeqs, ...).
// The call pipeline keeps up to six streams busy (tiles, the seeding chunk ahead, copies); the HIP runtime maps a process's streams
// onto four hardware queues unless told otherwise, and streams sharing a queue run one after the other.  Ask for eight when the
// process has not chosen a number itself.  Read by the runtime when it initialises, i.e. at the first HIP call: in a JVM that is
// this library's; in a process that already uses HIP the caller sets it (bench.py does).  Human-like genome: +5-9 %.
__attribute__((constructor)) static void hip_runtime_defaults() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

template <typename R, typename F> static R guarded(const char* what, R fail, F&& f) noexcept
{
    try { return f(); }
    catch (const std::exception& ex) { fprintf(stderr, "[bwamem_hip] %s: %s\n", what, ex.what()); }
    catch (...) { fprintf(stderr, "[bwamem_hip] %s: unknown exception\n", what); }
    return fail;
}

const std::vector<ContigInfo>& bwamem_index_contigs(const bwaidx_t* idx) { return idx->h.contigs; }     // (sam_writer.cpp)

extern "C" {

int bwamem_hip_set_device(int device) { g_device = device; g_device_explicit = true; return hipSetDevice(device) == hipSuccess ? 0 : -1; }
int bwamem_hip_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }

void bwamem_hip_stats_enable(int on) { g_stats.enabled = on != 0; }
void bwamem_hip_stats_reset(void) { std::lock_guard<std::mutex> lk(g_stats.mu); memset(&g_stats.s, 0, sizeof g_stats.s); }
void bwamem_hip_stats_get(bwamem_stats_t* out) { std::lock_guard<std::mutex> lk(g_stats.mu); *out = g_stats.s; }

void jnibwa_free(void* p) { free(p); }

const char* jnibwa_getVersion(void) { return "bwamem-hip-gfx950 (behavioural target: lh3/bwa cb950614ce7217788780b9a8d445c64cd4d8f62e)"; }

mem_opt_t* jnibwa_createDefaultOptions(void)
{   // upstream mem_opt_init + bwa_fill_scmat (defaults: SURVEY.md App. A.5)
    MemOpt* o = (MemOpt*)calloc(1, sizeof(MemOpt));
    o->a = 1; o->b = 4; o->o_del = o->o_ins = 6; o->e_del = o->e_ins = 1;
    o->w = 100; o->T = 30; o->zdrop = 100; o->pen_unpaired = 17; o->pen_clip5 = o->pen_clip3 = 5;
    o->max_mem_intv = 20; o->min_seed_len = 19; o->split_width = 10; o->max_occ = 500; o->max_chain_gap = 10000;
    o->max_ins = 10000; o->mask_level = 0.50f; o->drop_ratio = 0.50f; o->XA_drop_ratio = 0.80f; o->split_factor = 1.5f;
    o->chunk_size = 10000000; o->n_threads = 1; o->max_XA_hits = 5; o->max_XA_hits_alt = 200; o->max_matesw = 50;
    o->mask_level_redun = 0.95f; o->min_chain_weight = 0; o->max_chain_extend = 1 << 30;
    o->mapQ_coef_len = 50; o->mapQ_coef_fac = (int)log((double)o->mapQ_coef_len);
    int k = 0;
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) o->mat[k++] = (int8_t)(i == j ? o->a : -o->b); o->mat[k++] = -1; }
    for (int j = 0; j < 5; ++j) o->mat[k++] = -1;
    return (mem_opt_t*)o;
}

int jnibwa_createReferenceIndex(const char* refFileName, const char* indexPrefix, const char* algoName)
{
    return guarded("jnibwa_createReferenceIndex", 1, [&]() -> int {
        if (algoName && strcmp(algoName, "auto") && strcmp(algoName, "is") && strcmp(algoName, "rb2")) return -1;
        std::string err;
        if (!build_index_files(refFileName, indexPrefix, &err)) { fprintf(stderr, "[bwamem_hip] index build failed: %s\n", err.c_str()); return 1; }
        return 0;
    });
}

int jnibwa_createIndexFile(const char* refName, const char* imgName)
{
    return guarded("jnibwa_createIndexFile", 2, [&]() -> int {
        std::string err;
        std::vector<uint8_t> img = image_from_index_files(refName, &err);
        if (img.empty()) { printf("Failed to load index %s: %s\n", refName, err.c_str()); return 2; }
        int fd = open(imgName, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd == -1) { printf("Failed to open %s for writing: %s\n", imgName, strerror(errno)); return 2; }
        size_t len = img.size();
        const uint8_t* buf = img.data();
        while (len) {
            size_t to_write = std::min<size_t>(len, (size_t)1 << 30);
            if (write(fd, buf, to_write) != (ssize_t)to_write) { printf("Failed to write %s: %s\n", imgName, strerror(errno)); close(fd); return 2; }
            buf += to_write; len -= to_write;
        }
        if (close(fd) != 0) { printf("Failed to close %s: %s\n", imgName, strerror(errno)); return 2; }
        return 0;
    });
}

// The devices an index opened now goes to.  BWAMEM_HIP_DEVICES: "all" or a comma-separated list of device numbers (a number
// may be repeated: two replicas on one GPU, which is how the one-GPU test box exercises the multi-device path).  Unset: the
// device chosen with bwamem_hip_set_device if the process chose one (bench.py's ranks: one per GPU), else every visible
// device -- a JVM that loads this library on an 8-GPU node then uses all eight (BwaMemIndex.java:16-27: one shared index).
static std::vector<int> index_devices()
{
    std::vector<int> v;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) n = 0;
    const char* e = getenv("BWAMEM_HIP_DEVICES");
    if (e && *e && strcmp(e, "all")) {
        for (const char* p = e; *p; ) {
            char* end; long d = strtol(p, &end, 10);
            if (end == p) break;
            if (d >= 0 && d < n) v.push_back((int)d);
            p = *end == ',' ? end + 1 : end;
            if (*end && *end != ',') break;
        }
        if (v.empty()) v.push_back(g_device);
        return v;
    }
    if (g_device_explicit && !e) { v.push_back(g_device); return v; }
    for (int d = 0; d < n; ++d) v.push_back(d);
    if (v.empty()) v.push_back(g_device);
    return v;
}

static void destroy_replica(bwaidx_s* ix) { free_index(ix); delete ix; }

bwaidx_t* jnibwa_openIndex(int fd)
{
    return guarded("jnibwa_openIndex", (bwaidx_t*)0, [&]() -> bwaidx_t* {
        struct stat st;
        if (fstat(fd, &st) == -1) { close(fd); return 0; }
        void* mem = mmap(0, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
        close(fd);
        if (mem == MAP_FAILED) return 0;
        const std::vector<int> devs = index_devices();
        std::vector<bwaidx_s*> reps;
        std::vector<char> ok(devs.size(), 0);
        for (size_t k = 0; k < devs.size(); ++k) {
            bwaidx_s* ix = new bwaidx_s();
            ix->mem = (uint8_t*)mem; ix->l_mem = (size_t)st.st_size; ix->mmapped = k == 0; ix->device = devs[k];
            reps.push_back(ix);
        }
        auto load = [&](size_t k) {
            try { bwaidx_s* ix = reps[k]; ok[k] = parse_index_image(ix->mem, ix->l_mem, ix->h) && ix->h.seq_len < (1ull << 37) && upload_index(ix); }   // 37-bit ranks: packed SMEM candidates
            catch (...) { ok[k] = 0; }
        };
        std::vector<std::thread> th;
        for (size_t k = 1; k < reps.size(); ++k) th.emplace_back(load, k);      // every device loads from the one mapped image, side by side
        load(0);
        for (std::thread& t : th) t.join();
        bool all = true;
        for (char c : ok) all = all && c;
        if (!all) {
            fprintf(stderr, "[bwamem_hip] cannot open index image (malformed image or no usable HIP device)\n");
            for (bwaidx_s* ix : reps) destroy_replica(ix);
            munmap(mem, (size_t)st.st_size);
            return 0;
        }
        for (size_t k = 1; k < reps.size(); ++k) reps[0]->peers.push_back(reps[k]);
        return reps[0];
    });
}

int jnibwa_destroyIndex(bwaidx_t* pIdx)
{
    if (!pIdx) return 0;
    void* mem = pIdx->mem; size_t len = pIdx->l_mem;
    for (bwaidx_s* r : pIdx->peers) destroy_replica(r);
    pIdx->peers.clear();
    destroy_replica(pIdx);
    return munmap(mem, len);
}

int bwamem_hip_index_replicas(bwaidx_t* idx) { return idx ? 1 + (int)idx->peers.size() : 0; }

void* jnibwa_getRefContigNames(bwaidx_t* pIdx, size_t* pBufSize)
{
    return guarded("jnibwa_getRefContigNames", (void*)0, [&]() -> void* {
        const std::vector<ContigInfo>& c = pIdx->h.contigs;
        size_t bufSize = 4 + 4 * c.size();
        for (const ContigInfo& ci : c) bufSize += ci.name.size() + 1;   // the reference over-allocates by one byte per name (jnibwa.c:181)
        char* bufMem = (char*)calloc(bufSize, 1);
        *(int32_t*)bufMem = (int32_t)c.size();
        char* p = bufMem + 4;
        for (const ContigInfo& ci : c) {
            *(int32_t*)p = (int32_t)ci.name.size(); p += 4;
            memcpy(p, ci.name.data(), ci.name.size()); p += ci.name.size();
        }
        *pBufSize = bufSize;
        return bufMem;
    });
}

// Tooling (bench.py): an index image straight from base codes in host memory -- no FASTA, no five files in between -- built by
// the device half of the index builder (k_index.hip).  codes: l_pac bytes 0..3; contigs by name and length.  0 = ok.
int bwamem_hip_build_image(const uint8_t* codes, int64_t l_pac, int32_t n_contigs, const char* const* names, const int64_t* lens, const char* img_path)
{
    return guarded("bwamem_hip_build_image", -1, [&]() -> int {
        if (!codes || l_pac <= 0 || n_contigs <= 0 || !names || !lens || !img_path) return -1;
        IndexPieces p;
        int64_t off = 0;
        for (int i = 0; i < n_contigs; ++i) {
            ContigInfo c; c.offset = off; c.len = (int32_t)lens[i]; c.n_ambs = 0; c.gi = 0; c.is_alt = 0; c.name = names[i]; c.anno = "";
            p.contigs.push_back(c); off += lens[i];
        }
        if (off != l_pac) return -1;
        p.l_pac = l_pac; p.seed = 11;
        std::vector<uint8_t> fwd(codes, codes + l_pac);
        p.pac.assign((size_t)(l_pac / 4 + 1), 0);
        for (int64_t i = 0; i < l_pac; ++i) p.pac[i >> 2] |= (uint8_t)((fwd[i] & 3) << ((~i & 3) << 1));
        std::string err;
        if (!device_index_pieces(fwd, p, &err)) { fprintf(stderr, "[bwamem_hip] device index builder: %s\n", err.c_str()); return -1; }
        std::vector<uint8_t>().swap(fwd);
        const std::vector<uint8_t> img = image_from_pieces(p);
        FILE* fp = fopen(img_path, "wb");
        if (!fp) return -1;
        const bool ok = fwrite(img.data(), 1, img.size(), fp) == img.size();
        return fclose(fp) == 0 && ok ? 0 : -1;
    });
}

int bwamem_hip_index_contig_lengths(bwaidx_t* idx, int64_t* lens, int cap)
{
    if (!idx) return -1;
    const int n = (int)idx->h.contigs.size();
    if (lens) for (int i = 0; i < n && i < cap; ++i) lens[i] = idx->h.contigs[i].len;
    return n;
}

int bwamem_hip_index_unpack_pac(bwaidx_t* idx, int64_t start, int64_t n, void* d_dst)
{
    if (!idx || !d_dst || start < 0 || n < 0 || start + n > idx->d.l_pac) return -1;
    std::lock_guard<std::mutex> lk(idx->mu);
    if (hipSetDevice(idx->device) != hipSuccess) return -1;
    launch_unpack_pac(0, idx->d, start, n, (uint8_t*)d_dst);
    return hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}

// the request's strings (payload = pSeq + 4) with their offsets already known: device copies of both
static bwamem_batch_s* batch_from_offsets(bwaidx_t* idx, const char* payload, uint32_t n_reads, std::vector<int64_t>&& off)
{
    if (hipSetDevice(idx->device) != hipSuccess) return 0;
    bwamem_batch_s* b = new bwamem_batch_s();
    b->idx = idx; b->n_reads = n_reads;
    b->h_off = std::move(off);
    b->n_bytes = (size_t)b->h_off[n_reads];
    bool ok = b->d_raw.ensure(b->n_bytes + 64) && b->d_seq.ensure(b->n_bytes + 64) && b->d_off.ensure(((size_t)n_reads + 1) * 8);
    ok = ok && hipMemcpy(b->d_raw.p, payload, b->n_bytes, hipMemcpyHostToDevice) == hipSuccess
            && hipMemcpy(b->d_off.p, b->h_off.data(), ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { fprintf(stderr, "[bwamem_hip] request upload failed\n"); bwamem_hip_batch_free(b); return 0; }
    return b;
}

bwamem_batch_t* bwamem_hip_batch_upload(bwaidx_t* idx, const char* pSeq, size_t nBytes)
{
    return guarded("bwamem_hip_batch_upload", (bwamem_batch_t*)0, [&]() -> bwamem_batch_t* {
        if (!idx || !pSeq || nBytes < 4) return 0;
        uint32_t n_reads; memcpy(&n_reads, pSeq, 4);
        const char* p = pSeq + 4; const char* end = pSeq + nBytes;
        std::vector<int64_t> off((size_t)n_reads + 1);
        for (uint32_t i = 0; i < n_reads; ++i) {              // jnibwa.c:204-212 (strlen walk)
            off[i] = p - (pSeq + 4);
            const char* z = (const char*)memchr(p, 0, (size_t)(end - p));
            if (!z) { fprintf(stderr, "[bwamem_hip] request buffer ends inside read %u\n", i); return 0; }
            p = z + 1;
        }
        off[n_reads] = p - (pSeq + 4);
        return batch_from_offsets(idx, pSeq + 4, n_reads, std::move(off));
    });
}

bwamem_batch_t* bwamem_hip_batch_wrap_device(bwaidx_t* idx, const void* d_payload, size_t nBytes, uint32_t nReads, const int64_t* h_offsets)
{
    return guarded("bwamem_hip_batch_wrap_device", (bwamem_batch_t*)0, [&]() -> bwamem_batch_t* {
        if (!idx || !d_payload || !h_offsets) return 0;
        if (hipSetDevice(idx->device) != hipSuccess) return 0;
        bwamem_batch_s* b = new bwamem_batch_s();
        b->idx = idx; b->n_reads = nReads; b->n_bytes = nBytes;
        b->h_off.assign(h_offsets, h_offsets + (size_t)nReads + 1);
        bool ok = b->h_off[nReads] == (int64_t)nBytes
            && b->d_raw.ensure(nBytes + 64) && b->d_seq.ensure(nBytes + 64) && b->d_off.ensure(((size_t)nReads + 1) * 8)
            && hipMemcpy(b->d_raw.p, d_payload, nBytes, hipMemcpyDeviceToDevice) == hipSuccess
            && hipMemcpy(b->d_off.p, b->h_off.data(), ((size_t)nReads + 1) * 8, hipMemcpyHostToDevice) == hipSuccess;
        if (!ok) { fprintf(stderr, "[bwamem_hip] batch_wrap_device failed\n"); bwamem_hip_batch_free(b); return 0; }
        return b;
    });
}

int bwamem_hip_batch_align(bwaidx_t* idx, const mem_opt_t* opt, const mem_pestat_t* pes, bwamem_batch_t* b, int64_t read_id0)
{
    return guarded("bwamem_hip_batch_align", -1, [&]() -> int {
        if (!idx || !opt || !b) return -1;
        std::lock_guard<std::mutex> lk(idx->mu);
        MemOpt o; memcpy(&o, opt, sizeof o);
        return align_batch(idx, o, (const MemPestat*)pes, b, read_id0) ? 0 : -1;
    });
}

int bwamem_hip_batch_pe_begin(bwaidx_t* idx, const mem_opt_t* opt, bwamem_batch_t* b, int64_t read_id0)
{
    return guarded("bwamem_hip_batch_pe_begin", -1, [&]() -> int {
        if (!idx || !opt || !b) return -1;
        std::lock_guard<std::mutex> lk(idx->mu);
        MemOpt o; memcpy(&o, opt, sizeof o);
        if (!(o.flag & MEM_F_PE)) return -1;
        return align_batch(idx, o, nullptr, b, read_id0, 1) ? 0 : -1;
    });
}

size_t bwamem_hip_batch_pe_candidates(const bwamem_batch_t* b, int8_t* dir, int64_t* isize)
{
    return guarded("bwamem_hip_batch_pe_candidates", (size_t)0, [&]() -> size_t {
        if (!b) return 0;
        std::vector<int8_t> d; std::vector<int64_t> is;
        pe_candidates(b, d, is);
        if (dir && isize && !d.empty()) { memcpy(dir, d.data(), d.size()); memcpy(isize, is.data(), is.size() * 8); }
        return d.size();
    });
}

void bwamem_hip_pestat(const mem_opt_t* opt, const int8_t* dir, const int64_t* isize, size_t n, mem_pestat_t* pes)
{
    (void)guarded("bwamem_hip_pestat", 0, [&]() -> int {
        MemOpt o; memcpy(&o, opt, sizeof o);
        std::vector<int8_t> d(dir, dir + n); std::vector<int64_t> is(isize, isize + n);
        host_pestat(o, d, is, (MemPestat*)pes);
        return 0;
    });
}

int bwamem_hip_batch_pe_finish(bwaidx_t* idx, const mem_opt_t* opt, const mem_pestat_t* pes, bwamem_batch_t* b)
{
    return guarded("bwamem_hip_batch_pe_finish", -1, [&]() -> int {
        if (!idx || !opt || !b || !pes) return -1;
        std::lock_guard<std::mutex> lk(idx->mu);
        if (hipSetDevice(idx->device) != hipSuccess) return -1;
        MemOpt o; memcpy(&o, opt, sizeof o);
        return pe_finish(idx, o, (const MemPestat*)pes, b) ? 0 : -1;
    });
}

size_t bwamem_hip_batch_result_bytes(const bwamem_batch_t* b) { return b->result_bytes; }

int bwamem_hip_batch_download(bwamem_batch_t* b, void* dst)
{
    return guarded("bwamem_hip_batch_download", -1, [&]() -> int {
        if (hipSetDevice(b->idx->device) != hipSuccess) return -1;
        uint8_t* p = (uint8_t*)dst;
        for (const TileOut& t : b->tiles) {
            if (t.bytes && hipMemcpy(p, t.d, t.bytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
            p += t.bytes;
        }
        return 0;
    });
}

void bwamem_hip_batch_free(bwamem_batch_t* b)
{
    if (!b) return;
    (void)hipSetDevice(b->idx->device);
    release_tile_outputs(b);
    pe_call_free(b);
    b->d_raw.release(); b->d_seq.release(); b->d_off.release(); b->sink.pool.release();
    delete b;
}

// One (sub)call on one replica: reads [.., ..+n) of the request starting at payload, numbered from read_id0 within the logical
// call.  pe_step as in align_batch (1: stop after phase 1 of a paired-end call).  The replica's lock is the caller's business.
static bwamem_batch_s* new_streamed_batch(bwaidx_s* ix, const char* payload, uint32_t n)
{
    bwamem_batch_s* b = new bwamem_batch_s();
    b->idx = ix; b->n_reads = n; b->h_payload = payload; b->sink.to_host = true;
    return b;
}

// A call cut across the replicas of the handle (one contiguous range of reads per device, pairs kept together, each shard
// numbered from its first read's index in the call: the primary-marking hash needs it).  The request
// carries no length, so a walker (this thread) finds where each shard starts -- shard k can start as soon as the walk has
// passed the reads before it -- and every shard then streams its own range as a call of its own.  A paired-end call with
// inferred insert-size statistics stops every shard after phase 1, reduces all shards' candidates on the host (upstream's
// mem_pestat is a reduction over the whole call) and finishes every shard with the same statistics.
static void* create_alignments_split(bwaidx_s* ix0, const MemOpt& o, const MemPestat* pes0, const char* payload, uint32_t n, size_t* out_bytes)
{
    std::vector<bwaidx_s*> reps; reps.push_back(ix0);
    for (bwaidx_s* r : ix0->peers) reps.push_back(r);
    const bool pe = (o.flag & MEM_F_PE) != 0, two_step = pe && !pes0;
    int D = (int)reps.size();
    if ((uint64_t)D > (uint64_t)n / (pe ? 2 : 1)) D = std::max<int>(1, (int)(n / (pe ? 2 : 1)));
    struct Shard { const char* p = nullptr; uint32_t n = 0; int64_t id0 = 0; bool ready = false, p1_done = false, ok = false; bwamem_batch_s* b = nullptr; void* res = nullptr; size_t bytes = 0; };
    std::vector<Shard> sh((size_t)D);
    uint32_t share = n / (uint32_t)D;
    if (pe) share &= ~1u;
    for (int k = 0; k < D; ++k) { sh[k].n = k + 1 < D ? share : n - share * (uint32_t)(D - 1); sh[k].id0 = (int64_t)share * k; }
    std::mutex mu; std::condition_variable cv;
    bool failed = false, stats_ready = false;
    MemPestat pes[4];
    std::lock_guard<std::mutex> split_lk(ix0->split_mu);
    auto run = [&](int k) {
        bool ok = false;
        try {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return sh[k].ready || failed; }); if (failed) { sh[k].p1_done = true; cv.notify_all(); return; } }
            bwaidx_s* ix = reps[k];
            std::lock_guard<std::mutex> ilk(ix->mu);
            sh[k].b = new_streamed_batch(ix, sh[k].p, sh[k].n);
            ok = align_batch(ix, o, pes0, sh[k].b, sh[k].id0, two_step ? 1 : 0);
            if (two_step) {
                { std::lock_guard<std::mutex> lk(mu); sh[k].p1_done = true; if (!ok) failed = true; }
                cv.notify_all();
                { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stats_ready || failed; }); ok = ok && !failed; }
                if (ok) ok = hipSetDevice(ix->device) == hipSuccess && pe_finish(ix, o, pes, sh[k].b);
                else pe_call_free(sh[k].b);
            }
            if (ok) { sh[k].bytes = sh[k].b->result_bytes; sh[k].res = sh[k].b->sink.take(sh[k].bytes); ok = sh[k].res != nullptr; }
        } catch (const std::exception& ex) { fprintf(stderr, "[bwamem_hip] createAlignments shard %d: %s\n", k, ex.what()); ok = false; }
          catch (...) { ok = false; }
        { std::lock_guard<std::mutex> lk(mu); sh[k].ok = ok; sh[k].p1_done = true; if (!ok) failed = true; }
        cv.notify_all();
    };
    std::vector<std::thread> th;
    for (int k = 0; k < D; ++k) th.emplace_back(run, k);
    {   // the walk: shard k starts where the reads before it end
        const char* p = payload;
        for (int k = 0; k < D; ++k) {
            { std::lock_guard<std::mutex> lk(mu); sh[k].p = p; sh[k].ready = true; }
            cv.notify_all();
            if (k + 1 < D) { uint64_t got = 0; p = scan_reads(p, sh[k].n, (size_t)-1, &got); }
        }
    }
    if (two_step) {
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { for (const Shard& s : sh) if (!s.p1_done) return false; return true; }); }
        bool ok; { std::lock_guard<std::mutex> lk(mu); ok = !failed; }
        if (ok) {
            std::vector<int8_t> dir, d1; std::vector<int64_t> is, i1;
            for (const Shard& s : sh) { pe_candidates(s.b, d1, i1); dir.insert(dir.end(), d1.begin(), d1.end()); is.insert(is.end(), i1.begin(), i1.end()); }
            host_pestat(o, dir, is, pes);
        }
        { std::lock_guard<std::mutex> lk(mu); stats_ready = true; }
        cv.notify_all();
    }
    for (std::thread& t : th) t.join();
    void* res = nullptr;
    size_t total = 0;
    if (!failed) {
        for (const Shard& s : sh) total += s.bytes;
        res = malloc(total ? total : 1);
        if (res) {
            std::vector<std::thread> cp;
            size_t off = 0;
            for (const Shard& s : sh) { uint8_t* dst = (uint8_t*)res + off; cp.emplace_back([dst, &s] { if (s.bytes) memcpy(dst, s.res, s.bytes); }); off += s.bytes; }
            for (std::thread& t : cp) t.join();
        }
    }
    for (Shard& s : sh) { if (s.res) free(s.res); if (s.b) bwamem_hip_batch_free(s.b); }
    if (res && out_bytes) *out_bytes = total;
    return res;
}

void* jnibwa_createAlignments(bwaidx_t* pIdx, mem_opt_t* pOpts, mem_pestat_t* peStats, char* pSeq, size_t* pBufSize)
{
    if (pBufSize) *pBufSize = 0;
    if (!pIdx || !pOpts || !pSeq) return 0;
    // The request carries no length; the reference walks its NUL-terminated strings first (jnibwa.c:204-212), aligns, then
    // copies the records out.  Here the three overlap: a producer thread walks the buffer and streams it to the device
    // stretch by stretch (read offsets are found there), tiles are aligned as soon as their stretch is resident, and every
    // finished tile's records go straight into the malloc'ed block that is returned.
    // With several replicas behind the handle (jnibwa_openIndex): a large call is cut across all of them; a small one goes to
    // the next replica in turn (the first free one from there), so that concurrent callers of one BwaMemIndex
    // (BwaMemIndex.java:16-27) spread over the devices instead of queueing on one.
    void* res = 0;
    RoctxRange rr("jnibwa_createAlignments");
    try {
        uint32_t n; memcpy(&n, pSeq, 4);
        MemOpt o; memcpy(&o, pOpts, sizeof o);
        const int D = 1 + (int)pIdx->peers.size();
        if (D > 1) {
            static const long split_min = []{ const char* e = getenv("BWAMEM_HIP_SPLIT_MIN"); return e && atol(e) > 0 ? atol(e) : 131072L; }();   // reads per replica below which cutting a call does not pay
            if ((long)n >= split_min * D && n >= 2u * (unsigned)D) {
                size_t bytes = 0;
                res = create_alignments_split(pIdx, o, (const MemPestat*)peStats, pSeq + 4, n, &bytes);
                if (res && pBufSize) *pBufSize = bytes;
                return res;
            }
        }
        bwaidx_s* ix = pIdx;
        std::unique_lock<std::mutex> lk;
        if (D > 1) {
            const uint32_t first = pIdx->next_replica.fetch_add(1);
            for (int i = 0; i < D && !lk.owns_lock(); ++i) {
                bwaidx_s* r = (first + i) % D == 0 ? pIdx : pIdx->peers[(first + i) % D - 1];
                std::unique_lock<std::mutex> t(r->mu, std::try_to_lock);
                if (t.owns_lock()) { lk = std::move(t); ix = r; }
            }
            if (!lk.owns_lock()) { ix = first % D == 0 ? pIdx : pIdx->peers[first % D - 1]; lk = std::unique_lock<std::mutex>(ix->mu); }
        } else lk = std::unique_lock<std::mutex>(ix->mu);
        bwamem_batch_s* b = new_streamed_batch(ix, pSeq + 4, n);
        if (align_batch(ix, o, (const MemPestat*)peStats, b, 0)) {
            res = b->sink.take(b->result_bytes);
            if (res && pBufSize) *pBufSize = b->result_bytes;
        }
        lk.unlock();
        bwamem_hip_batch_free(b);
    } catch (const std::exception& ex) { fprintf(stderr, "[bwamem_hip] createAlignments: %s\n", ex.what()); res = 0; }
      catch (...) { res = 0; }
    return res;
}

} // extern "C"
