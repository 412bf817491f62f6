// bwamem_types.h -- plain-data types shared by the host pipeline and the HIP kernels.
//
// Layout contracts kept byte-identical with the reference boundary:
//   MemOpt      = upstream mem_opt_t, 168 bytes, offsets pinned by
//                 BwaMemAligner.java:46-138
//   MemPestat   = upstream mem_pestat_t as filled by ...BwaMemIndex.c:21-40
// Everything else is this implementation's own device-side layout.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <hip/hip_runtime.h>

struct MemOpt {
    int a, b;
    int o_del, e_del;
    int o_ins, e_ins;
    int pen_unpaired;
    int pen_clip5, pen_clip3;
    int w;
    int zdrop;
    uint64_t max_mem_intv;
    int T;
    int flag;
    int min_seed_len;
    int min_chain_weight;
    int max_chain_extend;
    float split_factor;
    int split_width;
    int max_occ;
    int max_chain_gap;
    int n_threads;
    int chunk_size;
    float mask_level;
    float drop_ratio;
    float XA_drop_ratio;
    float mask_level_redun;
    float mapQ_coef_len;
    int mapQ_coef_fac;
    int max_ins;
    int max_matesw;
    int max_XA_hits, max_XA_hits_alt;
    int8_t mat[25];
};
static_assert(sizeof(MemOpt) == 168, "mem_opt_t ABI (BwaMemAligner.java:137)");

struct MemPestat { int low, high; int failed; double avg, std; };
static_assert(sizeof(MemPestat) == 32, "mem_pestat_t ABI");

// mem_pair's insert-size score term log(2 erfc(|dist - avg| / std / sqrt 2)) for every integer distance an orientation
// admits, computed by the host's glibc (pipeline.cpp: build_pair_tab): t[off[d] + dist - lo[d]] for dist in
// [lo[d], lo[d] + n[d]); a distance within [low, high] but outside that range has erfc == 0 exactly, i.e. the term is -inf
struct PairTab { const double* t; int64_t lo[4]; int32_t n[4], off[4]; };

enum {
    MEM_F_PE = 0x2, MEM_F_NOPAIRING = 0x4, MEM_F_ALL = 0x8, MEM_F_NO_MULTI = 0x10,
    MEM_F_NO_RESCUE = 0x20, MEM_F_PRIMARY5 = 0x800
};

// ---- FM-index + reference resident in HBM ----
// The image's occ/bwt array (one 64-byte line per 128 symbols with 4 x u64 counts) is
// re-blocked once at openIndex into 32-byte blocks of 64 symbols: 16 bytes of counts + 4 x u32 of 16 symbols
// each (MSB first).  The counts are the absolute numbers of C, G and T before the block, 40 bits each: three low words plus one word of
// high bytes, so that unpacking costs three bit-field extracts (A follows
// from the block position: 64*b - C - G - T).  A rank query then needs two 16-byte lane loads instead of four and
// popcounts at most 4 words instead of 8.  The seeding kernel is bound by per-lane vector-memory requests, so
// this halves its cost.  40-bit counts cover texts up to 2^40 symbols.
struct DevIndex {
    const uint4* occ;          // 2 x uint4 per 64-symbol block: {C, G, T low words, high bytes}, {sym[4]}
    const uint32_t* sa_lo;     // suffix array, 40 bits per entry (low word / high byte), kept for every sa_intv-th rank:
    const uint8_t*  sa_hi;     // densified from the image's sampling at load time (k_sa_densify); entry 0 = -1
    const uint8_t*  pac;       // 2 bit/base, first base in the two MSBs
    const int64_t*  ann_offset;
    const int32_t*  ann_len;
    const int32_t*  ann_is_alt;
    const int32_t*  ann_name_off;  // n_seqs+1 offsets into names
    const char*     names;
    const double*   log_tab;   // glibc log(i) for i in [0, log_tab_n): libm stays on the host
    uint64_t primary, L2[5], seq_len;
    int64_t  l_pac;
    int32_t  n_seqs, sa_intv, log_tab_n, sa_shift;   // sa_intv = 1 << sa_shift
    int32_t  n_cu, lds_bytes;                        // of the device the index lives on (launch geometry; host side only)
};

struct Intv { uint64_t x0, x1, size, info; };          // info = start<<32 | end

struct Seed { int64_t rbeg; int32_t qbeg, len, score, next; };  // next: chain linked list / scratch

struct Chain {
    int64_t pos;
    int32_t n, first, rid;
    uint32_t w;
    int32_t kept, is_alt;
    int32_t seed0;          // index (tile-global) of the chain's first seed in the chain-ordered seed array
    int32_t last;           // linked-list tail while chaining
    float frac_rep;
    int32_t pad_;
};

struct AlnReg {
    int64_t rb, re;
    int32_t qb, qe;
    int32_t rid;
    int32_t score, truesc, sub, alt_sc, csub, sub_n, w, seedcov, secondary, secondary_all, seedlen0;
    int32_t n_comp, is_alt;
    float frac_rep;
    int32_t pad_;
    uint64_t hash;
};

// one ksw_align2 of a read (or a stretch of it) against a reference window, run by the wave SW kernel (k_pe.hip)
struct SwJob { int64_t rb; int32_t read, tag, l_ms, is_rev, tlen, xtra, q_off, pad_; };
struct KswR { int score, te, qe, score2, te2, tb, qb; };

// per-kernel algorithmic counters; accumulated with one atomic per wave
struct DevCounters {
    unsigned long long n_ext, n_lf, n_sa, n_dp_cells, n_ref_bases, n_reads;
    unsigned long long dbg[10];      // BWAMEM_HIP_DEBUGK bit 0x2000: shader clocks per phase of the lane-per-pair kernels, summed over waves (printed per tile)
};

// per-tile error / overflow flags set by kernels, read back by the host after each stage
enum { ERR_INTV_CAP = 1, ERR_OUT_CAP = 2, ERR_CIGAR_CAP = 4, ERR_LONG_READ = 8, ERR_SCRATCH = 16, ERR_BTREE = 32, ERR_BAD_REG = 64, ERR_JOB_CAP = 128, ERR_ZPOOL = 256, ERR_RESCUE_CAP = 512 };

#define BT_NODE_INTS 40          // ints per node of the chaining B-tree (k_chain.hip); the pool doubles as per-read scratch later

struct TileView {
    // reads of this tile
    int32_t n_reads;
    int32_t max_len;              // longest read in the tile
    int64_t read_id0;             // index of the tile's first read within the call (hash tie-break)
    const int64_t* seq_off;       // n_reads+1 offsets into seq
    uint8_t* seq;                 // base codes 0..4
    // seeding
    int32_t intv_cap;             // per-read capacity of intv / scratch vectors
    Intv* intv;                   // [n_reads][intv_cap]
    int32_t* n_intv;              // [n_reads]
    Intv* smem_scratch;           // [n_reads/64][2][smem_cap][64 lanes] x 16-byte packed candidates (prev, curr), lane-interleaved
    int32_t smem_cap;
    int32_t* l_rep;               // [n_reads]
    int32_t* n_seeds;             // [n_reads] -> exclusive scan in seed_off
    int64_t* seed_off;            // [n_reads+1]
    int32_t* intv_seed_off;       // [n_reads][intv_cap] offset of each interval's first occurrence within the read
    // seeds / chains / regions (pools indexed by seed_off)
    Seed* seeds;                  // occurrence order
    int32_t* seed_rid;            // rid per occurrence (<0: dropped)
    Seed* cseeds;                 // chain-ordered
    Chain* chains;
    int32_t* n_chains;            // [n_reads]
    int32_t* bt_nodes;            // B-tree node pool
    uint64_t* srt;                // per-seed sort keys
    AlnReg* regs;
    int32_t* n_regs;              // [n_reads]
    // output staging
    int32_t out_cap;              // bytes per read
    uint8_t* out;                 // [n_reads][out_cap]
    int32_t* out_len;             // [n_reads]
    int64_t* out_off;             // [n_reads+1]
    // per-read scratch for the post stage (global DP etc.)
    uint8_t* post_scratch;
    int64_t post_scratch_per_read;
    // flags + counters
    int32_t* err;
    DevCounters* cnt;
    // global-alignment jobs (single-end): regions whose CIGAR needs DP, compacted for the wave-parallel kernel
    int32_t* job_cnt;             // [1]
    void* jobs;                   // DpJob[job_cap]
    int32_t job_cap;
    int32_t smem_groups;          // workgroups the smem_scratch spill area covers (upper bound for the k_seed grid)
    int32_t* dp_rows;             // null: DP rows of k_extend / k_gcigar in LDS; else dp_rows_blocks slices of 3 x (max_len + 2) ints (very long reads)
    int32_t dp_rows_blocks;
    int32_t debug;                // BWAMEM_HIP_DEBUGK: device-side progress prints (debugging aid)
    const int32_t* order;         // [n_reads] or null: reads by falling seed count.  k_extend (one wavefront per read) takes read order[block]: the
                                  // heaviest reads start first, so a tile's tail is no longer whichever heavy read happened to be dispatched last
    int32_t ext_hbm;              // k_extend keeps its rows in dp_rows (BWAMEM_HIP_DP_ROWS=hbm: tests)
    int32_t gcigar_hbm_only;      // k_gcigar: every wave-form job through the rows in dp_rows (same switch)
    int32_t pad_;
};
