/*
 * bwa_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the BWA-MEM algorithm that the reference
 * (broadinstitute/gatk-bwamem-jni) reaches through the single call site
 * src/main/c/jnibwa.c:214 (mem_process_seqs).  The arithmetic lives in the
 * third-party module github.com/lh3/bwa pinned at commit
 * cb950614ce7217788780b9a8d445c64cd4d8f62e (src/main/c/Makefile:17,26), which
 * is NOT vendored under /root/reference and is absent from this build
 * environment.  The algorithm is therefore restated from its published
 * definition; parity is anchored on the reference's own call sites, on the
 * seven known-answer alignments of BwaMemIndexTest.java:45-127 and on the
 * byte-exact index fixtures in src/test/resources/.
 *
 * PARITY STATUS: pinned for POS / CIGAR / NM / FLAG / mate POS / TLEN on the
 * reference's 7 test reads and for the on-disk index format.  MAPQ, AS, XS,
 * MD, XA, supplementary/secondary output are "parity unpinned" (no reference
 * test or golden vector asserts them).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (gatk-bwamem-jni_amd/) never links it.
 */
#ifndef BWA_ORACLE_H_
#define BWA_ORACLE_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t bwtint_t;

/* ---- index objects (layouts of the .img dump: SURVEY.md App. A.4) ---- */
typedef struct {
	bwtint_t primary, L2[5], seq_len, bwt_size;
	const uint32_t *bwt;     /* interleaved occ/bwt blocks, App. A.2 */
	int sa_intv;
	bwtint_t n_sa;
	const bwtint_t *sa;      /* sa[0] == (u64)-1 */
} o_bwt_t;

typedef struct {
	int64_t offset;
	int32_t len, n_ambs;
	uint32_t gi;
	int32_t is_alt;
	const char *name, *anno;
} o_ann_t;

typedef struct { int64_t offset; int32_t len; char amb; } o_amb_t;

typedef struct {
	int64_t l_pac;
	int32_t n_seqs;
	uint32_t seed;
	o_ann_t *anns;
	int32_t n_holes;
	const o_amb_t *ambs;
} o_bns_t;

typedef struct {
	o_bwt_t bwt;
	o_bns_t bns;
	const uint8_t *pac;
	uint8_t *mem;
	size_t l_mem;
	int is_mmap;
} o_idx_t;

/* ---- options: 168 bytes, offsets pinned by BwaMemAligner.java:46-138 ---- */
typedef struct {
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
} o_opt_t;

#define O_F_PE        0x2
#define O_F_NOPAIRING 0x4
#define O_F_ALL       0x8
#define O_F_NO_MULTI  0x10
#define O_F_NO_RESCUE 0x20
#define O_F_PRIMARY5  0x800

/* jobject_to_mem_pestat_t target: ...BwaMemIndex.c:21-40 */
typedef struct { int low, high; int failed; double avg, std; } o_pestat_t;

/* ---- FM-index primitives (upstream bwt.c) ---- */
typedef struct { bwtint_t x[3], info; } o_intv_t;
typedef struct { size_t n, m; o_intv_t *a; } o_intv_v;

void o_bwt_occ4(const o_bwt_t *bwt, bwtint_t k, bwtint_t cnt[4]);
void o_bwt_extend(const o_bwt_t *bwt, const o_intv_t *ik, o_intv_t ok[4], int is_back);
bwtint_t o_bwt_sa(const o_bwt_t *bwt, bwtint_t k);
int o_bwt_smem1(const o_bwt_t *bwt, int len, const uint8_t *q, int x, int min_intv, o_intv_v *mem, o_intv_v tmpvec[2]);
int o_bwt_seed_strategy1(const o_bwt_t *bwt, int len, const uint8_t *q, int x, int min_len, int max_intv, o_intv_t *mem);

/* ---- DP kernels (upstream ksw.c) ---- */
int o_ksw_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                  int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0,
                  int *qle, int *tle, int *gtle, int *gscore, int *max_off);
int o_ksw_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                  int o_del, int e_del, int o_ins, int e_ins, int w, int *n_cigar, uint32_t **cigar);
typedef struct { int score, te, qe, score2, te2, tb, qb; } o_kswr_t;
#define O_KSW_XBYTE  0x10000
#define O_KSW_XSTOP  0x20000
#define O_KSW_XSUBO  0x40000
#define O_KSW_XSTART 0x80000
o_kswr_t o_ksw_align2(int qlen, uint8_t *query, int tlen, uint8_t *target, int m, const int8_t *mat,
                      int o_del, int e_del, int o_ins, int e_ins, int xtra);

/* ---- index I/O (upstream bwa.c / bntseq.c / bwt.c loaders) ---- */
o_idx_t *oracle_idx_load_files(const char *prefix);          /* 5-file reader */
o_idx_t *oracle_idx_from_image(uint8_t *mem, size_t l_mem, int is_mmap);
uint8_t *oracle_idx_to_image(const o_idx_t *idx, size_t *l_mem);
void oracle_idx_destroy(o_idx_t *idx);

/* ---- the jnibwa_* C ABI, oracle flavour (jnibwa.h:11-16) ---- */
int   oracle_createIndexFile(const char *prefix, const char *img);
o_idx_t *oracle_openIndex(int fd);
int   oracle_destroyIndex(o_idx_t *idx);
void *oracle_getRefContigNames(o_idx_t *idx, size_t *sz);
void *oracle_createAlignments(o_idx_t *idx, o_opt_t *opt, o_pestat_t *pes, char *seqs, size_t *sz);
int oracle_response_offsets(const uint8_t *buf, size_t len, uint32_t n_reads, int64_t *offs);   /* record boundaries of a response */
void *oracle_createAlignmentsAt(o_idx_t *idx, o_opt_t *opt, o_pestat_t *pes, char *seqs, size_t *sz, int64_t read_id0);   /* a slice of a larger call */
void *oracle_createDefaultOptions(void);
void  oracle_free(void *p);

/* ---- per-stage probes used by the kernel-level parity tests ---- */
/* all SMEM intervals of mem_collect_intv for one 2-bit read; returns count, fills out[4*i..] = x0,x1,size,info */
int oracle_collect_intv(o_idx_t *idx, const o_opt_t *opt, int len, const uint8_t *seq, uint64_t *out, int cap);
/* instrumentation counters accumulated by the oracle */
typedef struct { uint64_t n_ext, n_lf, n_sa, n_refbases, n_dp_cells, n_reads; } o_counters_t;
void oracle_counters_reset(void);
void oracle_counters_get(o_counters_t *c);

#ifdef __cplusplus
}
#endif
#endif
