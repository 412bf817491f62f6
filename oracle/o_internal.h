/*
 * o_internal.h -- CPU ORACLE (test infrastructure): shared internals.
 * See bwa_oracle.h for the provenance statement.
 */
#ifndef O_INTERNAL_H_
#define O_INTERNAL_H_

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "bwa_oracle.h"

extern __thread o_counters_t o_tl_cnt;

/*
 * The unstable introsort of upstream ksort.h, restated as a macro template.  The
 * permutation it produces on tied keys decides chain / region tie-breaking
 * (hard part 2), so the decision sequence is kept exactly:
 * n==2 special case; median-of-three quicksort with an explicit stack and a
 * depth limit of 2*ceil(log2 n) that falls back to combsort11; partitions of
 * <=16 elements are left for one final insertion-sort pass.
 */
#define O_SORT_DECL(name, type_t, lt) \
static void o_combsort_##name(size_t n, type_t a[]) \
{ \
	const double shrink = 1.2473309501039786540366528676643; \
	int do_swap; size_t gap = n; type_t tmp, *i, *j; \
	do { \
		if (gap > 2) { gap = (size_t)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; } \
		do_swap = 0; \
		for (i = a; i < a + n - gap; ++i) { \
			j = i + gap; \
			if (lt(*j, *i)) { tmp = *i; *i = *j; *j = tmp; do_swap = 1; } \
		} \
	} while (do_swap || gap > 2); \
	if (gap != 1) { \
		type_t *p, *q; \
		for (p = a + 1; p < a + n; ++p) \
			for (q = p; q > a && lt(*q, *(q-1)); --q) { tmp = *q; *q = *(q-1); *(q-1) = tmp; } \
	} \
} \
static void o_introsort_##name(size_t n, type_t a[]) \
{ \
	int d; \
	struct { type_t *left, *right; int depth; } *top, *stack; \
	type_t rp, swap_tmp; \
	type_t *s, *t, *i, *j, *k; \
	if (n < 1) return; \
	else if (n == 2) { \
		if (lt(a[1], a[0])) { swap_tmp = a[0]; a[0] = a[1]; a[1] = swap_tmp; } \
		return; \
	} \
	for (d = 2; 1ul << d < n; ++d); \
	stack = malloc(sizeof(*stack) * (sizeof(size_t) * d + 2)); \
	top = stack; s = a; t = a + (n - 1); d <<= 1; \
	while (1) { \
		if (s < t) { \
			if (--d == 0) { o_combsort_##name(t - s + 1, s); t = s; continue; } \
			i = s; j = t; k = i + ((j - i) >> 1) + 1; \
			if (lt(*k, *i)) { if (lt(*k, *j)) k = j; } \
			else k = lt(*j, *i) ? i : j; \
			rp = *k; \
			if (k != t) { swap_tmp = *k; *k = *t; *t = swap_tmp; } \
			for (;;) { \
				do ++i; while (lt(*i, rp)); \
				do --j; while (i <= j && lt(rp, *j)); \
				if (j <= i) break; \
				swap_tmp = *i; *i = *j; *j = swap_tmp; \
			} \
			swap_tmp = *i; *i = *t; *t = swap_tmp; \
			if (i - s > t - i) { \
				if (i - s > 16) { top->left = s; top->right = i - 1; top->depth = d; ++top; } \
				s = t - i > 16 ? i + 1 : t; \
			} else { \
				if (t - i > 16) { top->left = i + 1; top->right = t; top->depth = d; ++top; } \
				t = i - s > 16 ? i - 1 : s; \
			} \
		} else { \
			if (top == stack) { \
				type_t *p, *q; \
				free(stack); \
				for (p = a + 1; p < a + n; ++p) \
					for (q = p; q > a && lt(*q, *(q-1)); --q) { swap_tmp = *q; *q = *(q-1); *(q-1) = swap_tmp; } \
				return; \
			} else { --top; s = top->left; t = top->right; d = top->depth; } \
		} \
	} \
}

/* ---- aligner-side types (upstream bwamem.h) ---- */
typedef struct { int64_t rbeg; int32_t qbeg, len; int score; } o_seed_t;

typedef struct {
	int n, m, first, rid;
	uint32_t w:29, kept:2, is_alt:1;
	float frac_rep;
	int64_t pos;
	o_seed_t *seeds;
} o_chain_t;
typedef struct { size_t n, m; o_chain_t *a; } o_chain_v;

typedef struct {
	int64_t rb, re;
	int qb, qe;
	int rid;
	int score;
	int truesc;
	int sub;
	int alt_sc;
	int csub;
	int sub_n;
	int w;
	int seedcov;
	int secondary;
	int secondary_all;
	int seedlen0;
	int n_comp:30, is_alt:2;
	float frac_rep;
	uint64_t hash;
} o_alnreg_t;
typedef struct { size_t n, m; o_alnreg_t *a; } o_alnreg_v;

typedef struct {
	int64_t pos;
	int rid;
	int flag;
	uint32_t is_rev:1, is_alt:1, mapq:8, NM:22;
	int n_cigar;
	uint32_t *cigar;   /* n_cigar words followed by the NUL-terminated MD string */
	char *XA;
	int score, sub, alt_sc;
} o_aln_t;

typedef struct { size_t l, m; char *s; } o_str_t;

typedef struct { int l_seq; char *seq; o_str_t out; } o_read_t;

/* o_index.c */
int o_bns_pos2rid(const o_bns_t *bns, int64_t pos_f);
int o_bns_intv2rid(const o_bns_t *bns, int64_t rb, int64_t re);
uint8_t *o_bns_get_seq(int64_t l_pac, const uint8_t *pac, int64_t beg, int64_t end, int64_t *len);
uint8_t *o_bns_fetch_seq(const o_bns_t *bns, const uint8_t *pac, int64_t *beg, int64_t mid, int64_t *end, int *rid);
static inline int64_t o_bns_depos(const o_bns_t *bns, int64_t pos, int *is_rev)
{
	return (*is_rev = (pos >= bns->l_pac)) ? (bns->l_pac << 1) - 1 - pos : pos;
}

/* o_mem.c */
void o_align1_core(const o_opt_t *opt, const o_idx_t *idx, int l_seq, char *seq, o_alnreg_v *regs);
int  o_sort_dedup_patch(const o_opt_t *opt, const o_bns_t *bns, const uint8_t *pac, uint8_t *query, int n, o_alnreg_t *a);
int  o_mark_primary_se(const o_opt_t *opt, int n, o_alnreg_t *a, int64_t id);
void o_reorder_primary5(int T, o_alnreg_v *a);
int  o_approx_mapq_se(const o_opt_t *opt, const o_alnreg_t *a);
o_aln_t o_reg2aln(const o_opt_t *opt, const o_bns_t *bns, const uint8_t *pac, int l_query, const char *query, const o_alnreg_t *ar);
char **o_gen_alt(const o_opt_t *opt, const o_bns_t *bns, const uint8_t *pac, const o_alnreg_v *a, int l_query, const char *query);
void o_reg2sam(const o_opt_t *opt, const o_bns_t *bns, const uint8_t *pac, o_read_t *s, o_alnreg_v *a, int extra_flag, const o_aln_t *m);
void o_aln2out(const o_opt_t *opt, const o_bns_t *bns, o_str_t *str, o_read_t *s, int n, const o_aln_t *list, int which, const o_aln_t *m);
uint64_t o_hash_64(uint64_t key);

/* o_pair.c */
void o_pestat(const o_opt_t *opt, int64_t l_pac, int n, const o_alnreg_v *regs, o_pestat_t pes[4]);
int  o_sam_pe(const o_opt_t *opt, const o_bns_t *bns, const uint8_t *pac, const o_pestat_t pes[4], uint64_t id, o_read_t s[2], o_alnreg_v a[2]);

#endif
