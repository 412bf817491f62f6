/*
 * o_shim.c -- CPU ORACLE (test infrastructure): the jnibwa_* C ABI on top of the
 * restated algorithm, so tests can compare raw result buffers byte for byte with
 * the product library.
 *
 * Follows the reference's src/main/c/jnibwa.c:126-235 (index image write / mmap
 * open / destroy / contig names / batch parse + result concat) and upstream
 * mem_process_seqs / mem_opt_init (reached at jnibwa.c:214 and
 * ...BwaMemIndex.c:91).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
#include <math.h>
#include <pthread.h>
#include <sys/stat.h>
#include <sys/mman.h>
#include "bwa_oracle.h"
#include "o_internal.h"

static o_counters_t g_cnt;
static pthread_mutex_t g_cnt_lock = PTHREAD_MUTEX_INITIALIZER;

void oracle_counters_reset(void) { memset(&g_cnt, 0, sizeof g_cnt); }
void oracle_counters_get(o_counters_t *c) { *c = g_cnt; }
static void counters_flush(void)
{
	pthread_mutex_lock(&g_cnt_lock);
	g_cnt.n_ext += o_tl_cnt.n_ext; g_cnt.n_lf += o_tl_cnt.n_lf; g_cnt.n_sa += o_tl_cnt.n_sa;
	g_cnt.n_refbases += o_tl_cnt.n_refbases; g_cnt.n_dp_cells += o_tl_cnt.n_dp_cells; g_cnt.n_reads += o_tl_cnt.n_reads;
	pthread_mutex_unlock(&g_cnt_lock);
	memset(&o_tl_cnt, 0, sizeof o_tl_cnt);
}

void oracle_free(void *p) { free(p); }

/* mem_opt_init + bwa_fill_scmat (defaults: SURVEY.md App. A.5) */
void *oracle_createDefaultOptions(void)
{
	o_opt_t *o = calloc(1, sizeof(o_opt_t));
	int i, j, k;
	o->flag = 0;
	o->a = 1; o->b = 4;
	o->o_del = o->o_ins = 6;
	o->e_del = o->e_ins = 1;
	o->w = 100;
	o->T = 30;
	o->zdrop = 100;
	o->pen_unpaired = 17;
	o->pen_clip5 = o->pen_clip3 = 5;
	o->max_mem_intv = 20;
	o->min_seed_len = 19;
	o->split_width = 10;
	o->max_occ = 500;
	o->max_chain_gap = 10000;
	o->max_ins = 10000;
	o->mask_level = 0.50;
	o->drop_ratio = 0.50;
	o->XA_drop_ratio = 0.80;
	o->split_factor = 1.5;
	o->chunk_size = 10000000;
	o->n_threads = 1;
	o->max_XA_hits = 5;
	o->max_XA_hits_alt = 200;
	o->max_matesw = 50;
	o->mask_level_redun = 0.95;
	o->min_chain_weight = 0;
	o->max_chain_extend = 1 << 30;
	o->mapQ_coef_len = 50; o->mapQ_coef_fac = log(o->mapQ_coef_len);
	for (i = k = 0; i < 4; ++i) {
		for (j = 0; j < 4; ++j) o->mat[k++] = i == j ? o->a : -o->b;
		o->mat[k++] = -1;
	}
	for (j = 0; j < 5; ++j) o->mat[k++] = -1;
	return o;
}

int oracle_createIndexFile(const char *prefix, const char *img)
{
	o_idx_t *idx = oracle_idx_load_files(prefix);
	int fd;
	size_t len;
	uint8_t *buf;
	if (!idx) return 2;
	fd = open(img, O_WRONLY | O_CREAT | O_TRUNC, 0644);
	if (fd == -1) { oracle_idx_destroy(idx); return 2; }
	len = idx->l_mem; buf = idx->mem;
	while (len) {
		size_t to_write = len > (1L << 30) ? (1L << 30) : len;
		if (write(fd, buf, to_write) != (ssize_t)to_write) { close(fd); oracle_idx_destroy(idx); return 2; }
		buf += to_write; len -= to_write;
	}
	if (close(fd) != 0) { oracle_idx_destroy(idx); return 2; }
	oracle_idx_destroy(idx);
	return 0;
}

o_idx_t *oracle_openIndex(int fd)
{
	struct stat st;
	uint8_t *mem;
	o_idx_t *idx;
	if (fstat(fd, &st) == -1) return 0;
	mem = mmap(0, st.st_size, PROT_READ, MAP_SHARED, fd, 0);
	close(fd);
	if (mem == MAP_FAILED) return 0;
	idx = oracle_idx_from_image(mem, st.st_size, 1);
	if (!idx) munmap(mem, st.st_size);
	return idx;
}

int oracle_destroyIndex(o_idx_t *idx)
{
	void *mem = idx->mem;
	size_t len = idx->l_mem;
	int is_mmap = idx->is_mmap;
	oracle_idx_destroy(idx);
	return is_mmap ? munmap(mem, len) : 0;
}

void *oracle_getRefContigNames(o_idx_t *idx, size_t *sz)
{
	int n = idx->bns.n_seqs, i;
	int bufSize = 4 + 4 * n;
	char *buf, *p;
	for (i = 0; i < n; ++i) bufSize += (int)strlen(idx->bns.anns[i].name) + 1;
	buf = calloc(bufSize, 1);
	*(int32_t*)buf = n;
	p = buf + 4;
	for (i = 0; i < n; ++i) {
		size_t len = strlen(idx->bns.anns[i].name);
		*(int32_t*)p = (int32_t)len; p += 4;
		memcpy(p, idx->bns.anns[i].name, len); p += len;
	}
	*sz = bufSize;
	return buf;
}

/* ---- mem_process_seqs: phase 1 (regions), optional pestat, phase 2 (records) ---- */

typedef struct {
	const o_opt_t *opt;
	const o_idx_t *idx;
	o_read_t *seqs;
	o_alnreg_v *regs;
	const o_pestat_t *pes;
	int n_units, phase, n_threads, tid;
	int64_t n_processed;
} work_t;

static void *worker(void *arg)
{
	work_t *w = arg;
	const o_opt_t *opt = w->opt;
	int i, pe = (opt->flag & O_F_PE) != 0;
	for (i = w->tid; i < w->n_units; i += w->n_threads) {
		if (w->phase == 1) {
			if (!pe) {
				o_align1_core(opt, w->idx, w->seqs[i].l_seq, w->seqs[i].seq, &w->regs[i]);
				o_tl_cnt.n_reads++;
			} else {
				o_align1_core(opt, w->idx, w->seqs[i<<1|0].l_seq, w->seqs[i<<1|0].seq, &w->regs[i<<1|0]);
				o_align1_core(opt, w->idx, w->seqs[i<<1|1].l_seq, w->seqs[i<<1|1].seq, &w->regs[i<<1|1]);
				o_tl_cnt.n_reads += 2;
			}
		} else {
			if (!pe) {
				o_mark_primary_se(opt, (int)w->regs[i].n, w->regs[i].a, w->n_processed + i);
				if (opt->flag & O_F_PRIMARY5) o_reorder_primary5(opt->T, &w->regs[i]);
				o_reg2sam(opt, &w->idx->bns, w->idx->pac, &w->seqs[i], &w->regs[i], 0, 0);
				free(w->regs[i].a);
			} else {
				o_sam_pe(opt, &w->idx->bns, w->idx->pac, w->pes, (w->n_processed >> 1) + i, &w->seqs[i<<1], &w->regs[i<<1]);
				free(w->regs[i<<1|0].a); free(w->regs[i<<1|1].a);
			}
		}
	}
	counters_flush();
	return 0;
}

static void run_phase(work_t *proto, int phase)
{
	int nt = proto->n_threads, t;
	pthread_t *th = malloc(sizeof(pthread_t) * nt);
	work_t *w = malloc(sizeof(work_t) * nt);
	for (t = 0; t < nt; ++t) { w[t] = *proto; w[t].phase = phase; w[t].tid = t; }
	if (nt == 1) worker(&w[0]);
	else {
		for (t = 0; t < nt; ++t) pthread_create(&th[t], 0, worker, &w[t]);
		for (t = 0; t < nt; ++t) pthread_join(th[t], 0);
	}
	free(th); free(w);
}

void *oracle_createAlignmentsAt(o_idx_t *idx, o_opt_t *opt, o_pestat_t *pes0, char *pSeq, size_t *pBufSize, int64_t read_id0);

void *oracle_createAlignments(o_idx_t *idx, o_opt_t *opt, o_pestat_t *pes0, char *pSeq, size_t *pBufSize)
{
	return oracle_createAlignmentsAt(idx, opt, pes0, pSeq, pBufSize, 0);
}

/* read_id0: index of the first read within the (larger) call these reads are a slice of -- upstream's n_processed, which
 * enters the tie-breaking hash of mem_mark_primary_se (bwamem.c) */
void *oracle_createAlignmentsAt(o_idx_t *idx, o_opt_t *opt, o_pestat_t *pes0, char *pSeq, size_t *pBufSize, int64_t read_id0)
{
	uint32_t nSeqs = *(uint32_t*)pSeq, i;
	o_read_t *seqs = calloc(nSeqs ? nSeqs : 1, sizeof(o_read_t));
	o_pestat_t pes[4];
	work_t w;
	size_t tot = 0;
	char *res, *p;
	pSeq += 4;
	for (i = 0; i < nSeqs; ++i) {           /* jnibwa.c:204-212 */
		size_t l = strlen(pSeq);
		seqs[i].l_seq = (int)l; seqs[i].seq = pSeq;
		pSeq += l + 1;
	}
	memset(&w, 0, sizeof w);
	w.opt = opt; w.idx = idx; w.seqs = seqs; w.n_processed = read_id0; w.pes = pes;
	w.regs = calloc(nSeqs ? nSeqs : 1, sizeof(o_alnreg_v));
	w.n_units = (opt->flag & O_F_PE) ? nSeqs >> 1 : nSeqs;
	w.n_threads = opt->n_threads > 0 ? opt->n_threads : 1;
	run_phase(&w, 1);
	if (opt->flag & O_F_PE) {
		if (pes0) memcpy(pes, pes0, 4 * sizeof(o_pestat_t));
		else o_pestat(opt, idx->bns.l_pac, nSeqs, w.regs, pes);
	}
	run_phase(&w, 2);
	free(w.regs);
	for (i = 0; i < nSeqs; ++i) tot += seqs[i].out.l;  /* jnibwa.c:216-233 */
	res = malloc(tot ? tot : 1);
	for (i = 0, p = res; i < nSeqs; ++i) {
		if (seqs[i].out.s) { memcpy(p, seqs[i].out.s, seqs[i].out.l); p += seqs[i].out.l; free(seqs[i].out.s); }
	}
	free(seqs);
	*pBufSize = tot;
	return res;
}

/* byte offset of every read's records in a response (the reference's bufLen walk, jnibwa.c:99-124); offs[n_reads] = end.
 * Returns 0, or -1 when the walk runs past len. */
int oracle_response_offsets(const uint8_t *buf, size_t len, uint32_t n_reads, int64_t *offs)
{
	size_t off = 0;
	uint32_t r;
	for (r = 0; r < n_reads; ++r) {
		int32_t na, a;
		offs[r] = (int64_t)off;
		if (off + 4 > len) return -1;
		memcpy(&na, buf + off, 4); off += 4;
		for (a = 0; a < na; ++a) {
			int32_t fm, flag, nc, nmd, nxa;
			if (off + 4 > len) return -1;
			memcpy(&fm, buf + off, 4); off += 4;
			flag = (fm >> 16) & 0xffff;
			if (!(flag & 4)) {
				if (off + 24 > len) return -1;
				memcpy(&nc, buf + off + 20, 4); off += 24 + 4 * (size_t)nc;
				if (off + 4 > len) return -1;
				memcpy(&nmd, buf + off, 4); off += 4 + (((size_t)nmd + 3) & ~(size_t)3);
				if (off + 4 > len) return -1;
				memcpy(&nxa, buf + off, 4); off += 4 + (((size_t)nxa + 3) & ~(size_t)3);
			}
			if ((flag & 9) == 1) off += 12;
		}
	}
	offs[n_reads] = (int64_t)off;
	return off <= len ? 0 : -1;
}

/* ---- test hooks: expose the order-exact sort on (x, y) pairs compared by x only, so the tie permutation is observable ---- */
typedef struct { uint64_t x, y; } o_pairx_t;
#define pairx_lt(a, b) ((a).x < (b).x)
O_SORT_DECL(pairx, o_pairx_t, pairx_lt)
void oracle_test_sort_pairs(size_t n, uint64_t *xy) { o_introsort_pairx(n, (o_pairx_t*)xy); }
