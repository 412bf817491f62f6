/*
 * o_mem.c -- CPU ORACLE (test infrastructure): seeding -> chaining -> extension ->
 * region post-processing -> records, single-end.
 *
 * Restates upstream lh3/bwa@cb950614 bwamem.c, bwamem_extra.c and the CIGAR/MD
 * part of bwa.c, reached from the
 * reference only through jnibwa.c:214; the record hook is the reference's own
 * fmt_BAMish (jnibwa.c:43-97, row a18) and is restated in o_aln2out().
 *
 * Float/double usage mirrors the C promotion rules of the upstream expressions
 * (hard part 4); build with -ffp-contract=off.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>
#include <assert.h>
#include "bwa_oracle.h"
#include "o_internal.h"

/* ASCII -> 0..4 (upstream nst_nt4_table; anything that is not ACGTacgt becomes 4) */
static inline int nt4(int c)
{
	switch (c) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': return 3;
	default: return 4;
	}
}

uint64_t o_hash_64(uint64_t key)
{
	key += ~(key << 32);
	key ^= (key >> 22);
	key += ~(key << 13);
	key ^= (key >> 8);
	key += (key << 3);
	key ^= (key >> 15);
	key += ~(key << 27);
	key ^= (key >> 31);
	return key;
}

/* ---------------- sorts (exact ksort introsort; see o_internal.h) ---------------- */
#define intv_lt(a, b) ((a).info < (b).info)
O_SORT_DECL(intv, o_intv_t, intv_lt)
#define flt_lt(a, b) ((a).w > (b).w)
O_SORT_DECL(flt, o_chain_t, flt_lt)
#define u64_lt(a, b) ((a) < (b))
O_SORT_DECL(u64, uint64_t, u64_lt)
#define alnreg_slt2(a, b) ((a).re < (b).re)
O_SORT_DECL(ars2, o_alnreg_t, alnreg_slt2)
#define alnreg_slt(a, b) ((a).score > (b).score || ((a).score == (b).score && ((a).rb < (b).rb || ((a).rb == (b).rb && (a).qb < (b).qb))))
O_SORT_DECL(ars, o_alnreg_t, alnreg_slt)
#define alnreg_hlt(a, b)  ((a).score > (b).score || ((a).score == (b).score && ((a).is_alt < (b).is_alt || ((a).is_alt == (b).is_alt && (a).hash < (b).hash))))
O_SORT_DECL(ars_hash, o_alnreg_t, alnreg_hlt)
#define alnreg_hlt2(a, b) ((a).is_alt < (b).is_alt || ((a).is_alt == (b).is_alt && ((a).score > (b).score || ((a).score == (b).score && (a).hash < (b).hash))))
O_SORT_DECL(ars_hash2, o_alnreg_t, alnreg_hlt2)

void o_sort_u64(size_t n, uint64_t *a) { o_introsort_u64(n, a); }

/* ---------------- seeding: three passes (row a6) ---------------- */

typedef struct { o_intv_v mem, mem1, tmpv[2]; } smem_aux_t;

static inline void intv_push(o_intv_v *v, const o_intv_t *p)
{
	if (v->n == v->m) { v->m = v->m ? v->m << 1 : 4; v->a = realloc(v->a, v->m * sizeof(o_intv_t)); }
	v->a[v->n++] = *p;
}

static void collect_intv(const o_opt_t *opt, const o_bwt_t *bwt, int len, const uint8_t *seq, smem_aux_t *a)
{
	int i, k, x = 0, old_n;
	int start_width = 1;
	int split_len = (int)(opt->min_seed_len * opt->split_factor + .499);
	a->mem.n = 0;
	/* pass 1: all SMEMs */
	while (x < len) {
		if (seq[x] < 4) {
			x = o_bwt_smem1(bwt, len, seq, x, start_width, &a->mem1, a->tmpv);
			for (i = 0; i < (int)a->mem1.n; ++i) {
				o_intv_t *p = &a->mem1.a[i];
				int slen = (uint32_t)p->info - (p->info >> 32);
				if (slen >= opt->min_seed_len) intv_push(&a->mem, p);
			}
		} else ++x;
	}
	/* pass 2: re-seed inside long, low-occurrence SMEMs */
	old_n = (int)a->mem.n;
	for (k = 0; k < old_n; ++k) {
		o_intv_t *p = &a->mem.a[k];
		int start = p->info >> 32, end = (int32_t)p->info;
		if (end - start < split_len || p->x[2] > (bwtint_t)opt->split_width) continue;
		o_bwt_smem1(bwt, len, seq, (start + end) >> 1, (int)(p->x[2] + 1), &a->mem1, a->tmpv);
		for (i = 0; i < (int)a->mem1.n; ++i)
			if ((int)((uint32_t)a->mem1.a[i].info - (a->mem1.a[i].info >> 32)) >= opt->min_seed_len)
				intv_push(&a->mem, &a->mem1.a[i]);
	}
	/* pass 3: greedy forward seeds */
	if (opt->max_mem_intv > 0) {
		x = 0;
		while (x < len) {
			if (seq[x] < 4) {
				o_intv_t m;
				x = o_bwt_seed_strategy1(bwt, len, seq, x, opt->min_seed_len, (int)opt->max_mem_intv, &m);
				if (m.x[2] > 0) intv_push(&a->mem, &m);
			} else ++x;
		}
	}
	o_introsort_intv(a->mem.n, a->mem.a);
}

int oracle_collect_intv(o_idx_t *idx, const o_opt_t *opt, int len, const uint8_t *seq, uint64_t *out, int cap)
{
	smem_aux_t a;
	int i, n;
	memset(&a, 0, sizeof a);
	collect_intv(opt, &idx->bwt, len, seq, &a);
	n = (int)a.mem.n;
	for (i = 0; i < n && i < cap; ++i) {
		out[4*i] = a.mem.a[i].x[0]; out[4*i+1] = a.mem.a[i].x[1]; out[4*i+2] = a.mem.a[i].x[2]; out[4*i+3] = a.mem.a[i].info;
	}
	free(a.mem.a); free(a.mem1.a); free(a.tmpv[0].a); free(a.tmpv[1].a);
	return n;
}

/* ---------------- chaining (row a8): the klib B-tree, keys held by value ---------------- */

#define BT_T 5                     /* ((512-4-8)/(8+sizeof(o_chain_t)=40)+1)>>1 */
#define BT_MAXK (2 * BT_T - 1)

typedef struct btnode {
	int is_internal, n;
	o_chain_t key[BT_MAXK];
	struct btnode *ptr[BT_MAXK + 1];
} btnode_t;

typedef struct { btnode_t *root; int n_keys; } btree_t;

#define chain_cmp(a, b) (((b).pos < (a).pos) - ((a).pos < (b).pos))

static int bt_getp_aux(const btnode_t *x, const o_chain_t *k, int *r)
{
	int tr, *rr, begin = 0, end = x->n;
	if (x->n == 0) return -1;
	rr = r ? r : &tr;
	while (begin < end) {
		int mid = (begin + end) >> 1;
		if (chain_cmp(x->key[mid], *k) < 0) begin = mid + 1;
		else end = mid;
	}
	if (begin == x->n) { *rr = 1; return x->n - 1; }
	if ((*rr = chain_cmp(*k, x->key[begin])) < 0) --begin;
	return begin;
}

static void bt_interval(btree_t *b, const o_chain_t *k, o_chain_t **lower, o_chain_t **upper)
{
	int i, r = 0;
	btnode_t *x = b->root;
	*lower = *upper = 0;
	while (x) {
		i = bt_getp_aux(x, k, &r);
		if (i >= 0 && r == 0) { *lower = *upper = &x->key[i]; return; }
		if (i >= 0) *lower = &x->key[i];
		if (i < x->n - 1) *upper = &x->key[i + 1];
		if (x->is_internal == 0) return;
		x = x->ptr[i + 1];
	}
}

static void bt_split(btnode_t *x, int i, btnode_t *y)
{
	btnode_t *z = calloc(1, sizeof(btnode_t));
	z->is_internal = y->is_internal;
	z->n = BT_T - 1;
	memcpy(z->key, y->key + BT_T, sizeof(o_chain_t) * (BT_T - 1));
	if (y->is_internal) memcpy(z->ptr, y->ptr + BT_T, sizeof(void*) * BT_T);
	y->n = BT_T - 1;
	memmove(x->ptr + i + 2, x->ptr + i + 1, sizeof(void*) * (x->n - i));
	x->ptr[i + 1] = z;
	memmove(x->key + i + 1, x->key + i, sizeof(o_chain_t) * (x->n - i));
	x->key[i] = y->key[BT_T - 1];
	++x->n;
}

static void bt_put_aux(btnode_t *x, const o_chain_t *k)
{
	int i;
	if (x->is_internal == 0) {
		i = bt_getp_aux(x, k, 0);
		if (i != x->n - 1)
			memmove(x->key + i + 2, x->key + i + 1, (x->n - i - 1) * sizeof(o_chain_t));
		x->key[i + 1] = *k;
		++x->n;
	} else {
		i = bt_getp_aux(x, k, 0) + 1;
		if (x->ptr[i]->n == BT_MAXK) {
			bt_split(x, i, x->ptr[i]);
			if (chain_cmp(*k, x->key[i]) > 0) ++i;
		}
		bt_put_aux(x->ptr[i], k);
	}
}

static void bt_put(btree_t *b, const o_chain_t *k)
{
	btnode_t *r = b->root, *s;
	++b->n_keys;
	if (r->n == BT_MAXK) {
		s = calloc(1, sizeof(btnode_t));
		b->root = s; s->is_internal = 1; s->n = 0;
		s->ptr[0] = r;
		bt_split(s, 0, r);
		r = s;
	}
	bt_put_aux(r, k);
}

static void bt_traverse_free(btnode_t *x, o_chain_v *out)
{
	int i;
	if (x->is_internal) {
		for (i = 0; i < x->n; ++i) {
			bt_traverse_free(x->ptr[i], out);
			out->a[out->n++] = x->key[i];
		}
		bt_traverse_free(x->ptr[x->n], out);
	} else for (i = 0; i < x->n; ++i) out->a[out->n++] = x->key[i];
	free(x);
}

static int test_and_merge(const o_opt_t *opt, int64_t l_pac, o_chain_t *c, const o_seed_t *p, int seed_rid)
{
	int64_t qend, rend, x, y;
	const o_seed_t *last = &c->seeds[c->n - 1];
	qend = last->qbeg + last->len;
	rend = last->rbeg + last->len;
	if (seed_rid != c->rid) return 0;
	if (p->qbeg >= c->seeds[0].qbeg && p->qbeg + p->len <= qend && p->rbeg >= c->seeds[0].rbeg && p->rbeg + p->len <= rend)
		return 1; /* contained seed: absorbed */
	if ((last->rbeg < l_pac || c->seeds[0].rbeg < l_pac) && p->rbeg >= l_pac) return 0; /* different strand */
	x = p->qbeg - last->qbeg;
	y = p->rbeg - last->rbeg;
	if (y >= 0 && x - y <= opt->w && y - x <= opt->w && x - last->len < opt->max_chain_gap && y - last->len < opt->max_chain_gap) {
		if (c->n == c->m) {
			c->m <<= 1;
			c->seeds = realloc(c->seeds, c->m * sizeof(o_seed_t));
		}
		c->seeds[c->n++] = *p;
		return 1;
	}
	return 0;
}

static o_chain_v mem_chain(const o_opt_t *opt, const o_idx_t *idx, int len, const uint8_t *seq)
{
	const o_bwt_t *bwt = &idx->bwt;
	const o_bns_t *bns = &idx->bns;
	int i, b, e, l_rep;
	int64_t l_pac = bns->l_pac;
	o_chain_v chain = { 0, 0, 0 };
	btree_t tree;
	smem_aux_t aux;

	if (len < opt->min_seed_len) return chain;
	memset(&aux, 0, sizeof aux);
	tree.root = calloc(1, sizeof(btnode_t)); tree.n_keys = 0;
	collect_intv(opt, bwt, len, seq, &aux);
	for (i = 0, b = e = l_rep = 0; i < (int)aux.mem.n; ++i) { /* fraction of the read covered by repetitive seeds */
		o_intv_t *p = &aux.mem.a[i];
		int sb = (p->info >> 32), se = (uint32_t)p->info;
		if (p->x[2] <= (bwtint_t)opt->max_occ) continue;
		if (sb > e) l_rep += e - b, b = sb, e = se;
		else e = e > se ? e : se;
	}
	l_rep += e - b;
	for (i = 0; i < (int)aux.mem.n; ++i) {
		o_intv_t *p = &aux.mem.a[i];
		int step, count, slen = (uint32_t)p->info - (p->info >> 32);
		int64_t k;
		step = p->x[2] > (bwtint_t)opt->max_occ ? (int)(p->x[2] / opt->max_occ) : 1;
		for (k = count = 0; k < (int64_t)p->x[2] && count < opt->max_occ; k += step, ++count) {
			o_chain_t tmp, *lower, *upper;
			o_seed_t s;
			int rid, to_add = 0;
			memset(&tmp, 0, sizeof tmp);
			s.rbeg = tmp.pos = o_bwt_sa(bwt, p->x[0] + k);
			s.qbeg = p->info >> 32;
			s.score = s.len = slen;
			rid = o_bns_intv2rid(bns, s.rbeg, s.rbeg + s.len);
			if (rid < 0) continue; /* spans two contigs or the fwd/rev boundary */
			if (tree.n_keys) {
				bt_interval(&tree, &tmp, &lower, &upper);
				if (!lower || !test_and_merge(opt, l_pac, lower, &s, rid)) to_add = 1;
			} else to_add = 1;
			if (to_add) {
				tmp.n = 1; tmp.m = 4;
				tmp.seeds = calloc(tmp.m, sizeof(o_seed_t));
				tmp.seeds[0] = s;
				tmp.rid = rid;
				tmp.is_alt = !!bns->anns[rid].is_alt;
				bt_put(&tree, &tmp);
			}
		}
	}
	free(aux.mem.a); free(aux.mem1.a); free(aux.tmpv[0].a); free(aux.tmpv[1].a);
	chain.m = tree.n_keys ? tree.n_keys : 1;
	chain.a = malloc(chain.m * sizeof(o_chain_t));
	bt_traverse_free(tree.root, &chain);
	for (i = 0; i < (int)chain.n; ++i) chain.a[i].frac_rep = (float)l_rep / len;
	return chain;
}

/* ---------------- chain weight + filter (row a9) ---------------- */

static int chain_weight(const o_chain_t *c)
{
	int64_t end;
	int j, w = 0, tmp;
	for (j = 0, end = 0; j < c->n; ++j) {
		const o_seed_t *s = &c->seeds[j];
		if (s->qbeg >= end) w += s->len;
		else if (s->qbeg + s->len > end) w += s->qbeg + s->len - end;
		end = end > s->qbeg + s->len ? end : s->qbeg + s->len;
	}
	tmp = w; w = 0;
	for (j = 0, end = 0; j < c->n; ++j) {
		const o_seed_t *s = &c->seeds[j];
		if (s->rbeg >= end) w += s->len;
		else if (s->rbeg + s->len > end) w += s->rbeg + s->len - end;
		end = end > s->rbeg + s->len ? end : s->rbeg + s->len;
	}
	w = w < tmp ? w : tmp;
	return w < 1 << 30 ? w : (1 << 30) - 1;
}

#define chn_beg(ch) ((ch).seeds->qbeg)
#define chn_end(ch) ((ch).seeds[(ch).n-1].qbeg + (ch).seeds[(ch).n-1].len)

static int chain_flt(const o_opt_t *opt, int n_chn, o_chain_t *a)
{
	int i, k, n_kept = 0, *chains;
	if (n_chn == 0) return 0;
	for (i = k = 0; i < n_chn; ++i) {
		o_chain_t *c = &a[i];
		c->first = -1; c->kept = 0;
		c->w = chain_weight(c);
		if ((int)c->w < opt->min_chain_weight) free(c->seeds);
		else a[k++] = *c;
	}
	n_chn = k;
	if (n_chn == 0) return 0;
	o_introsort_flt(n_chn, a);
	chains = malloc(sizeof(int) * n_chn);
	a[0].kept = 3;
	chains[n_kept++] = 0;
	for (i = 1; i < n_chn; ++i) {
		int large_ovlp = 0;
		for (k = 0; k < n_kept; ++k) {
			int j = chains[k];
			int b_max = chn_beg(a[j]) > chn_beg(a[i]) ? chn_beg(a[j]) : chn_beg(a[i]);
			int e_min = chn_end(a[j]) < chn_end(a[i]) ? chn_end(a[j]) : chn_end(a[i]);
			if (e_min > b_max && (!a[j].is_alt || a[i].is_alt)) {
				int li = chn_end(a[i]) - chn_beg(a[i]);
				int lj = chn_end(a[j]) - chn_beg(a[j]);
				int min_l = li < lj ? li : lj;
				if ((float)(e_min - b_max) >= (float)min_l * opt->mask_level && min_l < opt->max_chain_gap) {
					large_ovlp = 1;
					if (a[j].first < 0) a[j].first = i;
					if ((float)(int)a[i].w < (float)(int)a[j].w * opt->drop_ratio && (int)a[j].w - (int)a[i].w >= opt->min_seed_len << 1)
						break;
				}
			}
		}
		if (k == n_kept) {
			chains[n_kept++] = i;
			a[i].kept = large_ovlp ? 2 : 3;
		}
	}
	for (i = 0; i < n_kept; ++i) {
		o_chain_t *c = &a[chains[i]];
		if (c->first >= 0) a[c->first].kept = 1;
	}
	free(chains);
	for (i = k = 0; i < n_chn; ++i) {
		if (a[i].kept == 0 || a[i].kept == 3) continue;
		if (++k >= opt->max_chain_extend) break;
	}
	for (; i < n_chn; ++i)
		if (a[i].kept < 3) a[i].kept = 0;
	for (i = k = 0; i < n_chn; ++i) {
		o_chain_t *c = &a[i];
		if (c->kept == 0) free(c->seeds);
		else a[k++] = a[i];
	}
	return k;
}

/* ---------------- seed re-scoring for long reads (row a10) ---------------- */

#define MEM_SHORT_EXT 50
#define MEM_SHORT_LEN 200
#define MEM_HSP_COEF 1.1f
#define MEM_MINSC_COEF 5.5f
#define MEM_SEEDSW_COEF 0.05f

static int seed_sw(const o_opt_t *opt, const o_bns_t *bns, const uint8_t *pac, int l_query, const uint8_t *query, const o_seed_t *s)
{
	int qb, qe, rid;
	int64_t rb, re, mid, l_pac = bns->l_pac;
	uint8_t *rseq, *qtmp;
	o_kswr_t x;
	if (s->len >= MEM_SHORT_LEN) return -1;
	qb = s->qbeg, qe = s->qbeg + s->len;
	rb = s->rbeg, re = s->rbeg + s->len;
	mid = (rb + re) >> 1;
	qb -= MEM_SHORT_EXT; qb = qb > 0 ? qb : 0;
	qe += MEM_SHORT_EXT; qe = qe < l_query ? qe : l_query;
	rb -= MEM_SHORT_EXT; rb = rb > 0 ? rb : 0;
	re += MEM_SHORT_EXT; re = re < l_pac << 1 ? re : l_pac << 1;
	if (rb < l_pac && l_pac < re) {
		if (mid < l_pac) re = l_pac;
		else rb = l_pac;
	}
	if (qe - qb >= MEM_SHORT_LEN || re - rb >= MEM_SHORT_LEN) return -1;
	rseq = o_bns_fetch_seq(bns, pac, &rb, mid, &re, &rid);
	qtmp = malloc(qe - qb + 1);
	memcpy(qtmp, query + qb, qe - qb);
	x = o_ksw_align2(qe - qb, qtmp, (int)(re - rb), rseq, 5, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, O_KSW_XSTART);
	free(qtmp); free(rseq);
	return x.score;
}

static void flt_chained_seeds(const o_opt_t *opt, const o_bns_t *bns, const uint8_t *pac, int l_query, const uint8_t *query, int n_chn, o_chain_t *a)
{
	double min_l = opt->min_chain_weight ? MEM_HSP_COEF * opt->min_chain_weight : MEM_MINSC_COEF * log(l_query);
	int i, j, k, min_HSP_score = (int)(opt->a * min_l + .499);
	if (min_l > MEM_SEEDSW_COEF * l_query) return; /* short reads: nothing to do */
	for (i = 0; i < n_chn; ++i) {
		o_chain_t *c = &a[i];
		for (j = k = 0; j < c->n; ++j) {
			o_seed_t *s = &c->seeds[j];
			s->score = seed_sw(opt, bns, pac, l_query, query, s);
			if (s->score < 0 || s->score >= min_HSP_score) {
				s->score = s->score < 0 ? s->len * opt->a : s->score;
				c->seeds[k++] = *s;
			}
		}
		c->n = k;
	}
}

/* ---------------- chain -> regions by banded extension (row a11) ---------------- */

static inline int cal_max_gap(const o_opt_t *opt, int qlen)
{
	int l_del = (int)((double)(qlen * opt->a - opt->o_del) / opt->e_del + 1.);
	int l_ins = (int)((double)(qlen * opt->a - opt->o_ins) / opt->e_ins + 1.);
	int l = l_del > l_ins ? l_del : l_ins;
	l = l > 1 ? l : 1;
	return l < opt->w << 1 ? l : opt->w << 1;
}

#define MAX_BAND_TRY 2

static inline o_alnreg_t *reg_pushp(o_alnreg_v *v)
{
	if (v->n == v->m) { v->m = v->m ? v->m << 1 : 2; v->a = realloc(v->a, v->m * sizeof(o_alnreg_t)); }
	return &v->a[v->n++];
}

static void chain2aln(const o_opt_t *opt, const o_bns_t *bns, const uint8_t *pac, int l_query, const uint8_t *query, const o_chain_t *c, o_alnreg_v *av)
{
	int i, k, rid, max_off[2], aw[2];
	int64_t l_pac = bns->l_pac, rmax[2], tmp, max = 0;
	const o_seed_t *s;
	uint8_t *rseq = 0;
	uint64_t *srt;

	if (c->n == 0) return;
	rmax[0] = l_pac << 1; rmax[1] = 0;
	for (i = 0; i < c->n; ++i) {
		int64_t b, e;
		const o_seed_t *t = &c->seeds[i];
		b = t->rbeg - (t->qbeg + cal_max_gap(opt, t->qbeg));
		e = t->rbeg + t->len + ((l_query - t->qbeg - t->len) + cal_max_gap(opt, l_query - t->qbeg - t->len));
		rmax[0] = rmax[0] < b ? rmax[0] : b;
		rmax[1] = rmax[1] > e ? rmax[1] : e;
		if (t->len > max) max = t->len;
	}
	rmax[0] = rmax[0] > 0 ? rmax[0] : 0;
	rmax[1] = rmax[1] < l_pac << 1 ? rmax[1] : l_pac << 1;
	if (rmax[0] < l_pac && l_pac < rmax[1]) {
		if (c->seeds[0].rbeg < l_pac) rmax[1] = l_pac;
		else rmax[0] = l_pac;
	}
	rseq = o_bns_fetch_seq(bns, pac, &rmax[0], c->seeds[0].rbeg, &rmax[1], &rid);
	assert(c->rid == rid);

	srt = malloc(c->n * 8);
	for (i = 0; i < c->n; ++i) srt[i] = (uint64_t)c->seeds[i].score << 32 | i;
	o_introsort_u64(c->n, srt);

	for (k = c->n - 1; k >= 0; --k) {
		o_alnreg_t *a;
		s = &c->seeds[(uint32_t)srt[k]];
		for (i = 0; i < (int)av->n; ++i) { /* is the seed already covered by an earlier region? */
			o_alnreg_t *p = &av->a[i];
			int64_t rd;
			int qd, w, max_gap;
			if (s->rbeg < p->rb || s->rbeg + s->len > p->re || s->qbeg < p->qb || s->qbeg + s->len > p->qe) continue;
			if (s->len - p->seedlen0 > .1 * l_query) continue;
			qd = s->qbeg - p->qb; rd = s->rbeg - p->rb;
			max_gap = cal_max_gap(opt, qd < rd ? qd : (int)rd);
			w = max_gap < p->w ? max_gap : p->w;
			if (qd - rd < w && rd - qd < w) break;
			qd = p->qe - (s->qbeg + s->len); rd = p->re - (s->rbeg + s->len);
			max_gap = cal_max_gap(opt, qd < rd ? qd : (int)rd);
			w = max_gap < p->w ? max_gap : p->w;
			if (qd - rd < w && rd - qd < w) break;
		}
		if (i < (int)av->n) {
			for (i = k + 1; i < c->n; ++i) { /* an overlapping off-diagonal seed forces extension */
				const o_seed_t *t;
				if (srt[i] == 0) continue;
				t = &c->seeds[(uint32_t)srt[i]];
				if (t->len < s->len * .95) continue;
				if (s->qbeg <= t->qbeg && s->qbeg + s->len - t->qbeg >= s->len >> 2 && t->qbeg - s->qbeg != t->rbeg - s->rbeg) break;
				if (t->qbeg <= s->qbeg && t->qbeg + t->len - s->qbeg >= s->len >> 2 && s->qbeg - t->qbeg != s->rbeg - t->rbeg) break;
			}
			if (i == c->n) {
				srt[k] = 0;
				continue;
			}
		}

		a = reg_pushp(av);
		memset(a, 0, sizeof(o_alnreg_t));
		a->w = aw[0] = aw[1] = opt->w;
		a->score = a->truesc = -1;
		a->rid = c->rid;

		if (s->qbeg) { /* left extension on reversed sequences */
			uint8_t *rs, *qs;
			int qle, tle, gtle, gscore;
			qs = malloc(s->qbeg);
			for (i = 0; i < s->qbeg; ++i) qs[i] = query[s->qbeg - 1 - i];
			tmp = s->rbeg - rmax[0];
			rs = malloc(tmp + 1);
			for (i = 0; i < tmp; ++i) rs[i] = rseq[tmp - 1 - i];
			for (i = 0; i < MAX_BAND_TRY; ++i) {
				int prev = a->score;
				aw[0] = opt->w << i;
				a->score = o_ksw_extend2(s->qbeg, qs, (int)tmp, rs, 5, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, aw[0], opt->pen_clip5, opt->zdrop, s->len * opt->a, &qle, &tle, &gtle, &gscore, &max_off[0]);
				if (a->score == prev || max_off[0] < (aw[0] >> 1) + (aw[0] >> 2)) break;
			}
			if (gscore <= 0 || gscore <= a->score - opt->pen_clip5) {
				a->qb = s->qbeg - qle, a->rb = s->rbeg - tle;
				a->truesc = a->score;
			} else {
				a->qb = 0, a->rb = s->rbeg - gtle;
				a->truesc = gscore;
			}
			free(qs); free(rs);
		} else a->score = a->truesc = s->len * opt->a, a->qb = 0, a->rb = s->rbeg;

		if (s->qbeg + s->len != l_query) { /* right extension */
			int qle, tle, qe, re, gtle, gscore, sc0 = a->score;
			qe = s->qbeg + s->len;
			re = (int)(s->rbeg + s->len - rmax[0]);
			assert(re >= 0);
			for (i = 0; i < MAX_BAND_TRY; ++i) {
				int prev = a->score;
				aw[1] = opt->w << i;
				a->score = o_ksw_extend2(l_query - qe, query + qe, (int)(rmax[1] - rmax[0] - re), rseq + re, 5, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, aw[1], opt->pen_clip3, opt->zdrop, sc0, &qle, &tle, &gtle, &gscore, &max_off[1]);
				if (a->score == prev || max_off[1] < (aw[1] >> 1) + (aw[1] >> 2)) break;
			}
			if (gscore <= 0 || gscore <= a->score - opt->pen_clip3) {
				a->qe = qe + qle, a->re = rmax[0] + re + tle;
				a->truesc += a->score - sc0;
			} else {
				a->qe = l_query, a->re = rmax[0] + re + gtle;
				a->truesc += gscore - sc0;
			}
		} else a->qe = l_query, a->re = s->rbeg + s->len;

		for (i = 0, a->seedcov = 0; i < c->n; ++i) {
			const o_seed_t *t = &c->seeds[i];
			if (t->qbeg >= a->qb && t->qbeg + t->len <= a->qe && t->rbeg >= a->rb && t->rbeg + t->len <= a->re)
				a->seedcov += t->len;
		}
		a->w = aw[0] > aw[1] ? aw[0] : aw[1];
		a->seedlen0 = s->len;
		a->frac_rep = c->frac_rep;
	}
	free(srt); free(rseq);
}

/* ---------------- CIGAR / NM / MD (row a15; upstream bwa.c bwa_gen_cigar2) ---------------- */

static inline void str_reserve(o_str_t *s, size_t add)
{
	if (s->l + add + 1 > s->m) {
		s->m = s->l + add + 2;
		s->m += s->m >> 1;
		s->s = realloc(s->s, s->m);
	}
}
static inline void str_putc(o_str_t *s, int c) { str_reserve(s, 1); s->s[s->l++] = (char)c; s->s[s->l] = 0; }
static inline void str_putsn(o_str_t *s, const char *p, size_t n) { str_reserve(s, n); memcpy(s->s + s->l, p, n); s->l += n; s->s[s->l] = 0; }
static inline void str_putl(o_str_t *s, long c)
{
	char buf[32];
	int l = 0;
	unsigned long x = c < 0 ? -(unsigned long)c : (unsigned long)c;
	do { buf[l++] = (char)(x % 10 + '0'); x /= 10; } while (x > 0);
	if (c < 0) buf[l++] = '-';
	while (l > 0) str_putc(s, buf[--l]);
}
static inline void str_put32(o_str_t *s, int32_t v) { str_putsn(s, (const char*)&v, 4); }

static uint32_t *gen_cigar2(const int8_t mat[25], int o_del, int e_del, int o_ins, int e_ins, int w_, int64_t l_pac, const uint8_t *pac,
                            int l_query, uint8_t *query, int64_t rb, int64_t re, int *score, int *n_cigar, int *NM)
{
	uint32_t *cigar = 0;
	uint8_t tmp, *rseq;
	int i;
	int64_t rlen;
	o_str_t str;
	const char *int2base;

	if (n_cigar) *n_cigar = 0;
	if (NM) *NM = -1;
	if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return 0;
	rseq = o_bns_get_seq(l_pac, pac, rb, re, &rlen);
	if (re - rb != rlen) goto ret_gen_cigar;
	if (rb >= l_pac) { /* reverse (not complement) both: gaps end up left-aligned on the forward strand */
		for (i = 0; i < l_query >> 1; ++i)
			tmp = query[i], query[i] = query[l_query - 1 - i], query[l_query - 1 - i] = tmp;
		for (i = 0; i < rlen >> 1; ++i)
			tmp = rseq[i], rseq[i] = rseq[rlen - 1 - i], rseq[rlen - 1 - i] = tmp;
	}
	if (l_query == re - rb && w_ == 0) { /* no gap: no DP */
		if (n_cigar) {
			cigar = malloc(4);
			cigar[0] = (uint32_t)l_query << 4 | 0;
			*n_cigar = 1;
		}
		for (i = 0, *score = 0; i < l_query; ++i)
			*score += mat[rseq[i] * 5 + query[i]];
	} else {
		int w, max_gap, max_ins, max_del, min_w;
		max_ins = (int)((double)(((l_query + 1) >> 1) * mat[0] - o_ins) / e_ins + 1.);
		max_del = (int)((double)(((l_query + 1) >> 1) * mat[0] - o_del) / e_del + 1.);
		max_gap = max_ins > max_del ? max_ins : max_del;
		max_gap = max_gap > 1 ? max_gap : 1;
		w = (max_gap + abs((int)rlen - l_query) + 1) >> 1;
		w = w < w_ ? w : w_;
		min_w = abs((int)rlen - l_query) + 3;
		w = w > min_w ? w : min_w;
		*score = o_ksw_global2(l_query, query, (int)rlen, rseq, 5, mat, o_del, e_del, o_ins, e_ins, w, n_cigar, &cigar);
	}
	if (NM && n_cigar) { /* NM and MD, appended after the CIGAR words */
		int k, x, y, u, n_mm = 0, n_gap = 0;
		str.l = str.m = (size_t)*n_cigar * 4; str.s = (char*)cigar;
		int2base = rb < l_pac ? "ACGTN" : "TGCAN";
		for (k = 0, x = y = u = 0; k < *n_cigar; ++k) {
			int op, len;
			cigar = (uint32_t*)str.s;
			op  = cigar[k] & 0xf, len = cigar[k] >> 4;
			if (op == 0) {
				for (i = 0; i < len; ++i) {
					if (query[x + i] != rseq[y + i]) {
						str_putl(&str, u);
						str_putc(&str, int2base[rseq[y+i]]);
						++n_mm; u = 0;
					} else ++u;
				}
				x += len; y += len;
			} else if (op == 2) {
				if (k > 0 && k < *n_cigar - 1) { /* not for a leading / trailing D */
					str_putl(&str, u); str_putc(&str, '^');
					for (i = 0; i < len; ++i) str_putc(&str, int2base[rseq[y+i]]);
					u = 0; n_gap += len;
				}
				y += len;
			} else if (op == 1) x += len, n_gap += len;
		}
		str_putl(&str, u); str_putc(&str, 0);
		*NM = n_mm + n_gap;
		cigar = (uint32_t*)str.s;
	}
	if (rb >= l_pac)
		for (i = 0; i < l_query >> 1; ++i)
			tmp = query[i], query[i] = query[l_query - 1 - i], query[l_query - 1 - i] = tmp;
ret_gen_cigar:
	free(rseq);
	return cigar;
}

/* ---------------- region de-duplication / patching (row a13) ---------------- */

#define PATCH_MAX_R_BW 0.05f
#define PATCH_MIN_SC_RATIO 0.90f

static int patch_reg(const o_opt_t *opt, const o_bns_t *bns, const uint8_t *pac, uint8_t *query, const o_alnreg_t *a, const o_alnreg_t *b, int *_w)
{
	int w, score, q_s, r_s;
	double r;
	if (bns == 0 || pac == 0 || query == 0) return 0;
	assert(a->rid == b->rid && a->rb <= b->rb);
	if (a->rb < bns->l_pac && b->rb >= bns->l_pac) return 0;
	if (a->qb >= b->qb || a->qe >= b->qe || a->re >= b->re) return 0;
	w = (int)((a->re - b->rb) - (a->qe - b->qb));
	w = w > 0 ? w : -w;
	r = (double)(a->re - b->rb) / (b->re - a->rb) - (double)(a->qe - b->qb) / (b->qe - a->qb);
	r = r > 0. ? r : -r;
	if (a->re < b->rb || a->qe < b->qb) {
		if (w > opt->w << 1 || r >= PATCH_MAX_R_BW) return 0;
	} else if (w > opt->w << 2 || r >= PATCH_MAX_R_BW * 2) return 0;
	w += a->w + b->w;
	w = w < opt->w << 2 ? w : opt->w << 2;
	free(gen_cigar2(opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w, bns->l_pac, pac, b->qe - a->qb, query + a->qb, a->rb, b->re, &score, 0, 0));
	q_s = (int)((double)(b->qe - a->qb) / ((b->qe - b->qb) + (a->qe - a->qb)) * (b->score + a->score) + .499);
	r_s = (int)((double)(b->re - a->rb) / ((b->re - b->rb) + (a->re - a->rb)) * (b->score + a->score) + .499);
	if ((double)score / (q_s > r_s ? q_s : r_s) < PATCH_MIN_SC_RATIO) return 0;
	*_w = w;
	return score;
}

int o_sort_dedup_patch(const o_opt_t *opt, const o_bns_t *bns, const uint8_t *pac, uint8_t *query, int n, o_alnreg_t *a)
{
	int m, i, j;
	if (n <= 1) return n;
	o_introsort_ars2(n, a); /* by END position */
	for (i = 0; i < n; ++i) a[i].n_comp = 1;
	for (i = 1; i < n; ++i) {
		o_alnreg_t *p = &a[i];
		if (p->rid != a[i-1].rid || p->rb >= a[i-1].re + opt->max_chain_gap) continue;
		for (j = i - 1; j >= 0 && p->rid == a[j].rid && p->rb < a[j].re + opt->max_chain_gap; --j) {
			o_alnreg_t *q = &a[j];
			int64_t or_, oq, mr, mq;
			int score, w;
			if (q->qe == q->qb) continue;
			or_ = q->re - p->rb;
			oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
			mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
			mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
			if ((float)or_ > opt->mask_level_redun * (float)mr && (float)oq > opt->mask_level_redun * (float)mq) {
				if (p->score < q->score) {
					p->qe = p->qb;
					break;
				} else q->qe = q->qb;
			} else if (q->rb < p->rb && (score = patch_reg(opt, bns, pac, query, q, p, &w)) > 0) {
				p->n_comp += q->n_comp + 1;
				p->seedcov = p->seedcov > q->seedcov ? p->seedcov : q->seedcov;
				p->sub = p->sub > q->sub ? p->sub : q->sub;
				p->csub = p->csub > q->csub ? p->csub : q->csub;
				p->qb = q->qb, p->rb = q->rb;
				p->truesc = p->score = score;
				p->w = w;
				q->qb = q->qe;
			}
		}
	}
	for (i = 0, m = 0; i < n; ++i)
		if (a[i].qe > a[i].qb) {
			if (m != i) a[m++] = a[i];
			else ++m;
		}
	n = m;
	o_introsort_ars(n, a);
	for (i = 1; i < n; ++i)
		if (a[i].score == a[i-1].score && a[i].rb == a[i-1].rb && a[i].qb == a[i-1].qb)
			a[i].qe = a[i].qb;
	for (i = 1, m = 1; i < n; ++i)
		if (a[i].qe > a[i].qb) {
			if (m != i) a[m++] = a[i];
			else ++m;
		}
	return m;
}

/* ---------------- per-read core (mem_align1_core) ---------------- */

void o_align1_core(const o_opt_t *opt, const o_idx_t *idx, int l_seq, char *seq, o_alnreg_v *regs)
{
	int i;
	o_chain_v chn;
	for (i = 0; i < l_seq; ++i) /* ASCII -> 0..4 in place, as upstream does to the caller's buffer */
		seq[i] = seq[i] < 4 ? seq[i] : nt4((unsigned char)seq[i]);
	chn = mem_chain(opt, idx, l_seq, (uint8_t*)seq);
	chn.n = chain_flt(opt, (int)chn.n, chn.a);
	flt_chained_seeds(opt, &idx->bns, idx->pac, l_seq, (uint8_t*)seq, (int)chn.n, chn.a);
	regs->n = regs->m = 0; regs->a = 0;
	for (i = 0; i < (int)chn.n; ++i) {
		chain2aln(opt, &idx->bns, idx->pac, l_seq, (uint8_t*)seq, &chn.a[i], regs);
		free(chn.a[i].seeds);
	}
	free(chn.a);
	regs->n = o_sort_dedup_patch(opt, &idx->bns, idx->pac, (uint8_t*)seq, (int)regs->n, regs->a);
	for (i = 0; i < (int)regs->n; ++i) {
		o_alnreg_t *p = &regs->a[i];
		if (p->rid >= 0 && idx->bns.anns[p->rid].is_alt) p->is_alt = 1;
	}
}

/* ---------------- primary marking (row a14) ---------------- */

static void mark_primary_core(const o_opt_t *opt, int n, o_alnreg_t *a, int *z, int *nz)
{
	int i, k, tmp;
	tmp = opt->a + opt->b;
	tmp = opt->o_del + opt->e_del > tmp ? opt->o_del + opt->e_del : tmp;
	tmp = opt->o_ins + opt->e_ins > tmp ? opt->o_ins + opt->e_ins : tmp;
	*nz = 0;
	z[(*nz)++] = 0;
	for (i = 1; i < n; ++i) {
		for (k = 0; k < *nz; ++k) {
			int j = z[k];
			int b_max = a[j].qb > a[i].qb ? a[j].qb : a[i].qb;
			int e_min = a[j].qe < a[i].qe ? a[j].qe : a[i].qe;
			if (e_min > b_max) {
				int min_l = a[i].qe - a[i].qb < a[j].qe - a[j].qb ? a[i].qe - a[i].qb : a[j].qe - a[j].qb;
				if ((float)(e_min - b_max) >= (float)min_l * opt->mask_level) {
					if (a[j].sub == 0) a[j].sub = a[i].score;
					if (a[j].score - a[i].score <= tmp && (a[j].is_alt || !a[i].is_alt))
						++a[j].sub_n;
					break;
				}
			}
		}
		if (k == *nz) z[(*nz)++] = i;
		else a[i].secondary = z[k];
	}
}

int o_mark_primary_se(const o_opt_t *opt, int n, o_alnreg_t *a, int64_t id)
{
	int i, n_pri, *z, nz;
	if (n == 0) return 0;
	z = malloc(sizeof(int) * n);
	for (i = n_pri = 0; i < n; ++i) {
		a[i].sub = a[i].alt_sc = 0, a[i].secondary = a[i].secondary_all = -1, a[i].hash = o_hash_64(id + i);
		if (!a[i].is_alt) ++n_pri;
	}
	o_introsort_ars_hash(n, a);
	mark_primary_core(opt, n, a, z, &nz);
	for (i = 0; i < n; ++i) {
		o_alnreg_t *p = &a[i];
		p->secondary_all = i;
		if (!p->is_alt && p->secondary >= 0 && a[p->secondary].is_alt)
			p->alt_sc = a[p->secondary].score;
	}
	if (n_pri >= 0 && n_pri < n) {
		if (n_pri > 0) o_introsort_ars_hash2(n, a);
		for (i = 0; i < n; ++i) z[a[i].secondary_all] = i;
		for (i = 0; i < n; ++i) {
			if (a[i].secondary >= 0) {
				a[i].secondary_all = z[a[i].secondary];
				if (a[i].is_alt) a[i].secondary = INT_MAX;
			} else a[i].secondary_all = -1;
		}
		if (n_pri > 0) {
			for (i = 0; i < n_pri; ++i) a[i].sub = 0, a[i].secondary = -1;
			mark_primary_core(opt, n_pri, a, z, &nz);
		}
	} else {
		for (i = 0; i < n; ++i) a[i].secondary_all = a[i].secondary;
	}
	free(z);
	return n_pri;
}

void o_reorder_primary5(int T, o_alnreg_v *a)
{
	int k, n_pri = 0, left_st = INT_MAX, left_k = -1;
	o_alnreg_t t;
	for (k = 0; k < (int)a->n; ++k)
		if (a->a[k].secondary < 0 && !a->a[k].is_alt && a->a[k].score >= T) ++n_pri;
	if (n_pri <= 1) return;
	for (k = 0; k < (int)a->n; ++k) {
		o_alnreg_t *p = &a->a[k];
		if (p->secondary >= 0 || p->is_alt || p->score < T) continue;
		if (p->qb < left_st) left_st = p->qb, left_k = k;
	}
	if (left_k == 0) return;
	t = a->a[0], a->a[0] = a->a[left_k], a->a[left_k] = t;
	for (k = 1; k < (int)a->n; ++k) {
		o_alnreg_t *p = &a->a[k];
		if (p->secondary == 0) p->secondary = left_k;
		else if (p->secondary == left_k) p->secondary = 0;
		if (p->secondary_all == 0) p->secondary_all = left_k;
		else if (p->secondary_all == left_k) p->secondary_all = 0;
	}
}

/* ---------------- MAPQ (row a16) ---------------- */

#define MEM_MAPQ_COEF 30.0

int o_approx_mapq_se(const o_opt_t *opt, const o_alnreg_t *a)
{
	int mapq, l, sub = a->sub ? a->sub : opt->min_seed_len * opt->a;
	double identity;
	sub = a->csub > sub ? a->csub : sub;
	if (sub >= a->score) return 0;
	l = a->qe - a->qb > a->re - a->rb ? a->qe - a->qb : (int)(a->re - a->rb);
	identity = 1. - (double)(l * opt->a - a->score) / (opt->a + opt->b) / l;
	if (a->score == 0) {
		mapq = 0;
	} else if (opt->mapQ_coef_len > 0) {
		double tmp;
		tmp = l < opt->mapQ_coef_len ? 1. : opt->mapQ_coef_fac / log(l);
		tmp *= identity * identity;
		mapq = (int)(6.02 * (a->score - sub) / opt->a * tmp * tmp + .499);
	} else {
		mapq = (int)(MEM_MAPQ_COEF * (1. - (double)sub / a->score) * log(a->seedcov) + .499);
		mapq = identity < 0.95 ? (int)(mapq * identity * identity + .499) : mapq;
	}
	if (a->sub_n > 0) mapq -= (int)(4.343 * log(a->sub_n + 1) + .499);
	if (mapq > 60) mapq = 60;
	if (mapq < 0) mapq = 0;
	mapq = (int)(mapq * (1. - a->frac_rep) + .499);
	return mapq;
}

/* ---------------- region -> alignment record (row a15) ---------------- */

static inline int infer_bw(int l1, int l2, int score, int a, int q, int r)
{
	int w;
	if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
	w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
	if (w < abs(l1 - l2)) w = abs(l1 - l2);
	return w;
}

o_aln_t o_reg2aln(const o_opt_t *opt, const o_bns_t *bns, const uint8_t *pac, int l_query, const char *query_, const o_alnreg_t *ar)
{
	o_aln_t a;
	int i, w2, tmp, qb, qe, NM, score, is_rev, last_sc = -(1 << 30), l_MD;
	int64_t pos, rb, re;
	uint8_t *query;

	memset(&a, 0, sizeof(o_aln_t));
	if (ar == 0 || ar->rb < 0 || ar->re < 0) {
		a.rid = -1; a.pos = -1; a.flag |= 0x4;
		return a;
	}
	qb = ar->qb, qe = ar->qe;
	rb = ar->rb, re = ar->re;
	query = malloc(l_query + 1);
	for (i = 0; i < l_query; ++i)
		query[i] = query_[i] < 5 ? query_[i] : nt4((unsigned char)query_[i]);
	a.mapq = ar->secondary < 0 ? o_approx_mapq_se(opt, ar) : 0;
	if (ar->secondary >= 0) a.flag |= 0x100;
	tmp = infer_bw(qe - qb, (int)(re - rb), ar->truesc, opt->a, opt->o_del, opt->e_del);
	w2  = infer_bw(qe - qb, (int)(re - rb), ar->truesc, opt->a, opt->o_ins, opt->e_ins);
	w2 = w2 > tmp ? w2 : tmp;
	if (w2 > opt->w) w2 = w2 < ar->w ? w2 : ar->w;
	i = 0; a.cigar = 0;
	do {
		free(a.cigar);
		w2 = w2 < opt->w << 2 ? w2 : opt->w << 2;
		a.cigar = gen_cigar2(opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w2, bns->l_pac, pac, qe - qb, &query[qb], rb, re, &score, &a.n_cigar, &NM);
		if (score == last_sc || w2 == opt->w << 2) break;
		last_sc = score;
		w2 <<= 1;
	} while (++i < 3 && score < ar->truesc - opt->a);
	l_MD = (int)strlen((char*)(a.cigar + a.n_cigar)) + 1;
	a.NM = NM;
	pos = o_bns_depos(bns, rb < bns->l_pac ? rb : re - 1, &is_rev);
	a.is_rev = is_rev;
	if (a.n_cigar > 0) { /* squeeze out a leading or trailing deletion */
		if ((a.cigar[0] & 0xf) == 2) {
			pos += a.cigar[0] >> 4;
			--a.n_cigar;
			memmove(a.cigar, a.cigar + 1, a.n_cigar * 4 + l_MD);
		} else if ((a.cigar[a.n_cigar-1] & 0xf) == 2) {
			--a.n_cigar;
			memmove(a.cigar + a.n_cigar, a.cigar + a.n_cigar + 1, l_MD);
		}
	}
	if (qb != 0 || qe != l_query) { /* clipping */
		int clip5, clip3;
		clip5 = is_rev ? l_query - qe : qb;
		clip3 = is_rev ? qb : l_query - qe;
		a.cigar = realloc(a.cigar, 4 * (a.n_cigar + 2) + l_MD);
		if (clip5) {
			memmove(a.cigar + 1, a.cigar, a.n_cigar * 4 + l_MD);
			a.cigar[0] = (uint32_t)clip5 << 4 | 3;
			++a.n_cigar;
		}
		if (clip3) {
			memmove(a.cigar + a.n_cigar + 1, a.cigar + a.n_cigar, l_MD);
			a.cigar[a.n_cigar++] = (uint32_t)clip3 << 4 | 3;
		}
	}
	a.rid = o_bns_pos2rid(bns, pos);
	assert(a.rid == ar->rid);
	a.pos = pos - bns->anns[a.rid].offset;
	a.score = ar->score; a.sub = ar->sub > ar->csub ? ar->sub : ar->csub;
	a.is_alt = ar->is_alt; a.alt_sc = ar->alt_sc;
	free(query);
	return a;
}

/* ---------------- XA (row a17; upstream bwamem_extra.c mem_gen_alt) ---------------- */

static inline int get_pri_idx(double XA_drop_ratio, const o_alnreg_t *a, int i)
{
	int k = a[i].secondary_all;
	if (k >= 0 && a[i].score >= a[k].score * XA_drop_ratio) return k;
	return -1;
}

char **o_gen_alt(const o_opt_t *opt, const o_bns_t *bns, const uint8_t *pac, const o_alnreg_v *a, int l_query, const char *query)
{
	int i, k, r, *cnt, tot;
	o_str_t *aln = 0, str = { 0, 0, 0 };
	char **XA = 0, *has_alt;

	cnt = calloc(a->n + 1, sizeof(int));
	has_alt = calloc(a->n + 1, 1);
	for (i = 0, tot = 0; i < (int)a->n; ++i) {
		r = get_pri_idx(opt->XA_drop_ratio, a->a, i);
		if (r >= 0) {
			++cnt[r], ++tot;
			if (a->a[i].is_alt) has_alt[r] = 1;
		}
	}
	if (tot == 0) goto end_gen_alt;
	aln = calloc(a->n, sizeof(o_str_t));
	for (i = 0; i < (int)a->n; ++i) {
		o_aln_t t;
		if ((r = get_pri_idx(opt->XA_drop_ratio, a->a, i)) < 0) continue;
		if (cnt[r] > opt->max_XA_hits_alt || (!has_alt[r] && cnt[r] > opt->max_XA_hits)) continue;
		t = o_reg2aln(opt, bns, pac, l_query, query, &a->a[i]);
		str.l = 0;
		str_putsn(&str, bns->anns[t.rid].name, strlen(bns->anns[t.rid].name));
		str_putc(&str, ','); str_putc(&str, "+-"[t.is_rev]); str_putl(&str, t.pos + 1);
		str_putc(&str, ',');
		for (k = 0; k < t.n_cigar; ++k) {
			str_putl(&str, t.cigar[k] >> 4);
			str_putc(&str, "MIDSHN"[t.cigar[k] & 0xf]);
		}
		str_putc(&str, ','); str_putl(&str, t.NM);
		str_putc(&str, ';');
		free(t.cigar);
		str_putsn(&aln[r], str.s, str.l);
	}
	XA = calloc(a->n, sizeof(char*));
	for (k = 0; k < (int)a->n; ++k) XA[k] = aln[k].s;
end_gen_alt:
	free(has_alt); free(cnt); free(aln); free(str.s);
	return XA;
}

/* ---------------- record writer: mem_aln2sam prologue + the reference's fmt_BAMish ---------------- */

static inline int cigar_ref_len(int n_cigar, const uint32_t *cigar)  /* jnibwa.c:30-41 */
{
	int i, len = 0;
	for (i = 0; i < n_cigar; ++i) {
		int op = cigar[i] & 0xf;
		if (!op || op == 2) len += cigar[i] >> 4;
	}
	return len;
}

void o_aln2out(const o_opt_t *opt, const o_bns_t *bns, o_str_t *str, o_read_t *s, int n, const o_aln_t *list, int which, const o_aln_t *m_)
{
	o_aln_t ptmp = list[which], *p = &ptmp, mtmp, *m = 0;
	int32_t flag_mapQ;
	(void)opt; (void)bns; (void)s;
	if (m_) mtmp = *m_, m = &mtmp;
	/* flag set-up done by upstream before the formatting hook is entered */
	p->flag |= m ? 0x1 : 0;
	p->flag |= p->rid < 0 ? 0x4 : 0;
	p->flag |= m && m->rid < 0 ? 0x8 : 0;
	if (p->rid < 0 && m && m->rid >= 0)
		p->rid = m->rid, p->pos = m->pos, p->is_rev = m->is_rev, p->n_cigar = 0;
	if (m && m->rid < 0 && p->rid >= 0)
		m->rid = p->rid, m->pos = p->pos, m->is_rev = p->is_rev, m->n_cigar = 0;
	p->flag |= p->is_rev ? 0x10 : 0;
	p->flag |= m && m->is_rev ? 0x20 : 0;
	/* fmt_BAMish, jnibwa.c:43-97 */
	if (!which) str_put32(str, n);
	flag_mapQ = p->flag;
	if (p->flag & 0x10000) flag_mapQ |= 0x100;
	flag_mapQ = (int32_t)((uint32_t)flag_mapQ << 16) | (p->mapq & 0xff);
	str_put32(str, flag_mapQ);
	if (!(p->flag & 0x4)) {
		int i, nMD, nXA;
		const char *md = (const char*)(p->cigar + p->n_cigar);
		str_put32(str, p->rid);
		str_put32(str, (int32_t)p->pos);
		str_put32(str, p->NM);
		str_put32(str, p->score);
		str_put32(str, p->sub);
		str_put32(str, p->n_cigar);
		for (i = 0; i < p->n_cigar; ++i) {
			uint32_t lenOp = p->cigar[i];
			if ((lenOp & 0xf) > 2) ++lenOp; /* MIDSH -> BAM MIDNSH */
			str_put32(str, (int32_t)lenOp);
		}
		nMD = p->n_cigar ? (int)strlen(md) : 0;
		str_put32(str, nMD);
		if (nMD) {
			char pad[4] = { 0, 0, 0, 0 };
			str_putsn(str, md, nMD);
			str_putsn(str, pad, ((nMD + 3) & ~3) - nMD);
		}
		nXA = p->XA ? (int)strlen(p->XA) : 0;
		str_put32(str, nXA);
		if (nXA) {
			char pad[4] = { 0, 0, 0, 0 };
			str_putsn(str, p->XA, nXA);
			str_putsn(str, pad, ((nXA + 3) & ~3) - nXA);
		}
	}
	if ((p->flag & 0x9) == 1) {
		str_put32(str, m->rid);
		str_put32(str, (int32_t)m->pos);
		if ((p->flag & 0x4) || p->rid != m->rid) str_put32(str, 0);
		else { /* jnibwa.c:82-95 */
			long p0 = p->pos, m0 = m->pos;
			if (p->is_rev) p0 += cigar_ref_len(p->n_cigar, p->cigar) - 1;
			if (m->is_rev) m0 += cigar_ref_len(m->n_cigar, m->cigar) - 1;
			str_put32(str, (int32_t)(m0 - p0 + (p0 > m0 ? -1 : p0 < m0 ? 1 : 0)));
		}
	}
}

/* ---------------- record selection (row a17; mem_reg2sam) ---------------- */

void o_reg2sam(const o_opt_t *opt, const o_bns_t *bns, const uint8_t *pac, o_read_t *s, o_alnreg_v *a, int extra_flag, const o_aln_t *m)
{
	o_str_t str = { 0, 0, 0 };
	o_aln_t *aa = 0;
	int n_aa = 0, k, l;
	char **XA = 0;

	if (!(opt->flag & O_F_ALL))
		XA = o_gen_alt(opt, bns, pac, a, s->l_seq, s->seq);
	aa = malloc(sizeof(o_aln_t) * (a->n + 1));
	for (k = l = 0; k < (int)a->n; ++k) {
		o_alnreg_t *p = &a->a[k];
		o_aln_t *q;
		if (p->score < opt->T) continue;
		if (p->secondary >= 0 && (p->is_alt || !(opt->flag & O_F_ALL))) continue;
		if (p->secondary >= 0 && p->secondary < INT_MAX && (float)p->score < (float)a->a[p->secondary].score * opt->drop_ratio) continue;
		q = &aa[n_aa++];
		*q = o_reg2aln(opt, bns, pac, s->l_seq, s->seq, p);
		q->XA = XA ? XA[k] : 0;
		q->flag |= extra_flag;
		if (p->secondary >= 0) q->sub = -1;
		if (l && p->secondary < 0)
			q->flag |= (opt->flag & O_F_NO_MULTI) ? 0x10000 : 0x800;
		if (l && !p->is_alt && q->mapq > aa[0].mapq) q->mapq = aa[0].mapq;
		++l;
	}
	if (n_aa == 0) {
		o_aln_t t = o_reg2aln(opt, bns, pac, s->l_seq, s->seq, 0);
		t.flag |= extra_flag;
		o_aln2out(opt, bns, &str, s, 1, &t, 0, m);
	} else {
		for (k = 0; k < n_aa; ++k) o_aln2out(opt, bns, &str, s, n_aa, aa, k, m);
		for (k = 0; k < n_aa; ++k) free(aa[k].cigar);
	}
	free(aa);
	s->out = str;
	if (XA) {
		for (k = 0; k < (int)a->n; ++k) free(XA[k]);
		free(XA);
	}
}
