/*
 * o_pair.c -- CPU ORACLE (test infrastructure): paired-end path.
 *
 * Restates upstream lh3/bwa@cb950614 bwamem_pair.c (mem_pestat, mem_matesw,
 * mem_pair, mem_sam_pe), reached from the reference through jnibwa.c:214 when
 * MEM_F_PE is set (BwaMemAligner.java:73); the pestat hand-over follows
 * ...BwaMemIndex.c:21-40 (only orientation slot 1 = FR can be supplied).
 * Pinned by BwaMemIndexTest.testPair (flags 0x61/0x63/0x91/0x93, mate POS, TLEN).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>
#include "bwa_oracle.h"
#include "o_internal.h"

void o_sort_u64(size_t n, uint64_t *a);

typedef struct { uint64_t x, y; } pair64_t;
#define pair64_lt(a, b) ((a).x < (b).x || ((a).x == (b).x && (a).y < (b).y))
O_SORT_DECL(p128, pair64_t, pair64_lt)

#define MIN_RATIO     0.8
#define MIN_DIR_CNT   10
#define MIN_DIR_RATIO 0.05
#define OUTLIER_BOUND 2.0
#define MAPPING_BOUND 3.0
#define MAX_STDDEV    4.0

static inline int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist)
{
	int64_t p2;
	int r1 = (b1 >= l_pac), r2 = (b2 >= l_pac);
	p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2; /* read 2 on the strand of read 1 */
	*dist = p2 > b1 ? p2 - b1 : b1 - p2;
	return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

static int cal_sub(const o_opt_t *opt, const o_alnreg_v *r)
{
	int j;
	for (j = 1; j < (int)r->n; ++j) {
		int b_max = r->a[j].qb > r->a[0].qb ? r->a[j].qb : r->a[0].qb;
		int e_min = r->a[j].qe < r->a[0].qe ? r->a[j].qe : r->a[0].qe;
		if (e_min > b_max) {
			int min_l = r->a[j].qe - r->a[j].qb < r->a[0].qe - r->a[0].qb ? r->a[j].qe - r->a[j].qb : r->a[0].qe - r->a[0].qb;
			if ((float)(e_min - b_max) >= (float)min_l * opt->mask_level) break;
		}
	}
	return j < (int)r->n ? r->a[j].score : opt->min_seed_len * opt->a;
}

void o_pestat(const o_opt_t *opt, int64_t l_pac, int n, const o_alnreg_v *regs, o_pestat_t pes[4])
{
	int i, d, max;
	struct { size_t n, m; uint64_t *a; } isize[4];
	memset(pes, 0, 4 * sizeof(o_pestat_t));
	memset(isize, 0, sizeof isize);
	for (i = 0; i < n >> 1; ++i) {
		int dir;
		int64_t is;
		const o_alnreg_v *r[2];
		r[0] = &regs[i << 1 | 0];
		r[1] = &regs[i << 1 | 1];
		if (r[0]->n == 0 || r[1]->n == 0) continue;
		if (cal_sub(opt, r[0]) > MIN_RATIO * r[0]->a[0].score) continue;
		if (cal_sub(opt, r[1]) > MIN_RATIO * r[1]->a[0].score) continue;
		if (r[0]->a[0].rid != r[1]->a[0].rid) continue;
		dir = infer_dir(l_pac, r[0]->a[0].rb, r[1]->a[0].rb, &is);
		if (is && is <= opt->max_ins) {
			if (isize[dir].n == isize[dir].m) {
				isize[dir].m = isize[dir].m ? isize[dir].m << 1 : 2;
				isize[dir].a = realloc(isize[dir].a, 8 * isize[dir].m);
			}
			isize[dir].a[isize[dir].n++] = is;
		}
	}
	for (d = 0; d < 4; ++d) {
		o_pestat_t *r = &pes[d];
		size_t qn = isize[d].n;
		uint64_t *qa = isize[d].a;
		int p25, p50, p75, x;
		if (qn < MIN_DIR_CNT) {
			r->failed = 1;
			free(qa);
			continue;
		}
		o_sort_u64(qn, qa);
		p25 = (int)qa[(int)(.25 * qn + .499)];
		p50 = (int)qa[(int)(.50 * qn + .499)];
		p75 = (int)qa[(int)(.75 * qn + .499)];
		(void)p50;
		r->low  = (int)(p25 - OUTLIER_BOUND * (p75 - p25) + .499);
		if (r->low < 1) r->low = 1;
		r->high = (int)(p75 + OUTLIER_BOUND * (p75 - p25) + .499);
		for (i = x = 0, r->avg = 0; i < (int)qn; ++i)
			if (qa[i] >= (uint64_t)r->low && qa[i] <= (uint64_t)r->high)
				r->avg += qa[i], ++x;
		r->avg /= x;
		for (i = 0, r->std = 0; i < (int)qn; ++i)
			if (qa[i] >= (uint64_t)r->low && qa[i] <= (uint64_t)r->high)
				r->std += (qa[i] - r->avg) * (qa[i] - r->avg);
		r->std = sqrt(r->std / x);
		r->low  = (int)(p25 - MAPPING_BOUND * (p75 - p25) + .499);
		r->high = (int)(p75 + MAPPING_BOUND * (p75 - p25) + .499);
		if (r->low  > r->avg - MAX_STDDEV * r->std) r->low  = (int)(r->avg - MAX_STDDEV * r->std + .499);
		if (r->high < r->avg + MAX_STDDEV * r->std) r->high = (int)(r->avg + MAX_STDDEV * r->std + .499);
		if (r->low < 1) r->low = 1;
		free(qa);
	}
	for (d = 0, max = 0; d < 4; ++d)
		max = max > (int)isize[d].n ? max : (int)isize[d].n;
	for (d = 0; d < 4; ++d)
		if (pes[d].failed == 0 && isize[d].n < max * MIN_DIR_RATIO)
			pes[d].failed = 1;
}

static int matesw(const o_opt_t *opt, const o_bns_t *bns, const uint8_t *pac, const o_pestat_t pes[4], const o_alnreg_t *a, int l_ms, const uint8_t *ms, o_alnreg_v *ma)
{
	int64_t l_pac = bns->l_pac;
	int i, r, skip[4], n = 0, rid = -1;
	for (r = 0; r < 4; ++r) skip[r] = pes[r].failed ? 1 : 0;
	for (i = 0; i < (int)ma->n; ++i) { /* orientations already explained by an existing hit */
		int64_t dist;
		r = infer_dir(l_pac, a->rb, ma->a[i].rb, &dist);
		if (dist >= pes[r].low && dist <= pes[r].high) skip[r] = 1;
	}
	if (skip[0] + skip[1] + skip[2] + skip[3] == 4) return 0;
	for (r = 0; r < 4; ++r) {
		int is_rev, is_larger;
		uint8_t *seq, *rev = 0, *ref = 0;
		int64_t rb, re;
		if (skip[r]) continue;
		is_rev = (r >> 1 != (r & 1));
		is_larger = !(r >> 1);
		seq = malloc(l_ms + 1);
		if (is_rev) {
			for (i = 0; i < l_ms; ++i) seq[l_ms - 1 - i] = ms[i] < 4 ? 3 - ms[i] : 4;
			rev = seq;
		} else memcpy(seq, ms, l_ms);
		if (!is_rev) {
			rb = is_larger ? a->rb + pes[r].low : a->rb - pes[r].high;
			re = (is_larger ? a->rb + pes[r].high : a->rb - pes[r].low) + l_ms;
		} else {
			rb = (is_larger ? a->rb + pes[r].low : a->rb - pes[r].high) - l_ms;
			re = is_larger ? a->rb + pes[r].high : a->rb - pes[r].low;
		}
		if (rb < 0) rb = 0;
		if (re > l_pac << 1) re = l_pac << 1;
		if (rb < re) ref = o_bns_fetch_seq(bns, pac, &rb, (rb + re) >> 1, &re, &rid);
		if (a->rid == rid && re - rb >= opt->min_seed_len) {
			o_kswr_t aln;
			o_alnreg_t b;
			int tmp, xtra = O_KSW_XSUBO | O_KSW_XSTART | (l_ms * opt->a < 250 ? O_KSW_XBYTE : 0) | (opt->min_seed_len * opt->a);
			aln = o_ksw_align2(l_ms, seq, (int)(re - rb), ref, 5, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, xtra);
			memset(&b, 0, sizeof(o_alnreg_t));
			if (aln.score >= opt->min_seed_len && aln.qb >= 0) {
				b.rid = a->rid;
				b.is_alt = a->is_alt;
				b.qb = is_rev ? l_ms - (aln.qe + 1) : aln.qb;
				b.qe = is_rev ? l_ms - aln.qb : aln.qe + 1;
				b.rb = is_rev ? (l_pac << 1) - (rb + aln.te + 1) : rb + aln.tb;
				b.re = is_rev ? (l_pac << 1) - (rb + aln.tb) : rb + aln.te + 1;
				b.score = aln.score;
				b.csub = aln.score2;
				b.secondary = -1;
				b.seedcov = (int)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
				if (ma->n == ma->m) { ma->m = ma->m ? ma->m << 1 : 2; ma->a = realloc(ma->a, ma->m * sizeof(o_alnreg_t)); }
				ma->n++;
				for (i = 0; i < (int)ma->n - 1; ++i) /* keep ma sorted by score */
					if (ma->a[i].score < b.score) break;
				tmp = i;
				for (i = (int)ma->n - 1; i > tmp; --i) ma->a[i] = ma->a[i-1];
				ma->a[i] = b;
			}
			++n;
		}
		if (n) ma->n = o_sort_dedup_patch(opt, 0, 0, 0, (int)ma->n, ma->a);
		(void)rev;
		free(seq);
		free(ref);
	}
	return n;
}

static int mem_pair(const o_opt_t *opt, const o_bns_t *bns, const o_pestat_t pes[4], o_alnreg_v a[2], int id, int *sub, int *n_sub, int z[2], int n_pri[2])
{
	struct { size_t n, m; pair64_t *a; } v = { 0, 0, 0 }, u = { 0, 0, 0 };
	int r, i, k, y[4], ret;
	int64_t l_pac = bns->l_pac;
	for (r = 0; r < 2; ++r) {
		for (i = 0; i < n_pri[r]; ++i) {
			pair64_t key;
			o_alnreg_t *e = &a[r].a[i];
			key.x = e->rb < l_pac ? e->rb : (l_pac << 1) - 1 - e->rb;
			key.x = (uint64_t)e->rid << 32 | (key.x - bns->anns[e->rid].offset);
			key.y = (uint64_t)e->score << 32 | i << 2 | (e->rb >= l_pac) << 1 | r;
			if (v.n == v.m) { v.m = v.m ? v.m << 1 : 2; v.a = realloc(v.a, v.m * sizeof(pair64_t)); }
			v.a[v.n++] = key;
		}
	}
	o_introsort_p128(v.n, v.a);
	y[0] = y[1] = y[2] = y[3] = -1;
	for (i = 0; i < (int)v.n; ++i) {
		for (r = 0; r < 2; ++r) {
			int dir = r << 1 | (v.a[i].y >> 1 & 1), which;
			if (pes[dir].failed) continue;
			which = r << 1 | ((v.a[i].y & 1) ^ 1);
			if (y[which] < 0) continue;
			for (k = y[which]; k >= 0; --k) {
				int64_t dist;
				int q;
				double ns;
				pair64_t *p;
				if ((int)(v.a[k].y & 3) != which) continue;
				dist = (int64_t)v.a[i].x - v.a[k].x;
				if (dist > pes[dir].high) break;
				if (dist < pes[dir].low)  continue;
				ns = (dist - pes[dir].avg) / pes[dir].std;
				q = (int)((v.a[i].y >> 32) + (v.a[k].y >> 32) + .721 * log(2. * erfc(fabs(ns) * M_SQRT1_2)) * opt->a + .499);
				if (q < 0) q = 0;
				if (u.n == u.m) { u.m = u.m ? u.m << 1 : 2; u.a = realloc(u.a, u.m * sizeof(pair64_t)); }
				p = &u.a[u.n++];
				p->y = (uint64_t)k << 32 | i;
				p->x = (uint64_t)q << 32 | (o_hash_64(p->y ^ id << 8) & 0xffffffffU);
			}
		}
		y[v.a[i].y & 3] = i;
	}
	if (u.n) {
		int tmp = opt->a + opt->b;
		tmp = tmp > opt->o_del + opt->e_del ? tmp : opt->o_del + opt->e_del;
		tmp = tmp > opt->o_ins + opt->e_ins ? tmp : opt->o_ins + opt->e_ins;
		o_introsort_p128(u.n, u.a);
		i = u.a[u.n-1].y >> 32; k = u.a[u.n-1].y << 32 >> 32;
		z[v.a[i].y & 1] = v.a[i].y << 32 >> 34;
		z[v.a[k].y & 1] = v.a[k].y << 32 >> 34;
		ret = u.a[u.n-1].x >> 32;
		*sub = u.n > 1 ? u.a[u.n-2].x >> 32 : 0;
		for (i = (long)u.n - 2, *n_sub = 0; i >= 0; --i)
			if (*sub - (int)(u.a[i].x >> 32) <= tmp) ++*n_sub;
	} else ret = 0, *sub = 0, *n_sub = 0;
	free(u.a); free(v.a);
	return ret;
}

#define raw_mapq(diff, a) ((int)(6.02 * (diff) / (a) + .499))

int o_sam_pe(const o_opt_t *opt, const o_bns_t *bns, const uint8_t *pac, const o_pestat_t pes[4], uint64_t id, o_read_t s[2], o_alnreg_v a[2])
{
	int n = 0, i, j, z[2], o, subo, n_sub, extra_flag = 1, n_pri[2], n_aa[2];
	o_str_t str = { 0, 0, 0 };
	o_aln_t h[2], g[2], aa[2][2];

	memset(h, 0, sizeof(o_aln_t) * 2);
	memset(g, 0, sizeof(o_aln_t) * 2);
	n_aa[0] = n_aa[1] = 0;
	if (!(opt->flag & O_F_NO_RESCUE)) { /* mate rescue by local SW */
		o_alnreg_v b[2];
		memset(b, 0, sizeof b);
		for (i = 0; i < 2; ++i)
			for (j = 0; j < (int)a[i].n; ++j)
				if (a[i].a[j].score >= a[i].a[0].score - opt->pen_unpaired) {
					if (b[i].n == b[i].m) { b[i].m = b[i].m ? b[i].m << 1 : 2; b[i].a = realloc(b[i].a, b[i].m * sizeof(o_alnreg_t)); }
					b[i].a[b[i].n++] = a[i].a[j];
				}
		for (i = 0; i < 2; ++i)
			for (j = 0; j < (int)b[i].n && j < opt->max_matesw; ++j)
				n += matesw(opt, bns, pac, pes, &b[i].a[j], s[!i].l_seq, (uint8_t*)s[!i].seq, &a[!i]);
		free(b[0].a); free(b[1].a);
	}
	n_pri[0] = o_mark_primary_se(opt, (int)a[0].n, a[0].a, id << 1 | 0);
	n_pri[1] = o_mark_primary_se(opt, (int)a[1].n, a[1].a, id << 1 | 1);
	if (opt->flag & O_F_PRIMARY5) {
		o_reorder_primary5(opt->T, &a[0]);
		o_reorder_primary5(opt->T, &a[1]);
	}
	if (opt->flag & O_F_NOPAIRING) goto no_pairing;
	if (n_pri[0] && n_pri[1] && (o = mem_pair(opt, bns, pes, a, (int)id, &subo, &n_sub, z, n_pri)) > 0) {
		int is_multi[2], q_pe, score_un, q_se[2];
		char **XA[2];
		for (i = 0; i < 2; ++i) {
			for (j = 1; j < n_pri[i]; ++j)
				if (a[i].a[j].secondary < 0 && a[i].a[j].score >= opt->T) break;
			is_multi[i] = j < n_pri[i] ? 1 : 0;
		}
		if (is_multi[0] || is_multi[1]) goto no_pairing;
		score_un = a[0].a[0].score + a[1].a[0].score - opt->pen_unpaired;
		subo = subo > score_un ? subo : score_un;
		q_pe = raw_mapq(o - subo, opt->a);
		if (n_sub > 0) q_pe -= (int)(4.343 * log(n_sub + 1) + .499);
		if (q_pe < 0) q_pe = 0;
		if (q_pe > 60) q_pe = 60;
		q_pe = (int)(q_pe * (1. - .5 * (a[0].a[0].frac_rep + a[1].a[0].frac_rep)) + .499);
		if (o > score_un) { /* the paired alignment is preferred */
			o_alnreg_t *c[2];
			c[0] = &a[0].a[z[0]]; c[1] = &a[1].a[z[1]];
			for (i = 0; i < 2; ++i) {
				if (c[i]->secondary >= 0)
					c[i]->sub = a[i].a[c[i]->secondary].score, c[i]->secondary = -2;
				q_se[i] = o_approx_mapq_se(opt, c[i]);
			}
			q_se[0] = q_se[0] > q_pe ? q_se[0] : q_pe < q_se[0] + 40 ? q_pe : q_se[0] + 40;
			q_se[1] = q_se[1] > q_pe ? q_se[1] : q_pe < q_se[1] + 40 ? q_pe : q_se[1] + 40;
			extra_flag |= 2;
			q_se[0] = q_se[0] < raw_mapq(c[0]->score - c[0]->csub, opt->a) ? q_se[0] : raw_mapq(c[0]->score - c[0]->csub, opt->a);
			q_se[1] = q_se[1] < raw_mapq(c[1]->score - c[1]->csub, opt->a) ? q_se[1] : raw_mapq(c[1]->score - c[1]->csub, opt->a);
		} else {
			z[0] = z[1] = 0;
			q_se[0] = o_approx_mapq_se(opt, &a[0].a[0]);
			q_se[1] = o_approx_mapq_se(opt, &a[1].a[0]);
		}
		for (i = 0; i < 2; ++i) {
			int k = a[i].a[z[i]].secondary_all;
			if (k >= 0 && k < n_pri[i]) { /* swap primary and secondary when both are non-ALT */
				for (j = 0; j < (int)a[i].n; ++j)
					if (a[i].a[j].secondary_all == k || j == k)
						a[i].a[j].secondary_all = z[i];
				a[i].a[z[i]].secondary_all = -1;
			}
		}
		if (!(opt->flag & O_F_ALL)) {
			for (i = 0; i < 2; ++i)
				XA[i] = o_gen_alt(opt, bns, pac, &a[i], s[i].l_seq, s[i].seq);
		} else XA[0] = XA[1] = 0;
		for (i = 0; i < 2; ++i) {
			h[i] = o_reg2aln(opt, bns, pac, s[i].l_seq, s[i].seq, &a[i].a[z[i]]);
			h[i].mapq = q_se[i];
			h[i].flag |= 0x40 << i | extra_flag;
			h[i].XA = XA[i] ? XA[i][z[i]] : 0;
			aa[i][n_aa[i]++] = h[i];
			if (n_pri[i] < (int)a[i].n) { /* the read has ALT hits */
				o_alnreg_t *p = &a[i].a[n_pri[i]];
				if (p->score < opt->T || p->secondary >= 0 || !p->is_alt) continue;
				g[i] = o_reg2aln(opt, bns, pac, s[i].l_seq, s[i].seq, p);
				g[i].flag |= 0x800 | 0x40 << i | extra_flag;
				g[i].XA = XA[i] ? XA[i][n_pri[i]] : 0;
				aa[i][n_aa[i]++] = g[i];
			}
		}
		for (i = 0; i < n_aa[0]; ++i)
			o_aln2out(opt, bns, &str, &s[0], n_aa[0], aa[0], i, &h[1]);
		s[0].out = str; str.l = str.m = 0; str.s = 0;
		for (i = 0; i < n_aa[1]; ++i)
			o_aln2out(opt, bns, &str, &s[1], n_aa[1], aa[1], i, &h[0]);
		s[1].out = str;
		for (i = 0; i < 2; ++i) {
			free(h[i].cigar); free(g[i].cigar);
			if (XA[i] == 0) continue;
			for (j = 0; j < (int)a[i].n; ++j) free(XA[i][j]);
			free(XA[i]);
		}
	} else goto no_pairing;
	return n;

no_pairing:
	for (i = 0; i < 2; ++i) {
		int which = -1;
		if (a[i].n) {
			if (a[i].a[0].score >= opt->T) which = 0;
			else if (n_pri[i] < (int)a[i].n && a[i].a[n_pri[i]].score >= opt->T)
				which = n_pri[i];
		}
		if (which >= 0) h[i] = o_reg2aln(opt, bns, pac, s[i].l_seq, s[i].seq, &a[i].a[which]);
		else h[i] = o_reg2aln(opt, bns, pac, s[i].l_seq, s[i].seq, 0);
	}
	if (!(opt->flag & O_F_NOPAIRING) && h[0].rid == h[1].rid && h[0].rid >= 0) {
		int64_t dist;
		int d;
		d = infer_dir(bns->l_pac, a[0].a[0].rb, a[1].a[0].rb, &dist);
		if (!pes[d].failed && dist >= pes[d].low && dist <= pes[d].high) extra_flag |= 2;
	}
	o_reg2sam(opt, bns, pac, &s[0], &a[0], 0x41 | extra_flag, &h[1]);
	o_reg2sam(opt, bns, pac, &s[1], &a[1], 0x81 | extra_flag, &h[0]);
	free(h[0].cigar); free(h[1].cigar);
	return n;
}
