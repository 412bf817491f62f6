/*
 * o_bwt.c -- CPU ORACLE (test infrastructure): FM-index primitives.
 *
 * Restates upstream lh3/bwa@cb950614 bwt.c (bwt_occ4, bwt_2occ4, bwt_extend,
 * bwt_sa/bwt_invPsi, bwt_smem1, bwt_seed_strategy1), reached from the reference
 * only through jnibwa.c:214.  The rank / interval / SMEM definitions were
 * verified against a brute-force suffix array of src/test/resources/ref.fa
 * (SURVEY.md App. B, items marked with a tick) and are re-verified by
 * tests/test_oracle_fmindex.py.
 */
#include <stdlib.h>
#include <string.h>
#include "bwa_oracle.h"
#include "o_internal.h"

__thread o_counters_t o_tl_cnt;

/* number of each of A,C,G,T among the 16 2-bit symbols of w; returned packed one byte per symbol */
static inline uint32_t cnt16(uint32_t w)
{
	uint32_t x = 0;
	int i;
	for (i = 0; i < 16; ++i) x += 1u << ((w >> (i << 1) & 3) << 3);
	return x;
}

/* occ(k, c) for all four c: # of c in BWT$[0..k]; SURVEY.md App. B "Rank/occ" */
void o_bwt_occ4(const o_bwt_t *bwt, bwtint_t k, bwtint_t cnt[4])
{
	const uint32_t *p, *end;
	uint32_t x = 0, tmp;
	if (k == (bwtint_t)-1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
	k -= (k >= bwt->primary);               /* the sentinel is not stored */
	p = bwt->bwt + (k >> 7 << 4);           /* 64-byte block: 4 u64 counts + 8 u32 of symbols */
	memcpy(cnt, p, 4 * sizeof(bwtint_t));
	p += 8;
	end = p + ((k >> 4) - ((k & ~(bwtint_t)127) >> 4));
	for (; p < end; ++p) x += cnt16(*p);
	tmp = *p & ~((1u << ((~k & 15) << 1)) - 1); /* keep symbols 0..k&15 (MSB first), zero the rest */
	x += cnt16(tmp) - (uint32_t)(~k & 15);      /* the zeroed tail was counted as 'A' */
	cnt[0] += x & 0xff; cnt[1] += x >> 8 & 0xff; cnt[2] += x >> 16 & 0xff; cnt[3] += x >> 24;
}

static inline bwtint_t occ1(const o_bwt_t *bwt, bwtint_t k, int c)
{
	bwtint_t cnt[4];
	o_bwt_occ4(bwt, k, cnt);
	return cnt[c];
}

/* bidirectional extension by one base, all four bases at once; App. B "Bidirectional interval" */
void o_bwt_extend(const o_bwt_t *bwt, const o_intv_t *ik, o_intv_t ok[4], int is_back)
{
	bwtint_t tk[4], tl[4];
	int i, f = !is_back;
	o_bwt_occ4(bwt, ik->x[f] - 1, tk);
	o_bwt_occ4(bwt, ik->x[f] - 1 + ik->x[2], tl);
	o_tl_cnt.n_ext++;
	for (i = 0; i < 4; ++i) {
		ok[i].x[f] = bwt->L2[i] + 1 + tk[i];
		ok[i].x[2] = tl[i] - tk[i];
	}
	ok[3].x[is_back] = ik->x[is_back] + (ik->x[f] <= bwt->primary && ik->x[f] + ik->x[2] - 1 >= bwt->primary);
	ok[2].x[is_back] = ok[3].x[is_back] + ok[3].x[2];
	ok[1].x[is_back] = ok[2].x[is_back] + ok[2].x[2];
	ok[0].x[is_back] = ok[1].x[is_back] + ok[1].x[2];
}

/* text position of rank k by LF-walk to the next sampled rank; App. B "SA lookup" */
bwtint_t o_bwt_sa(const o_bwt_t *bwt, bwtint_t k)
{
	bwtint_t sa = 0, mask = bwt->sa_intv - 1;
	while (k & mask) {
		bwtint_t x;
		int c;
		++sa;
		o_tl_cnt.n_lf++;
		if (k == bwt->primary) { k = 0; continue; }
		x = k - (k > bwt->primary);
		c = bwt->bwt[(x >> 7 << 4) + 8 + ((x & 0x7f) >> 4)] >> ((~x & 0xf) << 1) & 3;
		k = bwt->L2[c] + occ1(bwt, k, c);
	}
	o_tl_cnt.n_sa++;
	return sa + bwt->sa[k / bwt->sa_intv];
}

static inline void set_intv(const o_bwt_t *bwt, int c, o_intv_t *ik)
{
	ik->x[0] = bwt->L2[c] + 1;
	ik->x[2] = bwt->L2[c + 1] - bwt->L2[c];
	ik->x[1] = bwt->L2[3 - c] + 1;
	ik->info = 0;
}

static inline void intv_push(o_intv_v *v, const o_intv_t *p)
{
	if (v->n == v->m) {
		v->m = v->m ? v->m << 1 : 4;
		v->a = (o_intv_t*)realloc(v->a, v->m * sizeof(o_intv_t));
	}
	v->a[v->n++] = *p;
}

static void intv_reverse(o_intv_v *v)
{
	size_t i;
	for (i = 0; i < v->n >> 1; ++i) {
		o_intv_t t = v->a[i];
		v->a[i] = v->a[v->n - 1 - i];
		v->a[v->n - 1 - i] = t;
	}
}

/* all SMEMs covering position x with interval size >= min_intv; App. B "SMEM(x, min_intv)" */
int o_bwt_smem1(const o_bwt_t *bwt, int len, const uint8_t *q, int x, int min_intv, o_intv_v *mem, o_intv_v tmpvec[2])
{
	int i, j, c, ret;
	o_intv_t ik, ok[4];
	o_intv_v *prev = &tmpvec[0], *curr = &tmpvec[1], *swap;

	mem->n = 0;
	if (q[x] > 3) return x + 1;
	if (min_intv < 1) min_intv = 1;
	set_intv(bwt, q[x], &ik);
	ik.info = x + 1;
	for (i = x + 1, curr->n = 0; i < len; ++i) { /* forward extension */
		if (q[i] < 4) {
			c = 3 - q[i];
			o_bwt_extend(bwt, &ik, ok, 0);
			if (ok[c].x[2] != ik.x[2]) {
				intv_push(curr, &ik);
				if (ok[c].x[2] < (bwtint_t)min_intv) break;
			}
			ik = ok[c]; ik.info = i + 1;
		} else {
			intv_push(curr, &ik);
			break;
		}
	}
	if (i == len) intv_push(curr, &ik);
	intv_reverse(curr);                /* longest match first */
	ret = (int)curr->a[0].info;
	swap = curr; curr = prev; prev = swap;

	for (i = x - 1; i >= -1; --i) {    /* backward extension, all candidates in lock-step */
		c = i < 0 ? -1 : q[i] < 4 ? q[i] : -1;
		for (j = 0, curr->n = 0; j < (int)prev->n; ++j) {
			o_intv_t *p = &prev->a[j];
			if (c >= 0) o_bwt_extend(bwt, p, ok, 1);
			if (c < 0 || ok[c].x[2] < (bwtint_t)min_intv) {
				if (curr->n == 0) { /* no longer candidate survived this step */
					if (mem->n == 0 || (bwtint_t)(i + 1) < mem->a[mem->n - 1].info >> 32) {
						ik = *p; ik.info |= (uint64_t)(i + 1) << 32;
						intv_push(mem, &ik);
					}
				}
			} else if (curr->n == 0 || ok[c].x[2] != curr->a[curr->n - 1].x[2]) {
				ok[c].info = p->info;
				intv_push(curr, &ok[c]);
			}
		}
		if (curr->n == 0) break;
		swap = curr; curr = prev; prev = swap;
	}
	intv_reverse(mem);                 /* sorted by start */
	return ret;
}

/* pass-3 greedy seeds; App. B "Seed passes" P3 */
int o_bwt_seed_strategy1(const o_bwt_t *bwt, int len, const uint8_t *q, int x, int min_len, int max_intv, o_intv_t *mem)
{
	int i, c;
	o_intv_t ik, ok[4];
	memset(mem, 0, sizeof(o_intv_t));
	if (q[x] > 3) return x + 1;
	set_intv(bwt, q[x], &ik);
	for (i = x + 1; i < len; ++i) {
		if (q[i] < 4) {
			c = 3 - q[i];
			o_bwt_extend(bwt, &ik, ok, 0);
			if (ok[c].x[2] < (bwtint_t)max_intv && i - x >= min_len) {
				*mem = ok[c];
				mem->info = (uint64_t)x << 32 | (uint32_t)(i + 1);
				return i + 1;
			}
			ik = ok[c];
		} else return i + 1;
	}
	return len;
}
