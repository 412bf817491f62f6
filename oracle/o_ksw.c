/*
 * o_ksw.c -- CPU ORACLE (test infrastructure): the three DP kernels.
 *
 * Restates upstream lh3/bwa@cb950614 ksw.c:
 *   ksw_extend2  banded affine-gap extension with z-drop      
 *   ksw_global2  banded global alignment + traceback          (row a15)
 *   ksw_align2   local SW used by mate rescue / seed re-score (rows a10, a19)
 * reached from the reference only via jnibwa.c:214.  The local SW upstream is
 * an SSE2 "striped" kernel; its few observable quirks (E computed before the
 * lazy-F correction, row maxima taken before it, smallest-qe tie rule, the
 * second-best score bookkeeping) depend on the striped layout, so the layout is
 * emulated lane by lane here with scalar arithmetic.
 */
#include <stdlib.h>
#include <string.h>
#include "bwa_oracle.h"
#include "o_internal.h"

typedef struct { int32_t h, e; } eh_t;

int o_ksw_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                  int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0,
                  int *_qle, int *_tle, int *_gtle, int *_gscore, int *_max_off)
{
	eh_t *eh;
	int8_t *qp;
	int i, j, k, oe_del = o_del + e_del, oe_ins = o_ins + e_ins, beg, end, max, max_i, max_j, max_ins, max_del, max_ie, gscore, max_off;
	if (h0 < 0) h0 = 0;
	qp = malloc((size_t)qlen * m + 1);
	eh = calloc(qlen + 1, 8);
	for (k = i = 0; k < m; ++k) {
		const int8_t *p = &mat[k * m];
		for (j = 0; j < qlen; ++j) qp[i++] = p[query[j]];
	}
	/* first row: decay from h0 by insertion costs */
	eh[0].h = h0; eh[1].h = h0 > oe_ins ? h0 - oe_ins : 0;
	for (j = 2; j <= qlen && eh[j-1].h > e_ins; ++j)
		eh[j].h = eh[j-1].h - e_ins;
	/* clip the band by the longest affordable gap */
	k = m * m;
	for (i = 0, max = 0; i < k; ++i) max = max > mat[i] ? max : mat[i];
	max_ins = (int)((double)(qlen * max + end_bonus - o_ins) / e_ins + 1.);
	max_ins = max_ins > 1 ? max_ins : 1;
	w = w < max_ins ? w : max_ins;
	max_del = (int)((double)(qlen * max + end_bonus - o_del) / e_del + 1.);
	max_del = max_del > 1 ? max_del : 1;
	w = w < max_del ? w : max_del;
	max = h0, max_i = max_j = -1; max_ie = -1, gscore = -1;
	max_off = 0;
	beg = 0, end = qlen;
	for (i = 0; i < tlen; ++i) {
		int t, f = 0, h1, mx = 0, mj = -1;
		int8_t *q = &qp[target[i] * qlen];
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		if (beg == 0) {
			h1 = h0 - (o_del + e_del * (i + 1));
			if (h1 < 0) h1 = 0;
		} else h1 = 0;
		for (j = beg; j < end; ++j) {
			/* eh[j] = { H(i-1,j-1), E(i,j) }, f = F(i,j), h1 = H(i,j-1) */
			eh_t *p = &eh[j];
			int h, M = p->h, e = p->e;
			p->h = h1;
			M = M ? M + q[j] : 0;      /* a dead (zero) cell cannot restart the alignment */
			h = M > e ? M : e;
			h = h > f ? h : f;
			h1 = h;
			mj = mx > h ? mj : j;      /* row maximum, ties -> largest j */
			mx = mx > h ? mx : h;
			t = M - oe_del; t = t > 0 ? t : 0;
			e -= e_del; e = e > t ? e : t;
			p->e = e;
			t = M - oe_ins; t = t > 0 ? t : 0;
			f -= e_ins; f = f > t ? f : t;
		}
		o_tl_cnt.n_dp_cells += (uint64_t)(end > beg ? end - beg : 0);
		eh[end].h = h1; eh[end].e = 0;
		if (j == qlen) {
			max_ie = gscore > h1 ? max_ie : i;   /* ties -> largest i */
			gscore = gscore > h1 ? gscore : h1;
		}
		if (mx == 0) break;
		if (mx > max) {
			max = mx, max_i = i, max_j = mj;
			max_off = max_off > abs(mj - i) ? max_off : abs(mj - i);
		} else if (zdrop > 0) {
			if (i - max_i > mj - max_j) {
				if (max - mx - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break;
			} else {
				if (max - mx - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break;
			}
		}
		/* shrink the window to the non-zero span of this row */
		for (j = beg; j < end && eh[j].h == 0 && eh[j].e == 0; ++j);
		beg = j;
		for (j = end; j >= beg && eh[j].h == 0 && eh[j].e == 0; --j);
		end = j + 2 < qlen ? j + 2 : qlen;
	}
	free(eh); free(qp);
	if (_qle) *_qle = max_j + 1;
	if (_tle) *_tle = max_i + 1;
	if (_gtle) *_gtle = max_ie + 1;
	if (_gscore) *_gscore = gscore;
	if (_max_off) *_max_off = max_off;
	return max;
}

#define MINUS_INF -0x40000000

static inline uint32_t *push_cigar(int *n_cigar, int *m_cigar, uint32_t *cigar, int op, int len)
{
	if (*n_cigar == 0 || op != (int)(cigar[(*n_cigar) - 1] & 0xf)) {
		if (*n_cigar == *m_cigar) {
			*m_cigar = *m_cigar ? (*m_cigar) << 1 : 4;
			cigar = realloc(cigar, (*m_cigar) << 2);
		}
		cigar[(*n_cigar)++] = len << 4 | op;
	} else cigar[(*n_cigar) - 1] += len << 4;
	return cigar;
}

int o_ksw_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                  int o_del, int e_del, int o_ins, int e_ins, int w, int *n_cigar_, uint32_t **cigar_)
{
	eh_t *eh;
	int8_t *qp;
	int i, j, k, oe_del = o_del + e_del, oe_ins = o_ins + e_ins, score, n_col;
	uint8_t *z;
	if (n_cigar_) *n_cigar_ = 0;
	n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
	z = n_cigar_ && cigar_ ? malloc((size_t)n_col * tlen + 1) : 0;
	qp = malloc((size_t)qlen * m + 1);
	eh = calloc(qlen + 1, 8);
	for (k = i = 0; k < m; ++k) {
		const int8_t *p = &mat[k * m];
		for (j = 0; j < qlen; ++j) qp[i++] = p[query[j]];
	}
	eh[0].h = 0; eh[0].e = MINUS_INF;
	for (j = 1; j <= qlen && j <= w; ++j)
		eh[j].h = -(o_ins + e_ins * j), eh[j].e = MINUS_INF;
	for (; j <= qlen; ++j) eh[j].h = eh[j].e = MINUS_INF;
	for (i = 0; i < tlen; ++i) {
		int32_t f = MINUS_INF, h1, beg, end, t;
		int8_t *q = &qp[target[i] * qlen];
		uint8_t *zi = z ? &z[(size_t)i * n_col] : 0;
		beg = i > w ? i - w : 0;
		end = i + w + 1 < qlen ? i + w + 1 : qlen;
		h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : MINUS_INF;
		for (j = beg; j < end; ++j) {
			eh_t *p = &eh[j];
			int32_t h, M = p->h, e = p->e;
			uint8_t d;
			p->h = h1;
			M += q[j];
			d = M >= e ? 0 : 1;
			h = M >= e ? M : e;
			d = h >= f ? d : 2;
			h = h >= f ? h : f;
			h1 = h;
			t = M - oe_del;
			e -= e_del;
			d |= e > t ? 1 << 2 : 0;
			e  = e > t ? e : t;
			p->e = e;
			t = M - oe_ins;
			f -= e_ins;
			d |= f > t ? 2 << 4 : 0;
			f  = f > t ? f : t;
			if (zi) zi[j - beg] = d;
		}
		o_tl_cnt.n_dp_cells += (uint64_t)(end > beg ? end - beg : 0);
		eh[end].h = h1; eh[end].e = MINUS_INF;
	}
	score = eh[qlen].h;
	if (z) {
		int n_cigar = 0, m_cigar = 0, which = 0;
		uint32_t *cigar = 0, tmp;
		i = tlen - 1; k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
		while (i >= 0 && k >= 0) {
			which = z[(size_t)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
			if (which == 0)      cigar = push_cigar(&n_cigar, &m_cigar, cigar, 0, 1), --i, --k;
			else if (which == 1) cigar = push_cigar(&n_cigar, &m_cigar, cigar, 2, 1), --i;
			else                 cigar = push_cigar(&n_cigar, &m_cigar, cigar, 1, 1), --k;
		}
		if (i >= 0) cigar = push_cigar(&n_cigar, &m_cigar, cigar, 2, i + 1);
		if (k >= 0) cigar = push_cigar(&n_cigar, &m_cigar, cigar, 1, k + 1);
		for (i = 0; i < n_cigar >> 1; ++i)
			tmp = cigar[i], cigar[i] = cigar[n_cigar-1-i], cigar[n_cigar-1-i] = tmp;
		*n_cigar_ = n_cigar, *cigar_ = cigar;
	}
	free(eh); free(qp); free(z);
	return score;
}

/* ------------------------------------------------------------------ */
/* striped local SW: lane-by-lane emulation of the 128-bit vector code */

typedef struct {
	int size, p, slen, qlen, shift, max;
	int *qp;      /* [m][slen][p] */
	int *H0, *H1, *E, *Hmax; /* [slen][p] */
} sw_q_t;

static sw_q_t *sw_qinit(int size, int qlen, const uint8_t *query, int m, const int8_t *mat)
{
	sw_q_t *q = calloc(1, sizeof(sw_q_t));
	int a, i, k, lo = 127, hi = 0, *t;
	q->size = size > 1 ? 2 : 1;
	q->p = 8 * (3 - q->size);
	q->slen = (qlen + q->p - 1) / q->p;
	q->qlen = qlen;
	for (a = 0; a < m * m; ++a) {
		if (mat[a] < lo) lo = mat[a];
		if (mat[a] > hi) hi = mat[a];
	}
	q->max = hi;
	q->shift = (256 - (lo & 0xff)) & 0xff;
	q->qp = malloc(sizeof(int) * (size_t)m * q->slen * q->p + sizeof(int));
	q->H0 = calloc((size_t)q->slen * q->p + 1, sizeof(int));
	q->H1 = calloc((size_t)q->slen * q->p + 1, sizeof(int));
	q->E  = calloc((size_t)q->slen * q->p + 1, sizeof(int));
	q->Hmax = calloc((size_t)q->slen * q->p + 1, sizeof(int));
	t = q->qp;
	for (a = 0; a < m; ++a) {
		int nlen = q->slen * q->p;
		const int8_t *ma = mat + a * m;
		for (i = 0; i < q->slen; ++i)
			for (k = i; k < nlen; k += q->slen)
				*t++ = (k >= qlen ? 0 : ma[query[k]]) + (q->size == 1 ? q->shift : 0);
	}
	return q;
}

static void sw_qfree(sw_q_t *q) { free(q->qp); free(q->H0); free(q->H1); free(q->E); free(q->Hmax); free(q); }

static inline int sat_u8(int x) { return x < 0 ? 0 : x > 255 ? 255 : x; }
static inline int sat_i16(int x) { return x < -32768 ? -32768 : x > 32767 ? 32767 : x; }
static inline int subs_u16(int a, int b) { int x = (int)(uint16_t)a - (int)(uint16_t)b; return x < 0 ? 0 : (int)(int16_t)(uint16_t)x; }

static o_kswr_t sw_core(sw_q_t *q, int tlen, const uint8_t *target, int o_del, int e_del_, int o_ins, int e_ins_, int xtra)
{
	const int p = q->p, slen = q->slen, u8 = q->size == 1;
	int i, n_b = 0, m_b = 0, te = -1, gmax = 0, minsc, endsc;
	int oe_del = o_del + e_del_, oe_ins = o_ins + e_ins_, e_del = e_del_, e_ins = e_ins_;
	uint64_t *b = 0;
	int *H0 = q->H0, *H1 = q->H1, *E = q->E, *Hmax = q->Hmax, *S;
	int h[16], f[16], e[16], mx[16];
	o_kswr_t r = { 0, -1, -1, -1, -1, -1, -1 };
	minsc = (xtra & O_KSW_XSUBO) ? xtra & 0xffff : 0x10000;
	endsc = (xtra & O_KSW_XSTOP) ? xtra & 0xffff : 0x10000;
	memset(E, 0, sizeof(int) * slen * p);
	memset(H0, 0, sizeof(int) * slen * p);
	memset(Hmax, 0, sizeof(int) * slen * p);
	for (i = 0; i < tlen; ++i) {
		int j, k, l, imax, done = 0;
		const int *Sc = q->qp + (size_t)target[i] * slen * p;
		for (l = 0; l < p; ++l) f[l] = 0, mx[l] = 0;
		h[0] = 0;                      /* byte shift: lane l <- lane l-1 of the last vector of the previous row */
		for (l = 1; l < p; ++l) h[l] = H0[(slen - 1) * p + l - 1];
		for (j = 0; j < slen; ++j) {
			for (l = 0; l < p; ++l) {
				int hh, ee = E[j * p + l], t;
				if (u8) { hh = sat_u8(h[l] + Sc[j * p + l]); hh = sat_u8(hh - q->shift); }
				else hh = sat_i16(h[l] + Sc[j * p + l]);
				hh = hh > ee ? hh : ee;
				hh = hh > f[l] ? hh : f[l];
				mx[l] = mx[l] > hh ? mx[l] : hh;
				H1[j * p + l] = hh;
				if (u8) { ee = sat_u8(ee - e_del); t = sat_u8(hh - oe_del); }
				else    { ee = subs_u16(ee, e_del); t = subs_u16(hh, oe_del); }
				ee = ee > t ? ee : t;
				E[j * p + l] = ee;
				if (u8) { f[l] = sat_u8(f[l] - e_ins); t = sat_u8(hh - oe_ins); }
				else    { f[l] = subs_u16(f[l], e_ins); t = subs_u16(hh, oe_ins); }
				f[l] = f[l] > t ? f[l] : t;
				h[l] = H0[j * p + l];
			}
		}
		(void)e;
		for (k = 0; k < 16 && !done; ++k) {   /* lazy-F across segment boundaries */
			for (l = p - 1; l > 0; --l) f[l] = f[l - 1];
			f[0] = 0;
			for (j = 0; j < slen; ++j) {
				int all = 1;
				for (l = 0; l < p; ++l) {
					int hh = H1[j * p + l];
					hh = hh > f[l] ? hh : f[l];
					H1[j * p + l] = hh;
					if (u8) { hh = sat_u8(hh - oe_ins); f[l] = sat_u8(f[l] - e_ins); if (sat_u8(f[l] - hh) != 0) all = 0; }
					else    { hh = subs_u16(hh, oe_ins); f[l] = subs_u16(f[l], e_ins); if (f[l] > hh) all = 0; }
				}
				if (all) { done = 1; break; }
			}
		}
		for (l = 0, imax = mx[0]; l < p; ++l) imax = imax > mx[l] ? imax : mx[l];
		if (imax >= minsc) {
			if (n_b == 0 || (int32_t)b[n_b-1] + 1 != i) {
				if (n_b == m_b) { m_b = m_b ? m_b << 1 : 8; b = realloc(b, 8 * (size_t)m_b); }
				b[n_b++] = (uint64_t)imax << 32 | (uint32_t)i;
			} else if ((int)(b[n_b-1] >> 32) < imax) b[n_b-1] = (uint64_t)imax << 32 | (uint32_t)i;
		}
		if (imax > gmax) {
			gmax = imax; te = i;
			memcpy(Hmax, H1, sizeof(int) * slen * p);
			if (u8) { if (gmax + q->shift >= 255 || gmax >= endsc) break; }
			else if (gmax >= endsc) break;
		}
		S = H1; H1 = H0; H0 = S;
	}
	r.score = u8 ? (gmax + q->shift < 255 ? gmax : 255) : gmax;
	r.te = te;
	if (!u8 || r.score != 255) {
		int max = -1, tmp, low, high, n = slen * p;
		for (i = 0; i < n; ++i) {
			int v = Hmax[i];
			if (v > max) max = v, r.qe = i / p + i % p * slen;
			else if (v == max && (tmp = i / p + i % p * slen) < r.qe) r.qe = tmp;
		}
		if (b) {
			i = (r.score + q->max - 1) / q->max;
			low = te - i; high = te + i;
			for (i = 0; i < n_b; ++i) {
				int ee = (int32_t)b[i];
				if ((ee < low || ee > high) && (int)(b[i] >> 32) > r.score2)
					r.score2 = (int)(b[i] >> 32), r.te2 = ee;
			}
		}
	}
	free(b);
	return r;
}

static void revseq(int l, uint8_t *s)
{
	int i;
	for (i = 0; i < l >> 1; ++i) { uint8_t t = s[i]; s[i] = s[l - 1 - i]; s[l - 1 - i] = t; }
}

o_kswr_t o_ksw_align2(int qlen, uint8_t *query, int tlen, uint8_t *target, int m, const int8_t *mat,
                      int o_del, int e_del, int o_ins, int e_ins, int xtra)
{
	int size = (xtra & O_KSW_XBYTE) ? 1 : 2;
	sw_q_t *q = sw_qinit(size, qlen, query, m, mat);
	o_kswr_t r, rr;
	r = sw_core(q, tlen, target, o_del, e_del, o_ins, e_ins, xtra);
	sw_qfree(q);
	if ((xtra & O_KSW_XSTART) == 0 || ((xtra & O_KSW_XSUBO) && r.score < (xtra & 0xffff))) return r;
	revseq(r.qe + 1, query); revseq(r.te + 1, target);
	q = sw_qinit(size, r.qe + 1, query, m, mat);
	rr = sw_core(q, tlen, target, o_del, e_del, o_ins, e_ins, O_KSW_XSTOP | r.score);
	revseq(r.qe + 1, query); revseq(r.te + 1, target);
	sw_qfree(q);
	if (r.score == rr.score)
		r.tb = r.te - rr.te, r.qb = r.qe - rr.qe;
	return r;
}
