/*
 * o_index.c -- CPU ORACLE (test infrastructure): index I/O and reference access.
 *
 * Restates upstream lh3/bwa@cb950614 bwa.c (bwa_idx_load / bwa_idx2mem /
 * bwa_mem2idx), bwt.c (bwt_restore_bwt / bwt_restore_sa) and bntseq.c
 * (bns_restore, bns_pos2rid, bns_intv2rid, bns_get_seq, bns_fetch_seq), reached
 * from jnibwa.c:127-128,160.  Formats: SURVEY.md App. A.1-A.4, each verified
 * against src/test/resources/ref.fa.{amb,ann,bwt,pac,sa}.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "bwa_oracle.h"
#include "o_internal.h"

/* host-ABI struct dumps inside the .img (App. A.4); pointer fields are stale */
typedef struct {
	uint64_t primary, L2[5], seq_len, bwt_size;
	uint64_t ptr_bwt;
	uint32_t cnt_table[256];
	int32_t sa_intv, pad_;
	uint64_t n_sa;
	uint64_t ptr_sa;
} img_bwt_t;                       /* 1120 bytes */
typedef struct {
	int64_t l_pac;
	int32_t n_seqs;
	uint32_t seed;
	uint64_t ptr_anns;
	int32_t n_holes, pad_;
	uint64_t ptr_ambs, ptr_fp;
} img_bns_t;                       /* 48 bytes */
typedef struct {
	int64_t offset;
	int32_t len, n_ambs;
	uint32_t gi;
	int32_t is_alt;
	uint64_t ptr_name, ptr_anno;
} img_ann_t;                       /* 40 bytes */

static uint8_t *slurp(const char *fn, size_t *len)
{
	FILE *fp = fopen(fn, "rb");
	uint8_t *buf;
	long n;
	if (!fp) return 0;
	fseek(fp, 0, SEEK_END); n = ftell(fp); fseek(fp, 0, SEEK_SET);
	buf = malloc(n + 1);
	if (fread(buf, 1, n, fp) != (size_t)n) { free(buf); fclose(fp); return 0; }
	buf[n] = 0;
	fclose(fp);
	*len = n;
	return buf;
}

static void gen_cnt_table(uint32_t tab[256])
{
	int i, j;
	for (i = 0; i != 256; ++i) {
		uint32_t x = 0;
		for (j = 0; j != 4; ++j)
			x |= (uint32_t)(((i & 3) == j) + ((i >> 2 & 3) == j) + ((i >> 4 & 3) == j) + (i >> 6 == j)) << (j << 3);
		tab[i] = x;
	}
}

/* Build the contiguous image directly from the five index files (bwa_idx_load + bwa_idx2mem). */
static uint8_t *image_from_files(const char *prefix, size_t *l_mem)
{
	char fn[4096];
	size_t l_bwt, l_sa, l_ann, l_amb, l_pacf, l_alt = 0, k;
	uint8_t *f_bwt = 0, *f_sa = 0, *f_ann = 0, *f_amb = 0, *f_pac = 0, *f_alt = 0, *mem = 0;
	img_bwt_t hb;
	img_bns_t hn;
	img_ann_t *anns = 0;
	char **names = 0, **annos = 0;
	o_amb_t *ambs = 0;
	int i;
	size_t strbytes = 0;

	snprintf(fn, sizeof fn, "%s.bwt", prefix); f_bwt = slurp(fn, &l_bwt);
	snprintf(fn, sizeof fn, "%s.sa", prefix);  f_sa  = slurp(fn, &l_sa);
	snprintf(fn, sizeof fn, "%s.ann", prefix); f_ann = slurp(fn, &l_ann);
	snprintf(fn, sizeof fn, "%s.amb", prefix); f_amb = slurp(fn, &l_amb);
	snprintf(fn, sizeof fn, "%s.pac", prefix); f_pac = slurp(fn, &l_pacf);
	snprintf(fn, sizeof fn, "%s.alt", prefix); f_alt = slurp(fn, &l_alt);
	if (!f_bwt || !f_sa || !f_ann || !f_amb || !f_pac) goto fail;

	memset(&hb, 0, sizeof hb);
	/* .bwt: primary, L2[1..4], words (App. A.2) */
	memcpy(&hb.primary, f_bwt, 8);
	memcpy(&hb.L2[1], f_bwt + 8, 32);
	hb.bwt_size = (l_bwt - 40) >> 2;
	hb.seq_len = hb.L2[4];
	gen_cnt_table(hb.cnt_table);
	/* .sa: primary, L2[1..4] (skipped), sa_intv, seq_len, then sa[1..] (App. A.3) */
	{
		uint64_t primary, sa_intv, seq_len;
		memcpy(&primary, f_sa, 8);
		memcpy(&sa_intv, f_sa + 40, 8);
		memcpy(&seq_len, f_sa + 48, 8);
		if (primary != hb.primary || seq_len != hb.seq_len) goto fail;
		hb.sa_intv = (int32_t)sa_intv;
		hb.n_sa = (hb.seq_len + sa_intv) / sa_intv;
		if (l_sa < 56 + (hb.n_sa - 1) * 8) goto fail;
	}
	/* .ann / .amb text (App. A.4) */
	memset(&hn, 0, sizeof hn);
	{
		char *p = (char*)f_ann, *q;
		long long l_pac; int n_seqs; unsigned seed; int nread;
		if (sscanf(p, "%lld%d%u%n", &l_pac, &n_seqs, &seed, &nread) != 3) goto fail;
		p += nread;
		hn.l_pac = l_pac; hn.n_seqs = n_seqs; hn.seed = seed;
		anns = calloc(n_seqs, sizeof(img_ann_t));
		names = calloc(n_seqs, sizeof(char*)); annos = calloc(n_seqs, sizeof(char*));
		for (i = 0; i < n_seqs; ++i) {
			char name[1024];
			unsigned gi; long long off; int len, n_ambs;
			if (sscanf(p, "%u%1023s%n", &gi, name, &nread) != 2) goto fail;
			p += nread;
			names[i] = strdup(name);
			q = p; while (*q && *q != '\n') ++q;   /* rest of the line = " comment" */
			{
				size_t l = q - p;
				char *c = malloc(l + 1);
				memcpy(c, p, l); c[l] = 0;
				if (l > 1 && strcmp(c, " (null)") != 0) annos[i] = strdup(c + 1);
				else annos[i] = strdup("");
				free(c);
			}
			p = q;
			if (sscanf(p, "%lld%d%d%n", &off, &len, &n_ambs, &nread) != 3) goto fail;
			p += nread;
			anns[i].gi = gi; anns[i].offset = off; anns[i].len = len; anns[i].n_ambs = n_ambs;
			strbytes += strlen(names[i]) + strlen(annos[i]) + 2;
		}
	}
	{
		char *p = (char*)f_amb;
		long long l_pac; int n_seqs, n_holes, nread;
		if (sscanf(p, "%lld%d%d%n", &l_pac, &n_seqs, &n_holes, &nread) != 3) goto fail;
		p += nread;
		if (l_pac != hn.l_pac || n_seqs != hn.n_seqs) goto fail;
		hn.n_holes = n_holes;
		ambs = calloc(n_holes ? n_holes : 1, sizeof(o_amb_t));
		for (i = 0; i < n_holes; ++i) {
			long long off; int len; char c[8];
			if (sscanf(p, "%lld%d%1s%n", &off, &len, c, &nread) != 3) goto fail;
			p += nread;
			ambs[i].offset = off; ambs[i].len = len; ambs[i].amb = c[0];
		}
	}
	if (f_alt) { /* <prefix>.alt: first column of non-@ lines names ALT contigs */
		char *p = (char*)f_alt, *e = p + l_alt;
		while (p < e) {
			char *q = p, *ln;
			while (q < e && *q != '\t' && *q != '\n' && *q != '\r') ++q;
			ln = q; while (ln < e && *ln != '\n') ++ln;
			if (*p != '@') {
				size_t l = q - p;
				for (i = 0; i < hn.n_seqs; ++i)
					if (strlen(names[i]) == l && memcmp(names[i], p, l) == 0) anns[i].is_alt = 1;
			}
			p = ln + 1;
		}
	}
	if (l_pacf < (size_t)(hn.l_pac / 4 + 1)) goto fail;

	*l_mem = sizeof(img_bwt_t) + hb.bwt_size * 4 + hb.n_sa * 8 + sizeof(img_bns_t)
	       + (size_t)hn.n_holes * sizeof(o_amb_t) + (size_t)hn.n_seqs * sizeof(img_ann_t) + strbytes + hn.l_pac / 4 + 1;
	mem = calloc(*l_mem, 1);
	k = 0;
	memcpy(mem + k, &hb, sizeof hb); k += sizeof hb;
	memcpy(mem + k, f_bwt + 40, hb.bwt_size * 4); k += hb.bwt_size * 4;
	{
		uint64_t m1 = (uint64_t)-1;
		memcpy(mem + k, &m1, 8);                              /* sa[0] = -1 */
		memcpy(mem + k + 8, f_sa + 56, (hb.n_sa - 1) * 8); k += hb.n_sa * 8;
	}
	memcpy(mem + k, &hn, sizeof hn); k += sizeof hn;
	memcpy(mem + k, ambs, (size_t)hn.n_holes * sizeof(o_amb_t)); k += (size_t)hn.n_holes * sizeof(o_amb_t);
	memcpy(mem + k, anns, (size_t)hn.n_seqs * sizeof(img_ann_t)); k += (size_t)hn.n_seqs * sizeof(img_ann_t);
	for (i = 0; i < hn.n_seqs; ++i) {
		size_t l = strlen(names[i]) + 1; memcpy(mem + k, names[i], l); k += l;
		l = strlen(annos[i]) + 1; memcpy(mem + k, annos[i], l); k += l;
	}
	memcpy(mem + k, f_pac, hn.l_pac / 4 + 1); k += hn.l_pac / 4 + 1;
	assert(k == *l_mem);
fail:
	if (names) for (i = 0; i < hn.n_seqs; ++i) { free(names[i]); free(annos[i]); }
	free(names); free(annos); free(anns); free(ambs);
	free(f_bwt); free(f_sa); free(f_ann); free(f_amb); free(f_pac); free(f_alt);
	return mem;
}

/* bwa_mem2idx: pointer views into the contiguous image */
o_idx_t *oracle_idx_from_image(uint8_t *mem, size_t l_mem, int is_mmap)
{
	o_idx_t *idx = calloc(1, sizeof(o_idx_t));
	size_t k = 0;
	img_bwt_t hb;
	img_bns_t hn;
	int i;
	if (l_mem < sizeof hb) goto bad;
	memcpy(&hb, mem, sizeof hb); k += sizeof hb;
	idx->bwt.primary = hb.primary; memcpy(idx->bwt.L2, hb.L2, sizeof hb.L2);
	idx->bwt.seq_len = hb.seq_len; idx->bwt.bwt_size = hb.bwt_size;
	idx->bwt.sa_intv = hb.sa_intv; idx->bwt.n_sa = hb.n_sa;
	idx->bwt.bwt = (const uint32_t*)(mem + k); k += hb.bwt_size * 4;
	idx->bwt.sa = (const bwtint_t*)(mem + k); k += hb.n_sa * 8;
	if (k + sizeof hn > l_mem) goto bad;
	memcpy(&hn, mem + k, sizeof hn); k += sizeof hn;
	idx->bns.l_pac = hn.l_pac; idx->bns.n_seqs = hn.n_seqs; idx->bns.seed = hn.seed; idx->bns.n_holes = hn.n_holes;
	idx->bns.ambs = (const o_amb_t*)(mem + k); k += (size_t)hn.n_holes * sizeof(o_amb_t);
	idx->bns.anns = calloc(hn.n_seqs, sizeof(o_ann_t));
	for (i = 0; i < hn.n_seqs; ++i) {
		img_ann_t a;
		memcpy(&a, mem + k + (size_t)i * sizeof a, sizeof a);
		idx->bns.anns[i].offset = a.offset; idx->bns.anns[i].len = a.len; idx->bns.anns[i].n_ambs = a.n_ambs;
		idx->bns.anns[i].gi = a.gi; idx->bns.anns[i].is_alt = a.is_alt;
	}
	k += (size_t)hn.n_seqs * sizeof(img_ann_t);
	for (i = 0; i < hn.n_seqs; ++i) {
		idx->bns.anns[i].name = (const char*)(mem + k); k += strlen((const char*)(mem + k)) + 1;
		idx->bns.anns[i].anno = (const char*)(mem + k); k += strlen((const char*)(mem + k)) + 1;
	}
	idx->pac = mem + k; k += hn.l_pac / 4 + 1;
	if (k != l_mem) goto bad;
	idx->mem = mem; idx->l_mem = l_mem; idx->is_mmap = is_mmap;
	return idx;
bad:
	free(idx->bns.anns); free(idx);
	return 0;
}

o_idx_t *oracle_idx_load_files(const char *prefix)
{
	size_t l_mem;
	uint8_t *mem = image_from_files(prefix, &l_mem);
	o_idx_t *idx;
	if (!mem) return 0;
	idx = oracle_idx_from_image(mem, l_mem, 0);
	if (!idx) free(mem);
	return idx;
}

uint8_t *oracle_idx_to_image(const o_idx_t *idx, size_t *l_mem)
{
	uint8_t *m = malloc(idx->l_mem);
	memcpy(m, idx->mem, idx->l_mem);
	*l_mem = idx->l_mem;
	return m;
}

/* ---- reference coordinate helpers (bntseq.c) ---- */

int o_bns_pos2rid(const o_bns_t *bns, int64_t pos_f)
{
	int left, mid, right;
	if (pos_f >= bns->l_pac) return -1;
	left = 0; mid = 0; right = bns->n_seqs;
	while (left < right) {
		mid = (left + right) >> 1;
		if (pos_f >= bns->anns[mid].offset) {
			if (mid == bns->n_seqs - 1) break;
			if (pos_f < bns->anns[mid+1].offset) break;
			left = mid + 1;
		} else right = mid;
	}
	return mid;
}

int o_bns_intv2rid(const o_bns_t *bns, int64_t rb, int64_t re)
{
	int is_rev, rid_b, rid_e;
	if (rb < bns->l_pac && re > bns->l_pac) return -2;
	rid_b = o_bns_pos2rid(bns, o_bns_depos(bns, rb, &is_rev));
	rid_e = rb < re ? o_bns_pos2rid(bns, o_bns_depos(bns, re - 1, &is_rev)) : rid_b;
	return rid_b == rid_e ? rid_b : -1;
}

#define get_pac(pac, l) ((pac)[(l)>>2] >> ((~(l)&3)<<1) & 3)

uint8_t *o_bns_get_seq(int64_t l_pac, const uint8_t *pac, int64_t beg, int64_t end, int64_t *len)
{
	uint8_t *seq = 0;
	if (end < beg) { int64_t t = beg; beg = end; end = t; }
	if (end > l_pac << 1) end = l_pac << 1;
	if (beg < 0) beg = 0;
	if (beg >= l_pac || end <= l_pac) {
		int64_t k, l = 0;
		*len = end - beg;
		seq = malloc(end - beg + 1);
		if (beg >= l_pac) {
			int64_t beg_f = (l_pac << 1) - 1 - end;
			int64_t end_f = (l_pac << 1) - 1 - beg;
			for (k = end_f; k > beg_f; --k) seq[l++] = 3 - get_pac(pac, k);
		} else {
			for (k = beg; k < end; ++k) seq[l++] = get_pac(pac, k);
		}
		o_tl_cnt.n_refbases += (uint64_t)(end - beg);
	} else *len = 0;
	return seq;
}

uint8_t *o_bns_fetch_seq(const o_bns_t *bns, const uint8_t *pac, int64_t *beg, int64_t mid, int64_t *end, int *rid)
{
	int64_t far_beg, far_end, len;
	int is_rev;
	uint8_t *seq;
	if (*end < *beg) { int64_t t = *beg; *beg = *end; *end = t; }
	*rid = o_bns_pos2rid(bns, o_bns_depos(bns, mid, &is_rev));
	far_beg = bns->anns[*rid].offset;
	far_end = far_beg + bns->anns[*rid].len;
	if (is_rev) {
		int64_t tmp = far_beg;
		far_beg = (bns->l_pac << 1) - far_end;
		far_end = (bns->l_pac << 1) - tmp;
	}
	*beg = *beg > far_beg ? *beg : far_beg;
	*end = *end < far_end ? *end : far_end;
	seq = o_bns_get_seq(bns->l_pac, pac, *beg, *end, &len);
	assert(seq && *end - *beg == len);
	return seq;
}

void oracle_idx_destroy(o_idx_t *idx)
{
	if (!idx) return;
	free(idx->bns.anns);
	if (!idx->is_mmap) free(idx->mem);
	free(idx);
}
