/*
 * oracle/arx_oracle.c -- TEST INFRASTRUCTURE ONLY (see arx_oracle.h).
 *
 * From-scratch CPU restatement of the BWA-MEM half of Arachne's hot path.  "ref:" comments name
 * the file:line under /root/reference/src/gobwa/bwa/ (or /root/reference/src/gobwa/) that each
 * routine follows.  Tie-breaking matters more than arithmetic for bit parity, so the unstable
 * introsort (ksort.h:176-226) and the B-tree used for chaining (kbtree.h) are restated as well.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "arx_oracle.h"

/* ------------------------------------------------------------------------------------------
 * Options: compile-time defaults only, no flag reaches them.  ref: bwamem.c:48-84 (mem_opt_init)
 * ------------------------------------------------------------------------------------------ */
enum {
	OPT_A = 1, OPT_B = 4, OPT_O_DEL = 6, OPT_E_DEL = 1, OPT_O_INS = 6, OPT_E_INS = 1,
	OPT_W = 100, OPT_T = 30, OPT_ZDROP = 100, OPT_PEN_CLIP5 = 5, OPT_PEN_CLIP3 = 5,
	OPT_MAX_MEM_INTV = 20, OPT_MIN_SEED_LEN = 19, OPT_SPLIT_WIDTH = 10, OPT_MAX_OCC = 500,
	OPT_MAX_CHAIN_GAP = 10000, OPT_MIN_CHAIN_WEIGHT = 0, OPT_MAX_CHAIN_EXTEND = 1 << 30,
	OPT_MAX_BAND_TRY = 2
};
static const float OPT_SPLIT_FACTOR = 1.5f, OPT_MASK_LEVEL = 0.50f, OPT_DROP_RATIO = 0.50f, OPT_MASK_LEVEL_REDUN = 0.95f;
static const float OPT_MAPQ_COEF_LEN = 50.f;
#define KSW_XBYTE  0x10000
#define KSW_XSTOP  0x20000
#define KSW_XSUBO  0x40000
#define KSW_XSTART 0x80000

/* scoring matrix: match +a, mismatch -b, anything with N is -1.  ref: bwa.c:109-118 */
static int8_t g_mat[25];
static void fill_scmat(void)
{
	int i, j, k = 0;
	for (i = 0; i < 4; ++i) {
		for (j = 0; j < 4; ++j) g_mat[k++] = i == j? OPT_A : -OPT_B;
		g_mat[k++] = -1;
	}
	for (j = 0; j < 5; ++j) g_mat[k++] = -1;
}

/* ------------------------------------------------------------------------------------------
 * Index container + loader.  ref: bwt.h:46-62, bwt.c:421-462, bntseq.c:98-206, bwa.c:262-289
 * ------------------------------------------------------------------------------------------ */
typedef struct { int64_t offset; int32_t len, is_alt; char *name; } ann_t;
typedef struct {
	uint64_t primary, L2[5], seq_len, n_words;
	uint32_t *bwt;     /* interleaved: per 128 symbols, 4 x u64 cumulative counts then 8 x u32 packed bases */
	int sa_intv; uint64_t n_sa; uint64_t *sa;
	int64_t l_pac; int n_seqs; ann_t *ann;
	uint8_t *pac;
} index_t;

struct ora_ctx {
	index_t ix;
	ora_counters_t cnt;
	int64_t n_reads, n_regs, n_cig;
	int64_t *reg_off, *regs, *alns;
	uint32_t *cigars;
};

static void *read_file(const char *fn, int64_t *size)
{
	FILE *f = fopen(fn, "rb");
	void *p;
	if (!f) return 0;
	fseek(f, 0, SEEK_END); *size = ftell(f); fseek(f, 0, SEEK_SET);
	p = malloc(*size + 16);
	if (fread(p, 1, *size, f) != (size_t)*size) { free(p); fclose(f); return 0; }
	fclose(f);
	return p;
}

static int load_index(index_t *ix, const char *prefix)
{
	char fn[4096];
	int64_t sz, i;
	uint8_t *raw;
	FILE *f;
	memset(ix, 0, sizeof(*ix));
	/* .bwt: u64 primary, u64 L2[1..4], then u32 words.  ref: bwt.c:443-462 */
	snprintf(fn, sizeof fn, "%s.bwt", prefix);
	if (!(raw = read_file(fn, &sz))) return -1;
	memcpy(&ix->primary, raw, 8); memcpy(&ix->L2[1], raw + 8, 32); ix->L2[0] = 0;
	ix->n_words = (sz - 40) >> 2;
	ix->bwt = (uint32_t*)malloc(ix->n_words * 4 + 64);
	memcpy(ix->bwt, raw + 40, ix->n_words * 4);
	ix->seq_len = ix->L2[4];
	free(raw);
	/* .sa: u64 primary, 4 x u64 (skipped), u64 sa_intv, u64 seq_len, then n_sa-1 values; sa[0] = -1.  ref: bwt.c:421-441 */
	snprintf(fn, sizeof fn, "%s.sa", prefix);
	if (!(raw = read_file(fn, &sz))) return -2;
	{
		uint64_t prim, sintv, slen;
		memcpy(&prim, raw, 8); memcpy(&sintv, raw + 40, 8); memcpy(&slen, raw + 48, 8);
		if (prim != ix->primary || slen != ix->seq_len) { free(raw); return -3; }
		ix->sa_intv = (int)sintv;
		ix->n_sa = (ix->seq_len + ix->sa_intv) / ix->sa_intv;
		ix->sa = (uint64_t*)malloc(ix->n_sa * 8);
		ix->sa[0] = (uint64_t)-1;
		memcpy(ix->sa + 1, raw + 56, (ix->n_sa - 1) * 8);
	}
	free(raw);
	/* .ann (text).  ref: bntseq.c:98-140 */
	snprintf(fn, sizeof fn, "%s.ann", prefix);
	if (!(f = fopen(fn, "r"))) return -4;
	{
		long long lp; int ns; unsigned seed;
		if (fscanf(f, "%lld%d%u", &lp, &ns, &seed) != 3) { fclose(f); return -5; }
		ix->l_pac = lp; ix->n_seqs = ns;
		ix->ann = (ann_t*)calloc(ns, sizeof(ann_t));
		for (i = 0; i < ns; ++i) {
			unsigned gi; char name[8192]; int c; long long off; int len, namb;
			if (fscanf(f, "%u%8191s", &gi, name) != 2) { fclose(f); return -5; }
			while ((c = fgetc(f)) != '\n' && c != EOF) {}
			if (fscanf(f, "%lld%d%d", &off, &len, &namb) != 3) { fclose(f); return -5; }
			ix->ann[i].offset = off; ix->ann[i].len = len; ix->ann[i].name = strdup(name);
		}
	}
	fclose(f);
	/* .alt (optional): first token of each non-@ line names an ALT contig.  ref: bntseq.c:171-198 */
	snprintf(fn, sizeof fn, "%s.alt", prefix);
	if ((f = fopen(fn, "r"))) {
		char line[8192];
		while (fgets(line, sizeof line, f)) {
			char *e = line;
			if (line[0] == '@') continue;
			while (*e && *e != '\t' && *e != '\n' && *e != '\r') ++e;
			*e = 0;
			for (i = 0; i < ix->n_seqs; ++i)
				if (strcmp(ix->ann[i].name, line) == 0) ix->ann[i].is_alt = 1;
		}
		fclose(f);
	}
	/* .pac: l_pac/4+1 bytes of 2-bit bases, base l at pac[l>>2] >> ((~l&3)<<1) & 3.  ref: bwa.c:282, bntseq.c:225 */
	snprintf(fn, sizeof fn, "%s.pac", prefix);
	if (!(raw = read_file(fn, &sz))) return -6;
	if (sz < ix->l_pac / 4 + 1) { free(raw); return -7; }
	ix->pac = raw;
	return 0;
}

ora_ctx_t *ora_open(const char *prefix)
{
	ora_ctx_t *c = (ora_ctx_t*)calloc(1, sizeof(ora_ctx_t));
	fill_scmat();
	if (load_index(&c->ix, prefix) != 0) { free(c); return 0; }
	return c;
}

static void free_batch(ora_ctx_t *c)
{
	free(c->reg_off); free(c->regs); free(c->alns); free(c->cigars);
	c->reg_off = c->regs = c->alns = 0; c->cigars = 0; c->n_reads = c->n_regs = c->n_cig = 0;
}

void ora_close(ora_ctx_t *c)
{
	int i;
	if (!c) return;
	free_batch(c);
	for (i = 0; i < c->ix.n_seqs; ++i) free(c->ix.ann[i].name);
	free(c->ix.ann); free(c->ix.bwt); free(c->ix.sa); free(c->ix.pac); free(c);
}
int64_t ora_l_pac(ora_ctx_t *c) { return c->ix.l_pac; }
int64_t ora_seq_len(ora_ctx_t *c) { return c->ix.seq_len; }
int64_t ora_primary(ora_ctx_t *c) { return c->ix.primary; }
int ora_n_seqs(ora_ctx_t *c) { return c->ix.n_seqs; }
void ora_counters(ora_ctx_t *c, ora_counters_t *out, int reset)
{
	if (out) *out = c->cnt;
	if (reset) memset(&c->cnt, 0, sizeof(c->cnt));
}

/* ------------------------------------------------------------------------------------------
 * Reference sequence access.  ref: bntseq.c:349-435, bntseq.h:87
 * ------------------------------------------------------------------------------------------ */
static inline int pac_base(const uint8_t *pac, int64_t l) { return pac[l >> 2] >> ((~l & 3) << 1) & 3; }

static inline int64_t depos(const index_t *ix, int64_t pos, int *is_rev)
{
	*is_rev = pos >= ix->l_pac;
	return *is_rev? (ix->l_pac << 1) - 1 - pos : pos;
}

/* ref: bntseq.c:349-363 (bns_pos2rid) */
static int pos2rid(const index_t *ix, int64_t pos_f)
{
	int left = 0, mid = 0, right = ix->n_seqs;
	if (pos_f >= ix->l_pac) return -1;
	while (left < right) {
		mid = (left + right) >> 1;
		if (pos_f >= ix->ann[mid].offset) {
			if (mid == ix->n_seqs - 1) break;
			if (pos_f < ix->ann[mid + 1].offset) break;
			left = mid + 1;
		} else right = mid;
	}
	return mid;
}

/* ref: bntseq.c:365-373 (bns_intv2rid) */
static int intv2rid(const index_t *ix, int64_t rb, int64_t re)
{
	int r, rid_b, rid_e;
	if (rb < ix->l_pac && re > ix->l_pac) return -2;
	rid_b = pos2rid(ix, depos(ix, rb, &r));
	rid_e = rb < re? pos2rid(ix, depos(ix, re - 1, &r)) : rid_b;
	return rid_b == rid_e? rid_b : -1;
}

/* ref: bntseq.c:398-419 (bns_get_seq): [beg,end) on the doubled coordinate; reverse half = complement of mirrored forward */
static uint8_t *get_seq(const index_t *ix, int64_t beg, int64_t end, int64_t *len)
{
	uint8_t *s = 0;
	int64_t k, l = 0, L = ix->l_pac;
	if (end < beg) { int64_t t = beg; beg = end; end = t; }
	if (end > L << 1) end = L << 1;
	if (beg < 0) beg = 0;
	if (beg >= L || end <= L) {
		*len = end - beg;
		s = (uint8_t*)malloc(end - beg + 1);
		if (beg >= L) {
			int64_t beg_f = (L << 1) - 1 - end, end_f = (L << 1) - 1 - beg;
			for (k = end_f; k > beg_f; --k) s[l++] = 3 - pac_base(ix->pac, k);
		} else for (k = beg; k < end; ++k) s[l++] = pac_base(ix->pac, k);
	} else *len = 0;
	return s;
}

/* ref: bntseq.c:421-447 (bns_fetch_seq): clamp to the contig (and strand) that holds mid */
static uint8_t *fetch_seq(const index_t *ix, int64_t *beg, int64_t mid, int64_t *end, int *rid)
{
	int64_t far_beg, far_end, len;
	int is_rev;
	if (*end < *beg) { int64_t t = *beg; *beg = *end; *end = t; }
	*rid = pos2rid(ix, depos(ix, mid, &is_rev));
	far_beg = ix->ann[*rid].offset;
	far_end = far_beg + ix->ann[*rid].len;
	if (is_rev) {
		int64_t t = far_beg;
		far_beg = (ix->l_pac << 1) - far_end;
		far_end = (ix->l_pac << 1) - t;
	}
	if (*beg < far_beg) *beg = far_beg;
	if (*end > far_end) *end = far_end;
	return get_seq(ix, *beg, *end, &len);
}

int64_t ora_fetch_seq(ora_ctx_t *c, int64_t *beg, int64_t mid, int64_t *end, int *rid, uint8_t *out, int64_t cap)
{
	uint8_t *s = fetch_seq(&c->ix, beg, mid, end, rid);
	int64_t n = *end - *beg, i;
	for (i = 0; i < n && i < cap; ++i) out[i] = s[i];
	free(s);
	return n;
}

/* ------------------------------------------------------------------------------------------
 * FM-index primitives.  ref: bwt.h:72-78, bwt.c:53-115,169-274
 * ------------------------------------------------------------------------------------------ */
typedef struct { uint64_t k, l, s, info; } biv_t; /* k = x[0] (forward), l = x[1] (reverse complement), s = x[2] */

static inline const uint32_t *occ_block(const index_t *ix, uint64_t k) { return ix->bwt + ((k >> 7) << 4); }

/* number of 2-bit symbols equal to c among the top `n` symbols (1..16) of a big-endian packed word */
static inline int count16(uint32_t w, int c, int n)
{
	uint32_t x = w ^ (uint32_t)(0x55555555u * (3 - c)); /* symbols equal to c become 0b11 */
	x = x & (x >> 1) & 0x55555555u;
	if (n < 16) x &= ~((1u << ((16 - n) << 1)) - 1);
	return __builtin_popcount(x);
}

/* ref: bwt.c:169-187 (bwt_occ4): counts of A,C,G,T in B[0..k] of the $-removed string */
static void occ4(const index_t *ix, uint64_t k, uint64_t cnt[4])
{
	const uint32_t *p;
	int c, w, nfull, rem;
	if (k == (uint64_t)-1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
	k -= (k >= ix->primary);
	p = occ_block(ix, k);
	memcpy(cnt, p, 32);
	p += 8;
	nfull = (int)((k & 127) >> 4); rem = (int)(k & 15) + 1;
	for (c = 0; c < 4; ++c) {
		int n = 0;
		for (w = 0; w < nfull; ++w) n += count16(p[w], c, 16);
		n += count16(p[nfull], c, rem);
		cnt[c] += n;
	}
}

/* ref: bwt.c:107-130 (bwt_occ) */
static uint64_t occ1(const index_t *ix, uint64_t k, int c)
{
	uint64_t cnt[4];
	if (k == ix->seq_len) return ix->L2[c + 1] - ix->L2[c];
	if (k == (uint64_t)-1) return 0;
	occ4(ix, k, cnt);
	return cnt[c];
}

/* ref: bwt.c:262-274 (bwt_extend) + bwt.c:189-221 (bwt_2occ4: one block when both ends share it, else two) */
static void extend(const index_t *ix, const biv_t *ik, biv_t ok[4], int is_back, ora_counters_t *cnt)
{
	uint64_t tk[4], tl[4], a = is_back? ik->k : ik->l, b = is_back? ik->l : ik->k; /* a = x[!is_back], b = x[is_back] */
	uint64_t k = a - 1, l = a - 1 + ik->s;
	int i;
	occ4(ix, k, tk); occ4(ix, l, tl);
	if (cnt) {
		uint64_t _k = k - (k >= ix->primary), _l = l - (l >= ix->primary);
		if (_l >> 7 != _k >> 7 || k == (uint64_t)-1 || l == (uint64_t)-1) { ++cnt->ext_two_block; cnt->extb_two_block += is_back; } else { ++cnt->ext_same_block; cnt->extb_same_block += is_back; }
	}
	for (i = 0; i < 4; ++i) {
		uint64_t na = ix->L2[i] + 1 + tk[i];
		if (is_back) ok[i].k = na; else ok[i].l = na;
		ok[i].s = tl[i] - tk[i];
	}
	{
		uint64_t x3 = b + (a <= ix->primary && a + ik->s - 1 >= ix->primary), x2, x1, x0;
		x2 = x3 + ok[3].s; x1 = x2 + ok[2].s; x0 = x1 + ok[1].s;
		if (is_back) { ok[3].l = x3; ok[2].l = x2; ok[1].l = x1; ok[0].l = x0; }
		else { ok[3].k = x3; ok[2].k = x2; ok[1].k = x1; ok[0].k = x0; }
	}
}
/* NB: in the reference x[0] is extended by is_back=1 (backward) and x[1] by is_back=0: "ok[i].x[!is_back] = L2+1+tk".
 * With is_back=1 the field written is x[0]; our biv_t.k is x[0] and biv_t.l is x[1]. */

static inline void set_intv(const index_t *ix, int c, biv_t *ik) /* ref: bwt.h:78 (bwt_set_intv) */
{
	ik->k = ix->L2[c] + 1; ik->s = ix->L2[c + 1] - ix->L2[c]; ik->l = ix->L2[3 - c] + 1; ik->info = 0;
}

/* ref: bwt.c:53-59 (bwt_invPsi), bwt.c:86-96 (bwt_sa) */
static uint64_t sa_lookup(const index_t *ix, uint64_t k, ora_counters_t *cnt)
{
	uint64_t sa = 0, mask = ix->sa_intv - 1;
	int64_t steps8 = -1, steps4 = -1; /* LF steps until the first row that is a multiple of 8 / of 4: the walk to a sample every 8th / 4th row (bench.py roofline_locate) */
	while (k & mask) {
		uint64_t x = k - (k > ix->primary);
		if (steps8 < 0 && (k & 7) == 0) steps8 = (int64_t)sa;
		if (steps4 < 0 && (k & 3) == 0) steps4 = (int64_t)sa;
		int c = occ_block(ix, x)[8 + ((x & 127) >> 4)] >> ((~x & 15) << 1) & 3;
		++sa;
		k = k == ix->primary? 0 : ix->L2[c] + occ1(ix, k, c);
	}
	if (steps8 < 0) steps8 = (int64_t)sa;
	if (steps4 < 0) steps4 = (int64_t)sa;
	if (cnt) { ++cnt->sa_lookups; cnt->sa_lf_steps += sa; cnt->sa_lf_steps8 += steps8; cnt->sa_lf_steps4 += steps4; }
	return sa + ix->sa[k / ix->sa_intv];
}

void ora_occ4(ora_ctx_t *c, int n, const uint64_t *k, uint64_t *out) { int i; for (i = 0; i < n; ++i) occ4(&c->ix, k[i], out + 4 * i); }
void ora_extend(ora_ctx_t *c, int n, const uint64_t *ik3, int is_back, uint64_t *o)
{
	int i, j;
	for (i = 0; i < n; ++i) {
		biv_t ik, ok[4];
		ik.k = ik3[3*i]; ik.l = ik3[3*i+1]; ik.s = ik3[3*i+2]; ik.info = 0;
		memset(ok, 0, sizeof ok);
		extend(&c->ix, &ik, ok, is_back, 0);
		for (j = 0; j < 4; ++j) { o[12*i+3*j] = ok[j].k; o[12*i+3*j+1] = ok[j].l; o[12*i+3*j+2] = ok[j].s; }
	}
}
void ora_sa(ora_ctx_t *c, int n, const uint64_t *k, uint64_t *out) { int i; for (i = 0; i < n; ++i) out[i] = sa_lookup(&c->ix, k[i], 0); }

/* ------------------------------------------------------------------------------------------
 * Generic restatement of klib's introsort on an index array.  ref: ksort.h:146-153 (insertion sort),
 * :154-175 (comb sort), :176-226 (introsort).  Sorting indices with the same comparison sequence
 * yields the same permutation as moving the structs themselves.
 * ------------------------------------------------------------------------------------------ */
typedef int (*lt_fn)(const void *ctx, int a, int b);

static void ks_insertsort(int *s, int *t, lt_fn lt, const void *ctx)
{
	int *i, *j;
	for (i = s + 1; i < t; ++i)
		for (j = i; j > s && lt(ctx, *j, *(j - 1)); --j) { int x = *j; *j = *(j - 1); *(j - 1) = x; }
}

static void ks_combsort(size_t n, int *a, lt_fn lt, const void *ctx)
{
	const double shrink = 1.2473309501039786540366528676643;
	int do_swap;
	size_t gap = n;
	int *i, *j;
	do {
		if (gap > 2) {
			gap = (size_t)(gap / shrink);
			if (gap == 9 || gap == 10) gap = 11;
		}
		do_swap = 0;
		for (i = a; i < a + n - gap; ++i) {
			j = i + gap;
			if (lt(ctx, *j, *i)) { int x = *i; *i = *j; *j = x; do_swap = 1; }
		}
	} while (do_swap || gap > 2);
	if (gap != 1) ks_insertsort(a, a + n, lt, ctx);
}

static void ks_introsort(size_t n, int *a, lt_fn lt, const void *ctx)
{
	struct { int *left, *right; int depth; } stack[128], *top = stack;
	int d, rp, x, *s, *t, *i, *j, *k;
	if (n < 1) return;
	if (n == 2) { if (lt(ctx, a[1], a[0])) { x = a[0]; a[0] = a[1]; a[1] = x; } return; }
	for (d = 2; 1ul << d < n; ++d) {}
	s = a; t = a + (n - 1); d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { ks_combsort(t - s + 1, s, lt, ctx); t = s; continue; }
			i = s; j = t; k = i + ((j - i) >> 1) + 1;
			if (lt(ctx, *k, *i)) { if (lt(ctx, *k, *j)) k = j; }
			else k = lt(ctx, *j, *i)? i : j;
			rp = *k;
			if (k != t) { x = *k; *k = *t; *t = x; }
			for (;;) {
				do ++i; while (lt(ctx, *i, rp));
				do --j; while (i <= j && lt(ctx, rp, *j));
				if (j <= i) break;
				x = *i; *i = *j; *j = x;
			}
			x = *i; *i = *t; *t = x;
			if (i - s > t - i) {
				if (i - s > 16) { top->left = s; top->right = i - 1; top->depth = d; ++top; }
				s = t - i > 16? i + 1 : t;
			} else {
				if (t - i > 16) { top->left = i + 1; top->right = t; top->depth = d; ++top; }
				t = i - s > 16? i - 1 : s;
			}
		} else {
			if (top == stack) { ks_insertsort(a, a + n, lt, ctx); return; }
			--top; s = top->left; t = top->right; d = top->depth;
		}
	}
}

/* ------------------------------------------------------------------------------------------
 * SMEM seeding.  ref: bwt.c:289-351 (bwt_smem1a with max_intv = 0), :358-379 (bwt_seed_strategy1),
 * bwamem.c:114-162 (mem_collect_intv)
 * ------------------------------------------------------------------------------------------ */
typedef struct { int n, m; biv_t *a; } bivec_t;
static inline void bv_push(bivec_t *v, const biv_t *x)
{
	if (v->n == v->m) { v->m = v->m? v->m << 1 : 16; v->a = (biv_t*)realloc(v->a, v->m * sizeof(biv_t)); }
	v->a[v->n++] = *x;
}
static void bv_reverse(bivec_t *v)
{
	int j;
	for (j = 0; j < v->n >> 1; ++j) { biv_t t = v->a[v->n - 1 - j]; v->a[v->n - 1 - j] = v->a[j]; v->a[j] = t; }
}

static int smem1(const index_t *ix, int len, const uint8_t *q, int x, int min_intv, bivec_t *mem, bivec_t *v0, bivec_t *v1, ora_counters_t *cnt)
{
	if (cnt) ++cnt->n_smem_calls;
	int i, j, c, ret;
	biv_t ik, ok[4];
	bivec_t *prev = v0, *curr = v1, *sw;
	mem->n = 0;
	if (q[x] > 3) return x + 1;
	if (min_intv < 1) min_intv = 1;
	set_intv(ix, q[x], &ik);
	ik.info = x + 1;
	memset(ok, 0, sizeof ok);
	for (i = x + 1, curr->n = 0; i < len; ++i) { /* forward: push an interval every time its size changes */
		if (q[i] < 4) {
			c = 3 - q[i];
			extend(ix, &ik, ok, 0, cnt);
			if (ok[c].s != ik.s) {
				bv_push(curr, &ik);
				if (ok[c].s < (uint64_t)min_intv) break;
			}
			ik = ok[c]; ik.info = i + 1;
		} else { bv_push(curr, &ik); break; }
	}
	if (i == len) bv_push(curr, &ik);
	bv_reverse(curr); /* longest matches first */
	ret = (int)curr->a[0].info;
	sw = curr; curr = prev; prev = sw;
	for (i = x - 1; i >= -1; --i) { /* backward: keep matches that cannot be extended and are not contained */
		c = i < 0? -1 : q[i] < 4? q[i] : -1;
		for (j = 0, curr->n = 0; j < prev->n; ++j) {
			biv_t *p = &prev->a[j];
			if (c >= 0) extend(ix, p, ok, 1, cnt);
			if (c < 0 || ok[c].s < (uint64_t)min_intv) {
				if (curr->n == 0) {
					if (mem->n == 0 || (uint64_t)(i + 1) < mem->a[mem->n - 1].info >> 32) {
						ik = *p; ik.info |= (uint64_t)(i + 1) << 32;
						bv_push(mem, &ik);
					}
				}
			} else if (curr->n == 0 || ok[c].s != curr->a[curr->n - 1].s) {
				ok[c].info = p->info;
				bv_push(curr, &ok[c]);
			}
		}
		if (curr->n == 0) break;
		sw = curr; curr = prev; prev = sw;
	}
	bv_reverse(mem); /* sorted by start */
	return ret;
}

static int seed_strategy1(const index_t *ix, int len, const uint8_t *q, int x, int min_len, int max_intv, biv_t *mem, ora_counters_t *cnt)
{
	int i, c;
	biv_t ik, ok[4];
	memset(mem, 0, sizeof(*mem));
	if (q[x] > 3) return x + 1;
	set_intv(ix, q[x], &ik);
	memset(ok, 0, sizeof ok);
	for (i = x + 1; i < len; ++i) {
		if (q[i] < 4) {
			c = 3 - q[i];
			if (cnt) { const int64_t e1 = cnt->ext_same_block, e2 = cnt->ext_two_block; extend(ix, &ik, ok, 0, cnt); cnt->ext3_same_block += cnt->ext_same_block - e1; cnt->ext3_two_block += cnt->ext_two_block - e2; }
			else extend(ix, &ik, ok, 0, cnt);
			if (ok[c].s < (uint64_t)max_intv && i - x >= min_len) {
				*mem = ok[c];
				mem->info = (uint64_t)x << 32 | (i + 1);
				return i + 1;
			}
			ik = ok[c];
		} else return i + 1;
	}
	return len;
}

static int intv_lt(const void *ctx, int a, int b) { const biv_t *v = (const biv_t*)ctx; return v[a].info < v[b].info; }

static void collect_intv(const index_t *ix, int len, const uint8_t *seq, bivec_t *out, ora_counters_t *cnt)
{
	bivec_t mem1 = {0,0,0}, v0 = {0,0,0}, v1 = {0,0,0}, all = {0,0,0};
	int i, k, x = 0, old_n, split_len = (int)(OPT_MIN_SEED_LEN * OPT_SPLIT_FACTOR + .499);
	while (x < len) { /* pass 1: all SMEMs */
		if (seq[x] < 4) {
			x = smem1(ix, len, seq, x, 1, &mem1, &v0, &v1, cnt);
			for (i = 0; i < mem1.n; ++i) {
				int slen = (int)((uint32_t)mem1.a[i].info - (mem1.a[i].info >> 32));
				if (slen >= OPT_MIN_SEED_LEN) bv_push(&all, &mem1.a[i]);
			}
		} else ++x;
	}
	old_n = all.n;
	for (k = 0; k < old_n; ++k) { /* pass 2: re-seed from the middle of long, rare SMEMs */
		biv_t p = all.a[k];
		int start = (int)(p.info >> 32), end = (int32_t)p.info;
		if (end - start < split_len || p.s > OPT_SPLIT_WIDTH) continue;
		smem1(ix, len, seq, (start + end) >> 1, (int)p.s + 1, &mem1, &v0, &v1, cnt);
		for (i = 0; i < mem1.n; ++i)
			if ((int)((uint32_t)mem1.a[i].info - (mem1.a[i].info >> 32)) >= OPT_MIN_SEED_LEN) bv_push(&all, &mem1.a[i]);
	}
	x = 0;
	while (x < len) { /* pass 3: LAST-like */
		if (seq[x] < 4) {
			biv_t m;
			x = seed_strategy1(ix, len, seq, x, OPT_MIN_SEED_LEN, OPT_MAX_MEM_INTV, &m, cnt);
			if (m.s > 0) bv_push(&all, &m);
		} else ++x;
	}
	{ /* ks_introsort(mem_intv) by info */
		int *idx = (int*)malloc((all.n + 1) * sizeof(int));
		out->n = 0;
		for (i = 0; i < all.n; ++i) idx[i] = i;
		ks_introsort(all.n, idx, intv_lt, all.a);
		for (i = 0; i < all.n; ++i) bv_push(out, &all.a[idx[i]]);
		free(idx);
	}
	free(mem1.a); free(v0.a); free(v1.a); free(all.a);
}

int ora_collect_intv(ora_ctx_t *c, int len, const uint8_t *seq, uint64_t *out, int cap)
{
	bivec_t v = {0,0,0};
	int i, n;
	collect_intv(&c->ix, len, seq, &v, 0);
	n = v.n;
	for (i = 0; i < n && i < cap; ++i) { out[4*i] = v.a[i].k; out[4*i+1] = v.a[i].l; out[4*i+2] = v.a[i].s; out[4*i+3] = v.a[i].info; }
	free(v.a);
	return n;
}

/* ------------------------------------------------------------------------------------------
 * Chaining with the reference's B-tree semantics.  ref: bwamem.c:168-315, kbtree.h
 * ------------------------------------------------------------------------------------------ */
typedef struct { int64_t rbeg; int32_t qbeg, len, score; } seed_t;
typedef struct {
	int n, m, first, rid, w, kept, is_alt;
	float frac_rep;
	int64_t pos;
	seed_t *seeds;
} chain_t;

/* B-tree of order t=5 (<= 9 keys per node): KB_DEFAULT_SIZE 512 and a 40-byte key give
 * t = ((512-4-8)/(8+40)+1)>>1 = 5.  ref: kbtree.h:56-57,388.  Keys are chain indices ordered by pos. */
#define BT_T 5
#define BT_MAXK (2 * BT_T - 1)
typedef struct { int is_internal, n; int key[BT_MAXK]; int child[BT_MAXK + 1]; } btnode_t;
typedef struct { btnode_t *nodes; int n_nodes, m_nodes, root, n_keys; const chain_t *ch; } btree_t;

static int bt_new(btree_t *b)
{
	if (b->n_nodes == b->m_nodes) { b->m_nodes = b->m_nodes? b->m_nodes << 1 : 16; b->nodes = (btnode_t*)realloc(b->nodes, b->m_nodes * sizeof(btnode_t)); }
	memset(&b->nodes[b->n_nodes], 0, sizeof(btnode_t));
	return b->n_nodes++;
}
/* ref: kbtree.h:117-131 (__kb_getp_aux): lower-bound within a node; *r = sign(k - key[result]) */
static int bt_getp_aux(const btree_t *b, int xi, int64_t pos, int *r)
{
	const btnode_t *x = &b->nodes[xi];
	int begin = 0, end = x->n, rr;
	if (x->n == 0) return -1;
	while (begin < end) {
		int mid = (begin + end) >> 1;
		if (b->ch[x->key[mid]].pos < pos) begin = mid + 1; else end = mid;
	}
	if (begin == x->n) { if (r) *r = 1; return x->n - 1; }
	rr = (pos > b->ch[x->key[begin]].pos) - (pos < b->ch[x->key[begin]].pos);
	if (r) *r = rr;
	if (rr < 0) --begin;
	return begin;
}
/* ref: kbtree.h:151-168 (kb_intervalp); only `lower` is consumed by mem_chain */
static int bt_lower(const btree_t *b, int64_t pos)
{
	int xi = b->root, lower = -1, i, r = 0;
	for (;;) {
		const btnode_t *x = &b->nodes[xi];
		i = bt_getp_aux(b, xi, pos, &r);
		if (i >= 0 && r == 0) return x->key[i];
		if (i >= 0) lower = x->key[i];
		if (!x->is_internal) return lower;
		xi = x->child[i + 1];
	}
}
/* ref: kbtree.h:177-192 (__kb_split) */
static void bt_split(btree_t *b, int xi, int i, int yi)
{
	int zi = bt_new(b);
	btnode_t *x = &b->nodes[xi], *y = &b->nodes[yi], *z = &b->nodes[zi];
	z->is_internal = y->is_internal;
	z->n = BT_T - 1;
	memcpy(z->key, y->key + BT_T, sizeof(int) * (BT_T - 1));
	if (y->is_internal) memcpy(z->child, y->child + BT_T, sizeof(int) * BT_T);
	y->n = BT_T - 1;
	memmove(x->child + i + 2, x->child + i + 1, sizeof(int) * (x->n - i));
	x->child[i + 1] = zi;
	memmove(x->key + i + 1, x->key + i, sizeof(int) * (x->n - i));
	x->key[i] = y->key[BT_T - 1];
	++x->n;
}
/* ref: kbtree.h:193-211 (__kb_putp_aux) */
static void bt_put_aux(btree_t *b, int xi, int ci)
{
	int64_t pos = b->ch[ci].pos;
	for (;;) {
		btnode_t *x = &b->nodes[xi];
		int i;
		if (!x->is_internal) {
			i = bt_getp_aux(b, xi, pos, 0);
			if (i != x->n - 1) memmove(x->key + i + 2, x->key + i + 1, (x->n - i - 1) * sizeof(int));
			x->key[i + 1] = ci;
			++x->n;
			return;
		}
		i = bt_getp_aux(b, xi, pos, 0) + 1;
		if (b->nodes[x->child[i]].n == BT_MAXK) {
			bt_split(b, xi, i, x->child[i]);
			x = &b->nodes[xi];
			if (pos > b->ch[x->key[i]].pos) ++i;
		}
		xi = x->child[i];
	}
}
/* ref: kbtree.h:212-226 (kb_putp) */
static void bt_put(btree_t *b, int ci)
{
	++b->n_keys;
	if (b->nodes[b->root].n == BT_MAXK) {
		int si = bt_new(b), r = b->root;
		b->root = si; b->nodes[si].is_internal = 1; b->nodes[si].n = 0; b->nodes[si].child[0] = r;
		bt_split(b, si, 0, r);
	}
	bt_put_aux(b, b->root, ci);
}
static void bt_traverse(const btree_t *b, int xi, int *out, int *n) /* in-order.  ref: kbtree.h:352-375 */
{
	const btnode_t *x = &b->nodes[xi];
	int i;
	for (i = 0; i < x->n; ++i) {
		if (x->is_internal) bt_traverse(b, x->child[i], out, n);
		out[(*n)++] = x->key[i];
	}
	if (x->is_internal) bt_traverse(b, x->child[x->n], out, n);
}

/* ref: bwamem.c:190-211 (test_and_merge) */
static int test_and_merge(int64_t l_pac, chain_t *c, const seed_t *p, int seed_rid)
{
	int64_t qend, rend, x, y;
	const seed_t *last = &c->seeds[c->n - 1];
	qend = last->qbeg + last->len;
	rend = last->rbeg + last->len;
	if (seed_rid != c->rid) return 0;
	if (p->qbeg >= c->seeds[0].qbeg && p->qbeg + p->len <= qend && p->rbeg >= c->seeds[0].rbeg && p->rbeg + p->len <= rend)
		return 1; /* contained */
	if ((last->rbeg < l_pac || c->seeds[0].rbeg < l_pac) && p->rbeg >= l_pac) return 0;
	x = p->qbeg - last->qbeg;
	y = p->rbeg - last->rbeg;
	if (y >= 0 && x - y <= OPT_W && y - x <= OPT_W && x - last->len < OPT_MAX_CHAIN_GAP && y - last->len < OPT_MAX_CHAIN_GAP) {
		if (c->n == c->m) { c->m <<= 1; c->seeds = (seed_t*)realloc(c->seeds, c->m * sizeof(seed_t)); }
		c->seeds[c->n++] = *p;
		return 1;
	}
	return 0;
}

/* ref: bwamem.c:213-234 (mem_chain_weight) */
static int chain_weight(const chain_t *c)
{
	int64_t end;
	int j, w = 0, tmp;
	for (j = 0, end = 0; j < c->n; ++j) {
		const seed_t *s = &c->seeds[j];
		if (s->qbeg >= end) w += s->len;
		else if (s->qbeg + s->len > end) w += s->qbeg + s->len - end;
		end = end > s->qbeg + s->len? end : s->qbeg + s->len;
	}
	tmp = w; w = 0;
	for (j = 0, end = 0; j < c->n; ++j) {
		const seed_t *s = &c->seeds[j];
		if (s->rbeg >= end) w += s->len;
		else if (s->rbeg + s->len > end) w += s->rbeg + s->len - end;
		end = end > s->rbeg + s->len? end : s->rbeg + s->len;
	}
	w = w < tmp? w : tmp;
	return w < 1 << 30? w : (1 << 30) - 1;
}

typedef struct { int n; chain_t *a; } chainvec_t;

/* ref: bwamem.c:251-315 (mem_chain) */
static chainvec_t do_chain(const index_t *ix, int len, const uint8_t *seq, ora_counters_t *cnt)
{
	chainvec_t out = {0, 0};
	bivec_t mem = {0,0,0};
	btree_t bt;
	chain_t *ch = 0;
	int n_ch = 0, m_ch = 0, i, b, e, l_rep;
	if (len < OPT_MIN_SEED_LEN) return out;
	memset(&bt, 0, sizeof bt);
	bt.root = bt_new(&bt);
	collect_intv(ix, len, seq, &mem, cnt);
	for (i = 0, b = e = l_rep = 0; i < mem.n; ++i) { /* frac_rep: query span covered by seeds with occ > max_occ */
		int sb = (int)(mem.a[i].info >> 32), se = (int)(uint32_t)mem.a[i].info;
		if (mem.a[i].s <= OPT_MAX_OCC) continue;
		if (sb > e) l_rep += e - b, b = sb, e = se;
		else e = e > se? e : se;
	}
	l_rep += e - b;
	for (i = 0; i < mem.n; ++i) {
		const biv_t *p = &mem.a[i];
		int step, count, slen = (int)((uint32_t)p->info - (p->info >> 32));
		int64_t k;
		step = p->s > OPT_MAX_OCC? (int)(p->s / OPT_MAX_OCC) : 1;
		for (k = count = 0; k < (int64_t)p->s && count < OPT_MAX_OCC; k += step, ++count) {
			seed_t s;
			int rid, to_add = 0;
			s.rbeg = (int64_t)sa_lookup(ix, p->k + k, cnt);
			s.qbeg = (int32_t)(p->info >> 32);
			s.score = s.len = slen;
			rid = intv2rid(ix, s.rbeg, s.rbeg + s.len);
			if (rid < 0) continue;
			if (bt.n_keys) {
				int lower;
				bt.ch = ch;
				lower = bt_lower(&bt, s.rbeg);
				if (lower < 0 || !test_and_merge(ix->l_pac, &ch[lower], &s, rid)) to_add = 1;
			} else to_add = 1;
			if (to_add) {
				chain_t *c;
				if (n_ch == m_ch) { m_ch = m_ch? m_ch << 1 : 16; ch = (chain_t*)realloc(ch, m_ch * sizeof(chain_t)); }
				c = &ch[n_ch];
				memset(c, 0, sizeof(*c));
				c->n = 1; c->m = 4;
				c->seeds = (seed_t*)calloc(c->m, sizeof(seed_t));
				c->seeds[0] = s;
				c->rid = rid; c->pos = s.rbeg;
				c->is_alt = !!ix->ann[rid].is_alt;
				bt.ch = ch;
				bt_put(&bt, n_ch++);
			}
		}
	}
	if (n_ch) { /* in-order traversal = output order */
		int *order = (int*)malloc(n_ch * sizeof(int)), n = 0;
		bt.ch = ch;
		bt_traverse(&bt, bt.root, order, &n);
		out.a = (chain_t*)malloc(n_ch * sizeof(chain_t));
		for (i = 0; i < n; ++i) { out.a[i] = ch[order[i]]; out.a[i].frac_rep = (float)l_rep / len; }
		out.n = n;
		free(order);
	}
	free(ch); free(bt.nodes); free(mem.a);
	return out;
}

/* ref: bwamem.c:318-385 (mem_chain_flt) */
static int flt_lt(const void *ctx, int a, int b) { const chain_t *c = (const chain_t*)ctx; return c[a].w > c[b].w; }
#define CHN_BEG(ch) ((ch).seeds[0].qbeg)
#define CHN_END(ch) ((ch).seeds[(ch).n - 1].qbeg + (ch).seeds[(ch).n - 1].len)
static int chain_flt(int n_chn, chain_t *a)
{
	int i, k, n_kept = 0, *kept_idx, *idx;
	chain_t *tmp;
	if (n_chn == 0) return 0;
	for (i = k = 0; i < n_chn; ++i) {
		chain_t *c = &a[i];
		c->first = -1; c->kept = 0;
		c->w = chain_weight(c);
		if (c->w < OPT_MIN_CHAIN_WEIGHT) free(c->seeds);
		else a[k++] = *c;
	}
	n_chn = k;
	idx = (int*)malloc(n_chn * sizeof(int)); tmp = (chain_t*)malloc(n_chn * sizeof(chain_t));
	for (i = 0; i < n_chn; ++i) idx[i] = i;
	ks_introsort(n_chn, idx, flt_lt, a);
	for (i = 0; i < n_chn; ++i) tmp[i] = a[idx[i]];
	memcpy(a, tmp, n_chn * sizeof(chain_t));
	free(idx); free(tmp);
	kept_idx = (int*)malloc(n_chn * sizeof(int));
	a[0].kept = 3;
	kept_idx[n_kept++] = 0;
	for (i = 1; i < n_chn; ++i) {
		int large_ovlp = 0;
		for (k = 0; k < n_kept; ++k) {
			int j = kept_idx[k];
			int b_max = CHN_BEG(a[j]) > CHN_BEG(a[i])? CHN_BEG(a[j]) : CHN_BEG(a[i]);
			int e_min = CHN_END(a[j]) < CHN_END(a[i])? CHN_END(a[j]) : CHN_END(a[i]);
			if (e_min > b_max && (!a[j].is_alt || a[i].is_alt)) {
				int li = CHN_END(a[i]) - CHN_BEG(a[i]);
				int lj = CHN_END(a[j]) - CHN_BEG(a[j]);
				int min_l = li < lj? li : lj;
				if (e_min - b_max >= min_l * OPT_MASK_LEVEL && min_l < OPT_MAX_CHAIN_GAP) {
					large_ovlp = 1;
					if (a[j].first < 0) a[j].first = i;
					if (a[i].w < a[j].w * OPT_DROP_RATIO && a[j].w - a[i].w >= OPT_MIN_SEED_LEN << 1) break;
				}
			}
		}
		if (k == n_kept) {
			kept_idx[n_kept++] = i;
			a[i].kept = large_ovlp? 2 : 3;
		}
	}
	for (i = 0; i < n_kept; ++i) {
		chain_t *c = &a[kept_idx[i]];
		if (c->first >= 0) a[c->first].kept = 1;
	}
	free(kept_idx);
	for (i = k = 0; i < n_chn; ++i) {
		if (a[i].kept == 0 || a[i].kept == 3) continue;
		if (++k >= OPT_MAX_CHAIN_EXTEND) break;
	}
	for (; i < n_chn; ++i) if (a[i].kept < 3) a[i].kept = 0;
	for (i = k = 0; i < n_chn; ++i) {
		if (a[i].kept == 0) free(a[i].seeds);
		else a[k++] = a[i];
	}
	return k;
}

int ora_chains(ora_ctx_t *c, int len, const uint8_t *seq, int do_flt, int64_t *chains, int cap_c, int64_t *seeds, int cap_s, int *n_seeds_out, uint32_t *frac_rep_bits)
{
	chainvec_t v = do_chain(&c->ix, len, seq, 0);
	int i, j, ns = 0;
	if (do_flt) v.n = chain_flt(v.n, v.a);
	*frac_rep_bits = 0;
	for (i = 0; i < v.n; ++i) {
		chain_t *p = &v.a[i];
		if (i == 0) memcpy(frac_rep_bits, &p->frac_rep, 4);
		if (i < cap_c) {
			int64_t *r = chains + 8 * i;
			r[0] = p->pos; r[1] = p->rid; r[2] = p->n; r[3] = ns; r[4] = p->w; r[5] = p->kept; r[6] = p->first; r[7] = p->is_alt;
		}
		for (j = 0; j < p->n; ++j, ++ns)
			if (ns < cap_s) { int64_t *r = seeds + 4 * ns; r[0] = p->seeds[j].rbeg; r[1] = p->seeds[j].qbeg; r[2] = p->seeds[j].len; r[3] = p->seeds[j].score; }
		free(p->seeds);
	}
	free(v.a);
	*n_seeds_out = ns;
	return v.n;
}

/* ------------------------------------------------------------------------------------------
 * Banded extension.  ref: ksw.c:380-479 (ksw_extend2)
 * ------------------------------------------------------------------------------------------ */
typedef struct { int32_t h, e; } eh_t;

static int ksw_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int w, int end_bonus, int zdrop, int h0,
                       int *qle_, int *tle_, int *gtle_, int *gscore_, int *max_off_, ora_counters_t *cnt)
{
	const int o_del = OPT_O_DEL, e_del = OPT_E_DEL, o_ins = OPT_O_INS, e_ins = OPT_E_INS, oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	eh_t *eh = (eh_t*)calloc(qlen + 1, sizeof(eh_t));
	int i, j, beg, end, max, max_i, max_j, max_ins, max_del, max_ie, gscore, max_off, mx = OPT_A;
	int64_t cells = 0;
	/* first row */
	eh[0].h = h0; eh[1].h = h0 > oe_ins? h0 - oe_ins : 0;
	for (j = 2; j <= qlen && eh[j - 1].h > e_ins; ++j) eh[j].h = eh[j - 1].h - e_ins;
	/* the band cannot be wider than the longest gap the scores allow */
	max_ins = (int)((double)(qlen * mx + end_bonus - o_ins) / e_ins + 1.); max_ins = max_ins > 1? max_ins : 1;
	w = w < max_ins? w : max_ins;
	max_del = (int)((double)(qlen * mx + end_bonus - o_del) / e_del + 1.); max_del = max_del > 1? max_del : 1;
	w = w < max_del? w : max_del;
	max = h0; max_i = max_j = -1; max_ie = -1; gscore = -1; max_off = 0;
	beg = 0; end = qlen;
	for (i = 0; i < tlen; ++i) {
		int t, f = 0, h1, m = 0, mj = -1;
		const int8_t *srow = &g_mat[target[i] * 5];
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; } else h1 = 0;
		for (j = beg; j < end; ++j) {
			eh_t *p = &eh[j];
			int h, M = p->h, e = p->e;
			p->h = h1;
			M = M? M + srow[query[j]] : 0;
			h = M > e? M : e;
			h = h > f? h : f;
			h1 = h;
			mj = m > h? mj : j;
			m = m > h? m : h;
			t = M - oe_del; t = t > 0? t : 0;
			e -= e_del; e = e > t? e : t;
			p->e = e;
			t = M - oe_ins; t = t > 0? t : 0;
			f -= e_ins; f = f > t? f : t;
		}
		cells += end - beg;
		eh[end].h = h1; eh[end].e = 0;
		if (j == qlen) {
			max_ie = gscore > h1? max_ie : i;
			gscore = gscore > h1? gscore : h1;
		}
		if (m == 0) break;
		if (m > max) {
			max = m; max_i = i; max_j = mj;
			max_off = max_off > abs(mj - i)? max_off : abs(mj - i);
		} else if (zdrop > 0) {
			if (i - max_i > mj - max_j) { if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break; }
			else { if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break; }
		}
		for (j = beg; j < end && eh[j].h == 0 && eh[j].e == 0; ++j) {}
		beg = j;
		for (j = end; j >= beg && eh[j].h == 0 && eh[j].e == 0; --j) {}
		end = j + 2 < qlen? j + 2 : qlen;
	}
	free(eh);
	if (cnt) {
		const int rows = i < tlen ? i + 1 : tlen, C = qlen < 32 ? 2 : qlen < 48 ? 3 : qlen < 64 ? 4 : qlen < 96 ? 6 : qlen < 128 ? 8 : qlen < 160 ? 10 : 16;
		cnt->cells_extend += cells; ++cnt->n_extend_calls;
		cnt->ext_rows_qlen += (int64_t)rows * qlen; cnt->ext_rows_cols += (int64_t)rows * 16 * C;
	}
	*qle_ = max_j + 1; *tle_ = max_i + 1; *gtle_ = max_ie + 1; *gscore_ = gscore; *max_off_ = max_off;
	return max;
}

void ora_ksw_extend2(ora_ctx_t *c, int qlen, const uint8_t *q, int tlen, const uint8_t *t, int w, int end_bonus, int zdrop, int h0, int *out)
{
	(void)c;
	out[0] = ksw_extend2(qlen, q, tlen, t, w, end_bonus, zdrop, h0, &out[1], &out[2], &out[3], &out[4], &out[5], 0);
}

/* ------------------------------------------------------------------------------------------
 * Striped u8 local Smith-Waterman, emulated lane by lane.  ref: ksw.c:63-109 (ksw_qinit),
 * :111-230 (ksw_u8), :343-365 (ksw_align2).  The 16-lane striping is observable (E is fed the
 * pre-lazy-F H, and F restarts at every lane boundary), so the layout is kept: query position
 * j + lane*slen lives in vector j, byte `lane`.
 * ------------------------------------------------------------------------------------------ */
typedef struct { int score, te, qe, score2, te2, tb, qb; } kswr_t;
typedef struct { int qlen, slen; uint8_t shift, mdiff, max; uint8_t *qp, *H0, *H1, *E, *Hmax; } u8prof_t;

static inline uint8_t sat_add(uint8_t a, uint8_t b) { int s = a + b; return s > 255? 255 : (uint8_t)s; }
static inline uint8_t sat_sub(uint8_t a, uint8_t b) { return a > b? a - b : 0; }

static u8prof_t *u8_qinit(int qlen, const uint8_t *query)
{
	u8prof_t *q = (u8prof_t*)calloc(1, sizeof(u8prof_t));
	int slen = (qlen + 15) / 16, a, i, k, lo = 127, hi = 0;
	uint8_t *t;
	q->qlen = qlen; q->slen = slen;
	q->qp = (uint8_t*)malloc((size_t)16 * slen * 9);
	q->H0 = q->qp + 16 * slen * 5; q->H1 = q->H0 + 16 * slen; q->E = q->H1 + 16 * slen; q->Hmax = q->E + 16 * slen;
	for (a = 0; a < 25; ++a) { if (g_mat[a] < lo) lo = g_mat[a]; if (g_mat[a] > hi) hi = g_mat[a]; }
	q->max = (uint8_t)hi;
	q->shift = (uint8_t)(256 - (uint8_t)lo); /* = 4 */
	q->mdiff = (uint8_t)(hi + q->shift);
	t = q->qp;
	for (a = 0; a < 5; ++a) {
		int nlen = slen * 16;
		const int8_t *ma = g_mat + a * 5;
		for (i = 0; i < slen; ++i)
			for (k = i; k < nlen; k += slen)
				*t++ = (uint8_t)((k >= qlen? 0 : ma[query[k]]) + q->shift);
	}
	return q;
}
static void u8_free(u8prof_t *q) { free(q->qp); free(q); }

static kswr_t ksw_u8(u8prof_t *q, int tlen, const uint8_t *target, int xtra, ora_counters_t *cnt)
{
	const uint8_t oe_del = OPT_O_DEL + OPT_E_DEL, e_del = OPT_E_DEL, oe_ins = OPT_O_INS + OPT_E_INS, e_ins = OPT_E_INS;
	int slen = q->slen, i, j, k, l, n_b = 0, m_b = 0, te = -1, gmax = 0, minsc, endsc;
	uint64_t *b = 0;
	uint8_t *H0 = q->H0, *H1 = q->H1, *E = q->E, *Hmax = q->Hmax, *S;
	kswr_t r = { 0, -1, -1, -1, -1, -1, -1 };
	int64_t rows = 0;
	minsc = (xtra & KSW_XSUBO)? xtra & 0xffff : 0x10000;
	endsc = (xtra & KSW_XSTOP)? xtra & 0xffff : 0x10000;
	memset(E, 0, 16 * slen); memset(H0, 0, 16 * slen); memset(Hmax, 0, 16 * slen);
	for (i = 0; i < tlen; ++i) {
		uint8_t e, h[16], f[16], mx[16], t;
		int imax, done;
		const uint8_t *prof = q->qp + (size_t)target[i] * slen * 16;
		++rows;
		memset(f, 0, 16); memset(mx, 0, 16);
		h[0] = 0;
		for (l = 1; l < 16; ++l) h[l] = H0[(slen - 1) * 16 + l - 1]; /* H(i-1, last vector) shifted by one lane */
		for (j = 0; j < slen; ++j) {
			for (l = 0; l < 16; ++l) {
				uint8_t hh = sat_sub(sat_add(h[l], prof[j * 16 + l]), q->shift);
				e = E[j * 16 + l];
				hh = hh > e? hh : e;
				hh = hh > f[l]? hh : f[l];
				mx[l] = mx[l] > hh? mx[l] : hh;
				H1[j * 16 + l] = hh;
				e = sat_sub(e, e_del); t = sat_sub(hh, oe_del);
				E[j * 16 + l] = e > t? e : t;
				f[l] = sat_sub(f[l], e_ins); t = sat_sub(hh, oe_ins);
				f[l] = f[l] > t? f[l] : t;
				h[l] = H0[j * 16 + l];
			}
		}
		/* lazy-F: at most 16 shifts, early exit when no lane can still improve H */
		for (k = 0, done = 0; k < 16 && !done; ++k) {
			for (l = 15; l > 0; --l) f[l] = f[l - 1];
			f[0] = 0;
			for (j = 0; j < slen; ++j) {
				int all = 1;
				for (l = 0; l < 16; ++l) {
					uint8_t hh = H1[j * 16 + l];
					hh = hh > f[l]? hh : f[l];
					H1[j * 16 + l] = hh;
					hh = sat_sub(hh, oe_ins);
					f[l] = sat_sub(f[l], e_ins);
					if (sat_sub(f[l], hh) != 0) all = 0;
				}
				if (all) { done = 1; break; }
			}
		}
		for (l = 0, imax = 0; l < 16; ++l) imax = imax > mx[l]? imax : mx[l];
		if (imax >= minsc) {
			if (n_b == 0 || (int32_t)b[n_b - 1] + 1 != i) {
				if (n_b == m_b) { m_b = m_b? m_b << 1 : 8; b = (uint64_t*)realloc(b, 8 * m_b); }
				b[n_b++] = (uint64_t)imax << 32 | i;
			} else if ((int)(b[n_b - 1] >> 32) < imax) b[n_b - 1] = (uint64_t)imax << 32 | i;
		}
		if (imax > gmax) {
			gmax = imax; te = i;
			memcpy(Hmax, H1, 16 * slen);
			if (gmax + q->shift >= 255 || gmax >= endsc) break;
		}
		S = H1; H1 = H0; H0 = S;
	}
	if (cnt) { cnt->cells_u8 += rows * slen * 16; ++cnt->n_u8_calls; }
	r.score = gmax + q->shift < 255? gmax : 255;
	r.te = te;
	if (r.score != 255) {
		int max = -1, tmp, low, high, qlen = slen * 16;
		for (i = 0; i < qlen; ++i) {
			int v = Hmax[i];
			if (v > max) { max = v; r.qe = i / 16 + i % 16 * slen; }
			else if (v == max && (tmp = i / 16 + i % 16 * slen) < r.qe) r.qe = tmp;
		}
		if (b) {
			i = (r.score + q->max - 1) / q->max;
			low = te - i; high = te + i;
			for (i = 0; i < n_b; ++i) {
				int e2 = (int32_t)b[i];
				if ((e2 < low || e2 > high) && (int)(b[i] >> 32) > r.score2) { r.score2 = (int)(b[i] >> 32); r.te2 = e2; }
			}
		}
	}
	free(b);
	return r;
}


/* ------------------------------------------------------------------------------------------
 * Striped i16 local Smith-Waterman: what ksw_align2 dispatches to when the query is too long for the byte kernel
 * (bwamem_pair.c:150: KSW_XBYTE only while l_ms * a < 250).  ref: ksw.c:63-109 (ksw_qinit, size 2), :232-334 (ksw_i16).
 * Eight signed 16-bit lanes: query position j + lane * slen in vector j, lane `lane`; no bias, no saturation in reach
 * (scores stay far below 2^15), the gap states' subtractions are the unsigned saturating ones of the original.
 * ------------------------------------------------------------------------------------------ */
typedef struct { int qlen, slen, max; int16_t *qp, *H0, *H1, *E, *Hmax; } i16prof_t;
static i16prof_t *i16_qinit(int qlen, const uint8_t *query)
{
	i16prof_t *q = (i16prof_t*)calloc(1, sizeof(i16prof_t));
	int slen = (qlen + 7) / 8, a, i, k, hi = 0;
	int16_t *t;
	q->qlen = qlen; q->slen = slen;
	q->qp = (int16_t*)malloc(sizeof(int16_t) * 8 * (size_t)slen * 9);
	q->H0 = q->qp + 8 * slen * 5; q->H1 = q->H0 + 8 * slen; q->E = q->H1 + 8 * slen; q->Hmax = q->E + 8 * slen;
	for (a = 0; a < 25; ++a) if (g_mat[a] > hi) hi = g_mat[a];
	q->max = hi;
	t = q->qp;
	for (a = 0; a < 5; ++a) {
		int nlen = slen * 8;
		const int8_t *ma = g_mat + a * 5;
		for (i = 0; i < slen; ++i)
			for (k = i; k < nlen; k += slen)
				*t++ = (int16_t)(k >= qlen? 0 : ma[query[k]]);
	}
	return q;
}
static void i16_free(i16prof_t *q) { free(q->qp); free(q); }
static inline int16_t subs_u16(int16_t a, int b) { int x = (uint16_t)a; x -= b; return (int16_t)(x > 0? x : 0); } /* _mm_subs_epu16 */
static inline int16_t adds_i16(int16_t a, int16_t b) { int x = a + b; return (int16_t)(x > 32767? 32767 : x < -32768? -32768 : x); } /* _mm_adds_epi16 */

static kswr_t ksw_i16(i16prof_t *q, int tlen, const uint8_t *target, int xtra, ora_counters_t *cnt)
{
	const int oe_del = OPT_O_DEL + OPT_E_DEL, e_del = OPT_E_DEL, oe_ins = OPT_O_INS + OPT_E_INS, e_ins = OPT_E_INS;
	int slen = q->slen, i, j, k, l, n_b = 0, m_b = 0, te = -1, gmax = 0, minsc, endsc;
	uint64_t *b = 0;
	int16_t *H0 = q->H0, *H1 = q->H1, *E = q->E, *Hmax = q->Hmax, *S;
	kswr_t r = { 0, -1, -1, -1, -1, -1, -1 };
	int64_t rows = 0;
	minsc = (xtra & KSW_XSUBO)? xtra & 0xffff : 0x10000;
	endsc = (xtra & KSW_XSTOP)? xtra & 0xffff : 0x10000;
	memset(E, 0, 16 * slen); memset(H0, 0, 16 * slen); memset(Hmax, 0, 16 * slen);
	for (i = 0; i < tlen; ++i) {
		int16_t e, h[8], f[8], mx[8], t;
		int imax, done;
		const int16_t *prof = q->qp + (size_t)target[i] * slen * 8;
		++rows;
		memset(f, 0, sizeof f); memset(mx, 0, sizeof mx);
		h[0] = 0;
		for (l = 1; l < 8; ++l) h[l] = H0[(slen - 1) * 8 + l - 1]; /* H(i-1, last vector) shifted by one lane */
		for (j = 0; j < slen; ++j) {
			for (l = 0; l < 8; ++l) {
				int16_t hh = adds_i16(h[l], prof[j * 8 + l]);
				e = E[j * 8 + l];
				hh = hh > e? hh : e;
				hh = hh > f[l]? hh : f[l];
				mx[l] = mx[l] > hh? mx[l] : hh;
				H1[j * 8 + l] = hh;
				e = subs_u16(e, e_del); t = subs_u16(hh, oe_del);
				E[j * 8 + l] = e > t? e : t;
				f[l] = subs_u16(f[l], e_ins); t = subs_u16(hh, oe_ins);
				f[l] = f[l] > t? f[l] : t;
				h[l] = H0[j * 8 + l];
			}
		}
		/* lazy-F: the original loops k < 16 here as well (ksw.c:283), with eight lanes */
		for (k = 0, done = 0; k < 16 && !done; ++k) {
			for (l = 7; l > 0; --l) f[l] = f[l - 1];
			f[0] = 0;
			for (j = 0; j < slen; ++j) {
				int all = 1;
				for (l = 0; l < 8; ++l) {
					int16_t hh = H1[j * 8 + l];
					hh = hh > f[l]? hh : f[l];
					H1[j * 8 + l] = hh;
					hh = subs_u16(hh, oe_ins);
					f[l] = subs_u16(f[l], e_ins);
					if (f[l] > hh) all = 0; /* _mm_cmpgt_epi16 */
				}
				if (all) { done = 1; break; }
			}
		}
		for (l = 0, imax = 0; l < 8; ++l) imax = imax > mx[l]? imax : mx[l];
		if (imax >= minsc) {
			if (n_b == 0 || (int32_t)b[n_b - 1] + 1 != i) {
				if (n_b == m_b) { m_b = m_b? m_b << 1 : 8; b = (uint64_t*)realloc(b, 8 * m_b); }
				b[n_b++] = (uint64_t)imax << 32 | i;
			} else if ((int)(b[n_b - 1] >> 32) < imax) b[n_b - 1] = (uint64_t)imax << 32 | i;
		}
		if (imax > gmax) {
			gmax = imax; te = i;
			memcpy(Hmax, H1, 16 * slen);
			if (gmax >= endsc) break;
		}
		S = H1; H1 = H0; H0 = S;
	}
	if (cnt) { cnt->cells_u8 += rows * slen * 8; ++cnt->n_u8_calls; }
	r.score = gmax; r.te = te;
	{
		int max = -1, tmp, low, high, qlen = slen * 8;
		for (i = 0; i < qlen; ++i) {
			int v = (uint16_t)Hmax[i];
			if (v > max) { max = v; r.qe = i / 8 + i % 8 * slen; }
			else if (v == max && (tmp = i / 8 + i % 8 * slen) < r.qe) r.qe = tmp;
		}
		if (b) {
			i = (r.score + q->max - 1) / q->max;
			low = te - i; high = te + i;
			for (i = 0; i < n_b; ++i) {
				int e2 = (int32_t)b[i];
				if ((e2 < low || e2 > high) && (int)(b[i] >> 32) > r.score2) { r.score2 = (int)(b[i] >> 32); r.te2 = e2; }
			}
		}
	}
	free(b);
	return r;
}

static void revseq(int l, uint8_t *s) { int i; for (i = 0; i < l >> 1; ++i) { uint8_t t = s[i]; s[i] = s[l - 1 - i]; s[l - 1 - i] = t; } }

/* ref: ksw.c:343-365.  query/target are scratch copies (reversed in place and restored) */
static kswr_t ksw_align2_u8(int qlen, uint8_t *query, int tlen, uint8_t *target, int xtra, ora_counters_t *cnt)
{
	u8prof_t *q;
	kswr_t r, rr;
	if (!(xtra & KSW_XBYTE)) { /* ksw_align2's dispatch on the element size (ksw.c:350-353); the second pass keeps the size of the first */
		i16prof_t *p = i16_qinit(qlen, query);
		r = ksw_i16(p, tlen, target, xtra, cnt);
		i16_free(p);
		if ((xtra & KSW_XSTART) == 0 || ((xtra & KSW_XSUBO) && r.score < (xtra & 0xffff))) return r;
		revseq(r.qe + 1, query); revseq(r.te + 1, target);
		p = i16_qinit(r.qe + 1, query);
		rr = ksw_i16(p, tlen, target, KSW_XSTOP | r.score, cnt);
		revseq(r.qe + 1, query); revseq(r.te + 1, target);
		i16_free(p);
		if (r.score == rr.score) { r.tb = r.te - rr.te; r.qb = r.qe - rr.qe; }
		return r;
	}
	q = u8_qinit(qlen, query);
	r = ksw_u8(q, tlen, target, xtra, cnt);
	u8_free(q);
	if ((xtra & KSW_XSTART) == 0 || ((xtra & KSW_XSUBO) && r.score < (xtra & 0xffff))) return r;
	revseq(r.qe + 1, query); revseq(r.te + 1, target);
	q = u8_qinit(r.qe + 1, query);
	rr = ksw_u8(q, tlen, target, KSW_XSTOP | r.score, cnt);
	revseq(r.qe + 1, query); revseq(r.te + 1, target);
	u8_free(q);
	if (r.score == rr.score) { r.tb = r.te - rr.te; r.qb = r.qe - rr.qe; }
	return r;
}

void ora_ksw_align2_i16(ora_ctx_t *c, int qlen, const uint8_t *q, int tlen, const uint8_t *t, int xtra, int *out) /* the 16-bit kernel whatever the length (known-answer tests) */
{
	uint8_t *qq = (uint8_t*)malloc(qlen + 1), *tt = (uint8_t*)malloc(tlen + 1);
	kswr_t r;
	(void)c;
	memcpy(qq, q, qlen); memcpy(tt, t, tlen);
	r = ksw_align2_u8(qlen, qq, tlen, tt, xtra & ~KSW_XBYTE, 0);
	out[0] = r.score; out[1] = r.te; out[2] = r.qe; out[3] = r.score2; out[4] = r.te2; out[5] = r.tb; out[6] = r.qb;
	free(qq); free(tt);
}
void ora_ksw_align2(ora_ctx_t *c, int qlen, const uint8_t *q, int tlen, const uint8_t *t, int xtra, int *out)
{
	uint8_t *qq = (uint8_t*)malloc(qlen + 1), *tt = (uint8_t*)malloc(tlen + 1);
	kswr_t r;
	(void)c;
	memcpy(qq, q, qlen); memcpy(tt, t, tlen);
	r = ksw_align2_u8(qlen, qq, tlen, tt, xtra, 0);
	out[0] = r.score; out[1] = r.te; out[2] = r.qe; out[3] = r.score2; out[4] = r.te2; out[5] = r.tb; out[6] = r.qb;
	free(qq); free(tt);
}

/* ------------------------------------------------------------------------------------------
 * Banded global alignment with traceback.  ref: ksw.c:485-606 (push_cigar, ksw_global2)
 * ------------------------------------------------------------------------------------------ */
#define MINUS_INF (-0x40000000)
typedef struct { int n, m; uint32_t *a; } cigar_t;
static void push_cigar(cigar_t *c, int op, int len)
{
	if (c->n == 0 || op != (int)(c->a[c->n - 1] & 0xf)) {
		if (c->n == c->m) { c->m = c->m? c->m << 1 : 4; c->a = (uint32_t*)realloc(c->a, c->m * 4); }
		c->a[c->n++] = (uint32_t)len << 4 | op;
	} else c->a[c->n - 1] += (uint32_t)len << 4;
}

static int ksw_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int w, cigar_t *cig, ora_counters_t *cnt)
{
	const int o_del = OPT_O_DEL, e_del = OPT_E_DEL, o_ins = OPT_O_INS, e_ins = OPT_E_INS, oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	eh_t *eh = (eh_t*)calloc(qlen + 1, sizeof(eh_t));
	int i, j, k, score, n_col = qlen < 2 * w + 1? qlen : 2 * w + 1;
	uint8_t *z = cig? (uint8_t*)malloc((size_t)n_col * tlen + 1) : 0;
	int64_t cells = 0;
	eh[0].h = 0; eh[0].e = MINUS_INF;
	for (j = 1; j <= qlen && j <= w; ++j) { eh[j].h = -(o_ins + e_ins * j); eh[j].e = MINUS_INF; }
	for (; j <= qlen; ++j) eh[j].h = eh[j].e = MINUS_INF;
	for (i = 0; i < tlen; ++i) {
		int32_t f = MINUS_INF, h1, beg, end, t;
		const int8_t *srow = &g_mat[target[i] * 5];
		uint8_t *zi = z? &z[(size_t)i * n_col] : 0;
		beg = i > w? i - w : 0;
		end = i + w + 1 < qlen? i + w + 1 : qlen;
		h1 = beg == 0? -(o_del + e_del * (i + 1)) : MINUS_INF;
		for (j = beg; j < end; ++j) {
			eh_t *p = &eh[j];
			int32_t h, m = p->h, e = p->e;
			uint8_t d;
			p->h = h1;
			m += srow[query[j]];
			d = m >= e? 0 : 1;
			h = m >= e? m : e;
			d = h >= f? d : 2;
			h = h >= f? h : f;
			h1 = h;
			t = m - oe_del;
			e -= e_del;
			d |= e > t? 1 << 2 : 0;
			e = e > t? e : t;
			p->e = e;
			t = m - oe_ins;
			f -= e_ins;
			d |= f > t? 2 << 4 : 0;
			f = f > t? f : t;
			if (zi) zi[j - beg] = d;
		}
		cells += end - beg;
		eh[end].h = h1; eh[end].e = MINUS_INF;
	}
	score = eh[qlen].h;
	if (cig) {
		int which = 0;
		cig->n = 0;
		i = tlen - 1; k = (i + w + 1 < qlen? i + w + 1 : qlen) - 1;
		while (i >= 0 && k >= 0) {
			which = z[(size_t)i * n_col + (k - (i > w? i - w : 0))] >> (which << 1) & 3;
			if (which == 0) { push_cigar(cig, 0, 1); --i; --k; }
			else if (which == 1) { push_cigar(cig, 2, 1); --i; }
			else { push_cigar(cig, 1, 1); --k; }
		}
		if (i >= 0) push_cigar(cig, 2, i + 1);
		if (k >= 0) push_cigar(cig, 1, k + 1);
		for (i = 0; i < cig->n >> 1; ++i) { uint32_t t = cig->a[i]; cig->a[i] = cig->a[cig->n - 1 - i]; cig->a[cig->n - 1 - i] = t; }
	}
	if (cnt) { cnt->cells_global += cells; ++cnt->n_global_calls; }
	free(eh); free(z);
	return score;
}

int ora_ksw_global2(ora_ctx_t *c, int qlen, const uint8_t *q, int tlen, const uint8_t *t, int w, int *score, uint32_t *cigar, int cap)
{
	cigar_t cg = {0,0,0};
	int i, n;
	(void)c;
	*score = ksw_global2(qlen, q, tlen, t, w, &cg, 0);
	n = cg.n;
	for (i = 0; i < n && i < cap; ++i) cigar[i] = cg.a[i];
	free(cg.a);
	return n;
}

/* ref: bwa.c:121-207 (bwa_gen_cigar2).  Returns 0 and leaves *score untouched when the region is rejected.
 * `query` is reversed in place and restored, like the reference does. */
static int gen_cigar2(const index_t *ix, int w_, int l_query, uint8_t *query, int64_t rb, int64_t re, int *score, cigar_t *cig, int *NM, ora_counters_t *cnt)
{
	uint8_t *rseq;
	int64_t rlen, L = ix->l_pac;
	int i, ok = 0;
	if (cig) cig->n = 0;
	if (NM) *NM = -1;
	if (l_query <= 0 || rb >= re || (rb < L && re > L)) return 0;
	rseq = get_seq(ix, rb, re, &rlen);
	if (re - rb != rlen) goto done;
	if (rb >= L) { revseq(l_query, query); revseq((int)rlen, rseq); } /* so that indels end up left-aligned on the forward strand */
	if (l_query == re - rb && w_ == 0) { /* gap-free shortcut */
		if (cig) push_cigar(cig, 0, l_query);
		for (i = 0, *score = 0; i < l_query; ++i) *score += g_mat[rseq[i] * 5 + query[i]];
	} else {
		int w, max_gap, max_ins, max_del, min_w;
		max_ins = (int)((double)(((l_query + 1) >> 1) * g_mat[0] - OPT_O_INS) / OPT_E_INS + 1.);
		max_del = (int)((double)(((l_query + 1) >> 1) * g_mat[0] - OPT_O_DEL) / OPT_E_DEL + 1.);
		max_gap = max_ins > max_del? max_ins : max_del;
		max_gap = max_gap > 1? max_gap : 1;
		w = (max_gap + abs((int)rlen - l_query) + 1) >> 1;
		w = w < w_? w : w_;
		min_w = abs((int)rlen - l_query) + 3;
		w = w > min_w? w : min_w;
		*score = ksw_global2(l_query, query, (int)rlen, rseq, w, cig, cnt);
	}
	if (NM && cig) { /* NM = mismatches + gap bases (a leading/trailing D does not count); MD is not kept */
		int k, x = 0, y = 0, n_mm = 0, n_gap = 0;
		for (k = 0; k < cig->n; ++k) {
			int op = cig->a[k] & 0xf, len = cig->a[k] >> 4;
			if (op == 0) {
				for (i = 0; i < len; ++i) if (query[x + i] != rseq[y + i]) ++n_mm;
				x += len; y += len;
			} else if (op == 2) {
				if (k > 0 && k < cig->n - 1) n_gap += len;
				y += len;
			} else if (op == 1) { x += len; n_gap += len; }
		}
		*NM = n_mm + n_gap;
	}
	if (rb >= L) revseq(l_query, query);
	ok = 1;
done:
	free(rseq);
	return ok;
}

/* ------------------------------------------------------------------------------------------
 * Alignment regions.  ref: bwamem.h:66-87 (mem_alnreg_t)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
	int64_t rb, re;
	int qb, qe, rid, score, truesc, sub, alt_sc, csub, sub_n, w, seedcov, secondary, secondary_all, seedlen0, n_comp, is_alt;
	float frac_rep;
} reg_t;
typedef struct { int n, m; reg_t *a; } regvec_t;
static reg_t *rv_pushp(regvec_t *v)
{
	if (v->n == v->m) { v->m = v->m? v->m << 1 : 4; v->a = (reg_t*)realloc(v->a, v->m * sizeof(reg_t)); }
	return &v->a[v->n++];
}

/* ref: bwamem.c:621-628 (cal_max_gap) */
static inline int cal_max_gap(int qlen)
{
	int l_del = (int)((double)(qlen * OPT_A - OPT_O_DEL) / OPT_E_DEL + 1.);
	int l_ins = (int)((double)(qlen * OPT_A - OPT_O_INS) / OPT_E_INS + 1.);
	int l = l_del > l_ins? l_del : l_ins;
	l = l > 1? l : 1;
	return l < OPT_W << 1? l : OPT_W << 1;
}

static int u64_lt(const void *ctx, int a, int b) { const uint64_t *v = (const uint64_t*)ctx; return v[a] < v[b]; }

/* ref: bwamem.c:632-786 (mem_chain2aln) */
static void chain2aln(const index_t *ix, int l_query, const uint8_t *query, const chain_t *c, regvec_t *av, ora_counters_t *cnt)
{
	int i, k, rid, max_off[2], aw[2];
	int64_t l_pac = ix->l_pac, rmax[2], tmp, max = 0;
	const seed_t *s;
	uint8_t *rseq;
	uint64_t *srt;
	int *order;
	if (c->n == 0) return;
	rmax[0] = l_pac << 1; rmax[1] = 0;
	for (i = 0; i < c->n; ++i) {
		int64_t b, e;
		const seed_t *t = &c->seeds[i];
		b = t->rbeg - (t->qbeg + cal_max_gap(t->qbeg));
		e = t->rbeg + t->len + ((l_query - t->qbeg - t->len) + cal_max_gap(l_query - t->qbeg - t->len));
		rmax[0] = rmax[0] < b? rmax[0] : b;
		rmax[1] = rmax[1] > e? rmax[1] : e;
		if (t->len > max) max = t->len;
	}
	rmax[0] = rmax[0] > 0? rmax[0] : 0;
	rmax[1] = rmax[1] < l_pac << 1? rmax[1] : l_pac << 1;
	if (rmax[0] < l_pac && l_pac < rmax[1]) {
		if (c->seeds[0].rbeg < l_pac) rmax[1] = l_pac;
		else rmax[0] = l_pac;
	}
	rseq = fetch_seq(ix, &rmax[0], c->seeds[0].rbeg, &rmax[1], &rid);
	/* seeds are visited by score, highest first (keys score<<32|index are unique, so any sort agrees) */
	srt = (uint64_t*)malloc(c->n * 8); order = (int*)malloc(c->n * sizeof(int));
	for (i = 0; i < c->n; ++i) { srt[i] = (uint64_t)c->seeds[i].score << 32 | i; order[i] = i; }
	ks_introsort(c->n, order, u64_lt, srt);
	{ uint64_t *t2 = (uint64_t*)malloc(c->n * 8); for (i = 0; i < c->n; ++i) t2[i] = srt[order[i]]; free(srt); srt = t2; }
	free(order);
	for (k = c->n - 1; k >= 0; --k) {
		reg_t *a;
		s = &c->seeds[(uint32_t)srt[k]];
		for (i = 0; i < av->n; ++i) { /* is the seed already covered by an earlier region (of ANY chain of this read)? */
			reg_t *p = &av->a[i];
			int64_t rd;
			int qd, w, max_gap;
			if (s->rbeg < p->rb || s->rbeg + s->len > p->re || s->qbeg < p->qb || s->qbeg + s->len > p->qe) continue;
			if (s->len - p->seedlen0 > .1 * l_query) continue;
			qd = s->qbeg - p->qb; rd = s->rbeg - p->rb;
			max_gap = cal_max_gap(qd < rd? qd : (int)rd);
			w = max_gap < p->w? max_gap : p->w;
			if (qd - rd < w && rd - qd < w) break;
			qd = p->qe - (s->qbeg + s->len); rd = p->re - (s->rbeg + s->len);
			max_gap = cal_max_gap(qd < rd? qd : (int)rd);
			w = max_gap < p->w? max_gap : p->w;
			if (qd - rd < w && rd - qd < w) break;
		}
		if (i < av->n) {
			for (i = k + 1; i < c->n; ++i) { /* an overlapping off-diagonal seed that was extended keeps this one alive */
				const seed_t *t;
				if (srt[i] == 0) continue;
				t = &c->seeds[(uint32_t)srt[i]];
				if (t->len < s->len * .95) continue;
				if (s->qbeg <= t->qbeg && s->qbeg + s->len - t->qbeg >= s->len >> 2 && t->qbeg - s->qbeg != t->rbeg - s->rbeg) break;
				if (t->qbeg <= s->qbeg && t->qbeg + t->len - s->qbeg >= s->len >> 2 && s->qbeg - t->qbeg != s->rbeg - t->rbeg) break;
			}
			if (i == c->n) { srt[k] = 0; continue; }
		}
		a = rv_pushp(av);
		memset(a, 0, sizeof(reg_t));
		a->w = aw[0] = aw[1] = OPT_W;
		a->score = a->truesc = -1;
		a->rid = c->rid;
		if (s->qbeg) { /* left extension on reversed prefixes */
			uint8_t *rs, *qs;
			int qle, tle, gtle, gscore;
			qs = (uint8_t*)malloc(s->qbeg);
			for (i = 0; i < s->qbeg; ++i) qs[i] = query[s->qbeg - 1 - i];
			tmp = s->rbeg - rmax[0];
			rs = (uint8_t*)malloc(tmp + 1);
			for (i = 0; i < tmp; ++i) rs[i] = rseq[tmp - 1 - i];
			for (i = 0; i < OPT_MAX_BAND_TRY; ++i) {
				int prev = a->score;
				aw[0] = OPT_W << i;
				a->score = ksw_extend2(s->qbeg, qs, (int)tmp, rs, aw[0], OPT_PEN_CLIP5, OPT_ZDROP, s->len * OPT_A, &qle, &tle, &gtle, &gscore, &max_off[0], cnt);
				if (a->score == prev || max_off[0] < (aw[0] >> 1) + (aw[0] >> 2)) break;
			}
			if (gscore <= 0 || gscore <= a->score - OPT_PEN_CLIP5) { a->qb = s->qbeg - qle; a->rb = s->rbeg - tle; a->truesc = a->score; }
			else { a->qb = 0; a->rb = s->rbeg - gtle; a->truesc = gscore; }
			free(qs); free(rs);
		} else { a->score = a->truesc = s->len * OPT_A; a->qb = 0; a->rb = s->rbeg; }
		if (s->qbeg + s->len != l_query) { /* right extension */
			int qle, tle, qe, re, gtle, gscore, sc0 = a->score;
			qe = s->qbeg + s->len;
			re = (int)(s->rbeg + s->len - rmax[0]);
			for (i = 0; i < OPT_MAX_BAND_TRY; ++i) {
				int prev = a->score;
				aw[1] = OPT_W << i;
				a->score = ksw_extend2(l_query - qe, query + qe, (int)(rmax[1] - rmax[0] - re), rseq + re, aw[1], OPT_PEN_CLIP3, OPT_ZDROP, sc0, &qle, &tle, &gtle, &gscore, &max_off[1], cnt);
				if (a->score == prev || max_off[1] < (aw[1] >> 1) + (aw[1] >> 2)) break;
			}
			if (gscore <= 0 || gscore <= a->score - OPT_PEN_CLIP3) { a->qe = qe + qle; a->re = rmax[0] + re + tle; a->truesc += a->score - sc0; }
			else { a->qe = l_query; a->re = rmax[0] + re + gtle; a->truesc += gscore - sc0; }
		} else { a->qe = l_query; a->re = s->rbeg + s->len; }
		for (i = 0, a->seedcov = 0; i < c->n; ++i) {
			const seed_t *t = &c->seeds[i];
			if (t->qbeg >= a->qb && t->qbeg + t->len <= a->qe && t->rbeg >= a->rb && t->rbeg + t->len <= a->re) a->seedcov += t->len;
		}
		a->w = aw[0] > aw[1]? aw[0] : aw[1];
		a->seedlen0 = s->len;
		a->frac_rep = c->frac_rep;
	}
	free(srt); free(rseq);
}

/* ref: bwamem.c:403-435 (mem_patch_reg) */
static int patch_reg(const index_t *ix, uint8_t *query, const reg_t *a, const reg_t *b, int *w_, ora_counters_t *cnt)
{
	int w, score, q_s, r_s;
	double r;
	if (ix == 0 || query == 0) return 0;
	if (a->rb < ix->l_pac && b->rb >= ix->l_pac) return 0;
	if (a->qb >= b->qb || a->qe >= b->qe || a->re >= b->re) return 0;
	w = (int)((a->re - b->rb) - (a->qe - b->qb));
	w = w > 0? w : -w;
	r = (double)(a->re - b->rb) / (b->re - a->rb) - (double)(a->qe - b->qb) / (b->qe - a->qb);
	r = r > 0.? r : -r;
	if (a->re < b->rb || a->qe < b->qb) {
		if (w > OPT_W << 1 || r >= 0.05f) return 0;
	} else if (w > OPT_W << 2 || r >= 0.05f * 2) return 0;
	w += a->w + b->w;
	w = w < OPT_W << 2? w : OPT_W << 2;
	score = 0; /* NB: the reference leaves `score` uninitialised when bwa_gen_cigar2 rejects the region */
	gen_cigar2(ix, w, b->qe - a->qb, query + a->qb, a->rb, b->re, &score, 0, 0, cnt);
	q_s = (int)((double)(b->qe - a->qb) / ((b->qe - b->qb) + (a->qe - a->qb)) * (b->score + a->score) + .499);
	r_s = (int)((double)(b->re - a->rb) / ((b->re - b->rb) + (a->re - a->rb)) * (b->score + a->score) + .499);
	if ((double)score / (q_s > r_s? q_s : r_s) < 0.90f) return 0;
	*w_ = w;
	return score;
}

static int reg_lt_re(const void *ctx, int a, int b) { const reg_t *r = (const reg_t*)ctx; return r[a].re < r[b].re; }
static int reg_lt_score(const void *ctx, int x, int y)
{
	const reg_t *r = (const reg_t*)ctx, *a = &r[x], *b = &r[y];
	return a->score > b->score || (a->score == b->score && (a->rb < b->rb || (a->rb == b->rb && a->qb < b->qb)));
}
static void permute_regs(int n, reg_t *a, lt_fn lt)
{
	int *idx = (int*)malloc((n + 1) * sizeof(int)), i;
	reg_t *t = (reg_t*)malloc((n + 1) * sizeof(reg_t));
	for (i = 0; i < n; ++i) idx[i] = i;
	ks_introsort(n, idx, lt, a);
	for (i = 0; i < n; ++i) t[i] = a[idx[i]];
	memcpy(a, t, n * sizeof(reg_t));
	free(idx); free(t);
}

/* ref: bwamem.c:437-489 (mem_sort_dedup_patch); ix/query are NULL when called from mate rescue */
static int sort_dedup_patch(const index_t *ix, uint8_t *query, int n, reg_t *a, ora_counters_t *cnt)
{
	int m, i, j;
	if (n <= 1) return n;
	permute_regs(n, a, reg_lt_re);
	for (i = 0; i < n; ++i) a[i].n_comp = 1;
	for (i = 1; i < n; ++i) {
		reg_t *p = &a[i];
		if (p->rid != a[i - 1].rid || p->rb >= a[i - 1].re + OPT_MAX_CHAIN_GAP) continue;
		for (j = i - 1; j >= 0 && p->rid == a[j].rid && p->rb < a[j].re + OPT_MAX_CHAIN_GAP; --j) {
			reg_t *q = &a[j];
			int64_t orr, oq, mr, mq;
			int score, w;
			if (q->qe == q->qb) continue;
			orr = q->re - p->rb;
			oq = q->qb < p->qb? q->qe - p->qb : p->qe - q->qb;
			mr = q->re - q->rb < p->re - p->rb? q->re - q->rb : p->re - p->rb;
			mq = q->qe - q->qb < p->qe - p->qb? q->qe - q->qb : p->qe - p->qb;
			if (orr > OPT_MASK_LEVEL_REDUN * mr && oq > OPT_MASK_LEVEL_REDUN * mq) {
				if (p->score < q->score) { p->qe = p->qb; break; }
				else q->qe = q->qb;
			} else if (q->rb < p->rb && (score = patch_reg(ix, query, q, p, &w, cnt)) > 0) {
				p->n_comp += q->n_comp + 1;
				p->seedcov = p->seedcov > q->seedcov? p->seedcov : q->seedcov;
				p->sub = p->sub > q->sub? p->sub : q->sub;
				p->csub = p->csub > q->csub? p->csub : q->csub;
				p->qb = q->qb; p->rb = q->rb;
				p->truesc = p->score = score;
				p->w = w;
				q->qb = q->qe;
			}
		}
	}
	for (i = 0, m = 0; i < n; ++i)
		if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
	n = m;
	permute_regs(n, a, reg_lt_score);
	for (i = 1; i < n; ++i)
		if (a[i].score == a[i - 1].score && a[i].rb == a[i - 1].rb && a[i].qb == a[i - 1].qb) a[i].qe = a[i].qb;
	for (i = 1, m = 1; i < n; ++i)
		if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
	return m;
}

/* ref: bwamem.c:1048-1084 (mem_align1_core).  mem_flt_chained_seeds (:598) returns at once for short
 * reads (5.5*ln(l) > 0.05*l up to l ~ 730), which is asserted here rather than restated. */
static regvec_t align1_core(const index_t *ix, int l_seq, uint8_t *seq, ora_counters_t *cnt)
{
	regvec_t regs = {0,0,0};
	chainvec_t chn;
	int i;
	chn = do_chain(ix, l_seq, seq, cnt);
	chn.n = chain_flt(chn.n, chn.a);
	if (!(5.5f * log(l_seq) > 0.05f * l_seq)) { fprintf(stderr, "[oracle] read too long for the short-read path\n"); abort(); }
	for (i = 0; i < chn.n; ++i) {
		chain2aln(ix, l_seq, seq, &chn.a[i], &regs, cnt);
		free(chn.a[i].seeds);
	}
	free(chn.a);
	regs.n = sort_dedup_patch(ix, seq, regs.n, regs.a, cnt);
	for (i = 0; i < regs.n; ++i)
		if (regs.a[i].rid >= 0 && ix->ann[regs.a[i].rid].is_alt) regs.a[i].is_alt = 1;
	return regs;
}

static void put_reg(int64_t *r, const reg_t *p)
{
	uint32_t fb;
	memcpy(&fb, &p->frac_rep, 4);
	r[0] = p->rb; r[1] = p->re; r[2] = p->qb; r[3] = p->qe; r[4] = p->rid; r[5] = p->score; r[6] = p->truesc;
	r[7] = p->sub; r[8] = p->alt_sc; r[9] = p->csub; r[10] = p->sub_n; r[11] = p->w; r[12] = p->seedcov;
	r[13] = p->secondary; r[14] = p->secondary_all; r[15] = p->seedlen0; r[16] = p->n_comp; r[17] = p->is_alt;
	r[18] = fb; r[19] = 0;
}

int ora_align1(ora_ctx_t *c, int len, const uint8_t *seq, int64_t *regs, int cap)
{
	uint8_t *s = (uint8_t*)malloc(len + 1);
	regvec_t v;
	int i, n;
	memcpy(s, seq, len);
	v = align1_core(&c->ix, len, s, 0);
	n = v.n;
	for (i = 0; i < n && i < cap; ++i) put_reg(regs + (size_t)i * ORA_REG_W, &v.a[i]);
	free(v.a); free(s);
	return n;
}

/* ------------------------------------------------------------------------------------------
 * Mate rescue with Arachne's fixed insert model.  ref: bwamem_pair.c:23-31 (mem_infer_dir),
 * :111-180 (mem_matesw); gobwa.go:229-237 (only FR valid, low -35, high 500)
 * ------------------------------------------------------------------------------------------ */
static const int PES_FAILED[4] = { 1, 0, 1, 1 };
static const int PES_LOW[4] = { 0, -35, 0, 0 }, PES_HIGH[4] = { 0, 500, 0, 0 };

static inline int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist)
{
	int64_t p2;
	int r1 = (b1 >= l_pac), r2 = (b2 >= l_pac);
	p2 = r1 == r2? b2 : (l_pac << 1) - 1 - b2;
	*dist = p2 > b1? p2 - b1 : b1 - p2;
	return (r1 == r2? 0 : 1) ^ (p2 > b1? 0 : 3);
}

static int matesw(const index_t *ix, const reg_t *a, int l_ms, const uint8_t *ms, regvec_t *ma, ora_counters_t *cnt)
{
	int64_t l_pac = ix->l_pac;
	int i, r, skip[4], n = 0, rid = -1;
	for (r = 0; r < 4; ++r) skip[r] = PES_FAILED[r]? 1 : 0;
	for (i = 0; i < ma->n; ++i) {
		int64_t dist;
		r = infer_dir(l_pac, a->rb, ma->a[i].rb, &dist);
		if (dist >= PES_LOW[r] && dist <= PES_HIGH[r]) skip[r] = 1;
	}
	if (skip[0] + skip[1] + skip[2] + skip[3] == 4) return 0;
	for (r = 0; r < 4; ++r) {
		int is_rev, is_larger;
		uint8_t *seq, *ref = 0;
		int64_t rb, re;
		if (skip[r]) continue;
		is_rev = (r >> 1 != (r & 1));
		is_larger = !(r >> 1);
		seq = (uint8_t*)malloc(l_ms + 1);
		if (is_rev) { for (i = 0; i < l_ms; ++i) seq[l_ms - 1 - i] = ms[i] < 4? 3 - ms[i] : 4; }
		else memcpy(seq, ms, l_ms);
		if (!is_rev) {
			rb = is_larger? a->rb + PES_LOW[r] : a->rb - PES_HIGH[r];
			re = (is_larger? a->rb + PES_HIGH[r] : a->rb - PES_LOW[r]) + l_ms;
		} else {
			rb = (is_larger? a->rb + PES_LOW[r] : a->rb - PES_HIGH[r]) - l_ms;
			re = is_larger? a->rb + PES_HIGH[r] : a->rb - PES_LOW[r];
		}
		if (rb < 0) rb = 0;
		if (re > l_pac << 1) re = l_pac << 1;
		if (rb < re) ref = fetch_seq(ix, &rb, (rb + re) >> 1, &re, &rid);
		if (ref && a->rid == rid && re - rb >= OPT_MIN_SEED_LEN) {
			kswr_t aln;
			reg_t b;
			int tmp, xtra;
			xtra = KSW_XSUBO | KSW_XSTART | (l_ms * OPT_A < 250? KSW_XBYTE : 0) | (OPT_MIN_SEED_LEN * OPT_A); /* bwamem_pair.c:150 */
			aln = ksw_align2_u8(l_ms, seq, (int)(re - rb), ref, xtra, cnt);
			memset(&b, 0, sizeof(b));
			if (aln.score >= OPT_MIN_SEED_LEN && aln.qb >= 0) {
				b.rid = a->rid;
				b.is_alt = a->is_alt;
				b.qb = is_rev? l_ms - (aln.qe + 1) : aln.qb;
				b.qe = is_rev? l_ms - aln.qb : aln.qe + 1;
				b.rb = is_rev? (l_pac << 1) - (rb + aln.te + 1) : rb + aln.tb;
				b.re = is_rev? (l_pac << 1) - (rb + aln.tb) : rb + aln.te + 1;
				b.score = aln.score;
				b.csub = aln.score2;
				b.secondary = -1;
				b.seedcov = (int)((b.re - b.rb < b.qe - b.qb? b.re - b.rb : b.qe - b.qb) >> 1);
				rv_pushp(ma);
				for (i = 0; i < ma->n - 1; ++i) if (ma->a[i].score < b.score) break;
				tmp = i;
				for (i = ma->n - 1; i > tmp; --i) ma->a[i] = ma->a[i - 1];
				ma->a[i] = b;
			}
			++n;
		}
		if (n) ma->n = sort_dedup_patch(0, 0, ma->n, ma->a, cnt);
		free(seq); free(ref);
	}
	return n;
}

/* ------------------------------------------------------------------------------------------
 * Region -> alignment record.  ref: bwamem.c:792-799 (infer_bw), :950-979 (mem_approx_mapq_se),
 * :1086-1156 (mem_reg2aln)
 * ------------------------------------------------------------------------------------------ */
typedef struct { int64_t pos; int rid, flag, is_rev, is_alt, mapq, NM, n_cigar; uint32_t *cigar; int score, sub, alt_sc; } aln_t;

static inline int infer_bw(int l1, int l2, int score, int a, int q, int r)
{
	int w;
	if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
	w = (int)((double)((l1 < l2? l1 : l2) * a - score - q) / r + 2.);
	if (w < abs(l1 - l2)) w = abs(l1 - l2);
	return w;
}

static int approx_mapq_se(const reg_t *a)
{
	int mapq, l, sub = a->sub? a->sub : OPT_MIN_SEED_LEN * OPT_A;
	double identity, tmp;
	sub = a->csub > sub? a->csub : sub;
	if (sub >= a->score) return 0;
	l = a->qe - a->qb > a->re - a->rb? a->qe - a->qb : (int)(a->re - a->rb);
	identity = 1. - (double)(l * OPT_A - a->score) / (OPT_A + OPT_B) / l;
	if (a->score == 0) mapq = 0;
	else {
		int coef_fac = (int)log(OPT_MAPQ_COEF_LEN); /* mapQ_coef_fac is declared int in mem_opt_t (bwamem.h:58) */
		tmp = l < OPT_MAPQ_COEF_LEN? 1. : coef_fac / log(l);
		tmp *= identity * identity;
		mapq = (int)(6.02 * (a->score - sub) / OPT_A * tmp * tmp + .499);
	}
	if (a->sub_n > 0) mapq -= (int)(4.343 * log(a->sub_n + 1) + .499);
	if (mapq > 60) mapq = 60;
	if (mapq < 0) mapq = 0;
	mapq = (int)(mapq * (1. - a->frac_rep) + .499);
	return mapq;
}

static aln_t reg2aln(const index_t *ix, int l_query, const uint8_t *query_, const reg_t *ar, ora_counters_t *cnt)
{
	aln_t a;
	int i, w2, tmp, qb, qe, NM = -1, score = 0, is_rev, last_sc = -(1 << 30);
	int64_t pos, rb, re;
	uint8_t *query;
	cigar_t cg = {0,0,0};
	memset(&a, 0, sizeof a);
	if (ar == 0 || ar->rb < 0 || ar->re < 0) { a.rid = -1; a.pos = -1; a.flag |= 0x4; return a; }
	qb = ar->qb; qe = ar->qe; rb = ar->rb; re = ar->re;
	query = (uint8_t*)malloc(l_query + 1);
	memcpy(query, query_, l_query);
	a.mapq = ar->secondary < 0? approx_mapq_se(ar) : 0;
	if (ar->secondary >= 0) a.flag |= 0x100;
	tmp = infer_bw(qe - qb, (int)(re - rb), ar->truesc, OPT_A, OPT_O_DEL, OPT_E_DEL);
	w2 = infer_bw(qe - qb, (int)(re - rb), ar->truesc, OPT_A, OPT_O_INS, OPT_E_INS);
	w2 = w2 > tmp? w2 : tmp;
	if (w2 > OPT_W) w2 = w2 < ar->w? w2 : ar->w;
	i = 0;
	do {
		w2 = w2 < OPT_W << 2? w2 : OPT_W << 2;
		gen_cigar2(ix, w2, qe - qb, &query[qb], rb, re, &score, &cg, &NM, cnt);
		if (score == last_sc || w2 == OPT_W << 2) break;
		last_sc = score;
		w2 <<= 1;
	} while (++i < 3 && score < ar->truesc - OPT_A);
	a.NM = NM;
	pos = depos(ix, rb < ix->l_pac? rb : re - 1, &is_rev);
	a.is_rev = is_rev;
	if (cg.n > 0) { /* squeeze out a leading or trailing deletion */
		if ((cg.a[0] & 0xf) == 2) {
			pos += cg.a[0] >> 4;
			--cg.n;
			memmove(cg.a, cg.a + 1, cg.n * 4);
		} else if ((cg.a[cg.n - 1] & 0xf) == 2) --cg.n;
	}
	if (qb != 0 || qe != l_query) { /* soft clips (op 3) */
		int clip5 = is_rev? l_query - qe : qb, clip3 = is_rev? qb : l_query - qe;
		cg.a = (uint32_t*)realloc(cg.a, 4 * (cg.n + 2));
		if (clip5) { memmove(cg.a + 1, cg.a, cg.n * 4); cg.a[0] = (uint32_t)clip5 << 4 | 3; ++cg.n; }
		if (clip3) cg.a[cg.n++] = (uint32_t)clip3 << 4 | 3;
	}
	a.n_cigar = cg.n; a.cigar = cg.a;
	a.rid = pos2rid(ix, pos);
	a.pos = pos - ix->ann[a.rid].offset;
	a.score = ar->score; a.sub = ar->sub > ar->csub? ar->sub : ar->csub;
	a.is_alt = ar->is_alt; a.alt_sc = ar->alt_sc;
	free(query);
	return a;
}

/* ------------------------------------------------------------------------------------------
 * The pair path as the Go bridge drives it.  ref: gobwa.go:226-337 (GoBwaMemMateSW), :400-415
 * (GoBwaSmithWaterman, called for every candidate by aligner.go:1496-1501)
 * ------------------------------------------------------------------------------------------ */
typedef struct { regvec_t r[2]; aln_t *aln[2]; } pair_res_t;

static void do_pair(const index_t *ix, int l1, const uint8_t *s1, int l2, const uint8_t *s2, int score_delta, pair_res_t *res, ora_counters_t *cnt)
{
	uint8_t *q[2];
	int l[2], i, e, num, best[2] = { 0, 0 };
	l[0] = l1; l[1] = l2;
	q[0] = (uint8_t*)malloc(l1 + 1); q[1] = (uint8_t*)malloc(l2 + 1);
	memcpy(q[0], s1, l1); memcpy(q[1], s2, l2);
	memset(res, 0, sizeof(*res));
	for (e = 0; e < 2; ++e) {
		if (l[e] > 0) res->r[e] = align1_core(ix, l[e], q[e], cnt);
		for (i = 0; i < res->r[e].n; ++i) if (res->r[e].a[i].score > best[e]) best[e] = res->r[e].a[i].score;
	}
	/* read1 is rescued from read2's hits; then read2 from the POST-rescue read1 list, still against the PRE-rescue best1 */
	for (e = 1; e >= 0; --e) {
		int o = 1 - e, n_snap = res->r[e].n;
		const reg_t *snap = res->r[e].a;
		for (i = 0, num = 0; i < n_snap && num < 50 && l[o] > 0; ++i)
			if (snap[i].score >= best[e] - score_delta) { ++num; matesw(ix, &snap[i], l[o], q[o], &res->r[o], cnt); }
	}
	for (e = 0; e < 2; ++e) {
		res->aln[e] = (aln_t*)calloc(res->r[e].n + 1, sizeof(aln_t));
		for (i = 0; i < res->r[e].n; ++i) res->aln[e][i] = reg2aln(ix, l[e], q[e], &res->r[e].a[i], cnt);
	}
	free(q[0]); free(q[1]);
}

static void add_counters(ora_counters_t *d, const ora_counters_t *s)
{
	int64_t *a = (int64_t*)d; const int64_t *b = (const int64_t*)s;
	size_t i;
	for (i = 0; i < sizeof(*d) / 8; ++i) a[i] += b[i];
}

double ora_batch_run(ora_ctx_t *c, int64_t n_pairs, const uint8_t *seqs, const int32_t *lens, int score_delta, int n_threads)
{
	pair_res_t *res = (pair_res_t*)calloc(n_pairs + 1, sizeof(pair_res_t));
	int64_t *off = (int64_t*)malloc((2 * n_pairs + 1) * 8), i, nr, nc;
	struct timespec t0, t1;
	off[0] = 0;
	for (i = 0; i < 2 * n_pairs; ++i) off[i + 1] = off[i] + lens[i];
	free_batch(c);
	clock_gettime(CLOCK_MONOTONIC, &t0);
#ifdef _OPENMP
	if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel
#endif
	{
		ora_counters_t local;
		memset(&local, 0, sizeof local);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
		for (i = 0; i < n_pairs; ++i)
			do_pair(&c->ix, lens[2*i], seqs + off[2*i], lens[2*i+1], seqs + off[2*i+1], score_delta, &res[i], &local);
#ifdef _OPENMP
#pragma omp critical
#endif
		add_counters(&c->cnt, &local);
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	c->cnt.n_reads += 2 * n_pairs;
	c->n_reads = 2 * n_pairs;
	c->reg_off = (int64_t*)malloc((c->n_reads + 1) * 8);
	nr = nc = 0;
	for (i = 0; i < n_pairs; ++i) {
		int e, k;
		for (e = 0; e < 2; ++e) {
			c->reg_off[2*i + e] = nr;
			nr += res[i].r[e].n;
			for (k = 0; k < res[i].r[e].n; ++k) nc += res[i].aln[e][k].n_cigar;
		}
	}
	c->reg_off[c->n_reads] = nr; c->n_regs = nr; c->n_cig = nc; c->cnt.n_regs += nr;
	c->regs = (int64_t*)malloc((nr + 1) * ORA_REG_W * 8);
	c->alns = (int64_t*)malloc((nr + 1) * ORA_ALN_W * 8);
	c->cigars = (uint32_t*)malloc((nc + 1) * 4);
	nr = nc = 0;
	for (i = 0; i < n_pairs; ++i) {
		int e, k, j;
		for (e = 0; e < 2; ++e) {
			for (k = 0; k < res[i].r[e].n; ++k, ++nr) {
				const aln_t *a = &res[i].aln[e][k];
				int64_t *r = c->alns + nr * ORA_ALN_W;
				put_reg(c->regs + nr * ORA_REG_W, &res[i].r[e].a[k]);
				r[0] = a->pos; r[1] = a->rid; r[2] = a->flag; r[3] = a->is_rev; r[4] = a->is_alt; r[5] = a->mapq; r[6] = a->NM;
				r[7] = a->n_cigar; r[8] = nc; r[9] = a->score; r[10] = a->sub; r[11] = a->alt_sc;
				for (j = 0; j < a->n_cigar; ++j) c->cigars[nc++] = a->cigar[j];
				free(a->cigar);
			}
			free(res[i].r[e].a); free(res[i].aln[e]);
		}
	}
	free(res); free(off);
	return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}

void ora_batch_get(ora_ctx_t *c, int64_t *n_reads, int64_t *n_regs, int64_t *n_cig, int64_t **reg_off, int64_t **regs, int64_t **alns, uint32_t **cigars)
{
	*n_reads = c->n_reads; *n_regs = c->n_regs; *n_cig = c->n_cig;
	*reg_off = c->reg_off; *regs = c->regs; *alns = c->alns; *cigars = c->cigars;
}
