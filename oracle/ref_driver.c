/*
 * oracle/ref_driver.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Thin driver around the reference's own BWA 0.7.16a C core, compiled IN PLACE from
 * /root/reference/src/gobwa/bwa (see oracle/Makefile; outputs go to oracle/_ref/).
 * No reference source is copied into this repository: this file is our own code and
 * only #includes the reference translation unit bwamem.c so that its static helpers
 * (mem_collect_intv, smem_aux_init, ...) can be reached for known-answer vectors.
 *
 * What it replays: the exact cgo call sequence of the reference's Go bridge
 *   GoBwaMemMateSW      src/gobwa/gobwa.go:226-337   (mem_align1_core x2, two mem_matesw loops)
 *   GoBwaSmithWaterman  src/gobwa/gobwa.go:400-415   (mem_reg2aln for EVERY candidate reg)
 * plus per-function entry points used to pin the CPU restatement (oracle/arx_oracle.c)
 * and the HIP kernels.
 *
 * All outputs are flat int64 rows so that ctypes/numpy can read them:
 *   REG row  (ARX_REG_W = 20 int64):  rb re qb qe rid score truesc sub alt_sc csub sub_n w
 *                                      seedcov secondary secondary_all seedlen0 n_comp is_alt
 *                                      frac_rep(float bits) 0
 *   ALN row  (ARX_ALN_W = 12 int64):  pos rid flag is_rev is_alt mapq NM n_cigar cigar_off
 *                                      score sub alt_sc
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "bwamem.c" /* reference TU, included from -I/root/reference/src/gobwa/bwa */
#include "ksw.h"

extern int mem_matesw(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const mem_pestat_t pes[4],
                      const mem_alnreg_t *a, int l_ms, const uint8_t *ms, mem_alnreg_v *ma);

#define ARX_REG_W 20
#define ARX_ALN_W 12

typedef struct {
	bwaidx_t *idx;
	mem_opt_t *opt;
	/* batch results */
	int64_t n_reads;
	int64_t *reg_off; /* n_reads+1 */
	int64_t *regs;    /* n_regs * ARX_REG_W */
	int64_t *alns;    /* n_regs * ARX_ALN_W */
	uint32_t *cigars;
	int64_t n_regs, n_cig;
} ref_ctx_t;

int ref_index_build(const char *fa, const char *prefix)
{
	bwa_verbose = 1;
	return bwa_idx_build(fa, prefix, 0, -1);
}

ref_ctx_t *ref_open(const char *prefix)
{
	ref_ctx_t *c;
	bwaidx_t *idx;
	bwa_verbose = 1;
	idx = bwa_idx_load(prefix, BWA_IDX_ALL);
	if (idx == 0) return 0;
	c = (ref_ctx_t*)calloc(1, sizeof(ref_ctx_t));
	c->idx = idx;
	c->opt = mem_opt_init();
	return c;
}

static void ref_free_batch(ref_ctx_t *c)
{
	free(c->reg_off); free(c->regs); free(c->alns); free(c->cigars);
	c->reg_off = 0; c->regs = 0; c->alns = 0; c->cigars = 0; c->n_regs = c->n_cig = c->n_reads = 0;
}

void ref_close(ref_ctx_t *c)
{
	if (c == 0) return;
	ref_free_batch(c);
	bwa_idx_destroy(c->idx);
	free(c->opt);
	free(c);
}

int64_t ref_l_pac(ref_ctx_t *c) { return c->idx->bns->l_pac; }
int64_t ref_seq_len(ref_ctx_t *c) { return c->idx->bwt->seq_len; }
int64_t ref_primary(ref_ctx_t *c) { return c->idx->bwt->primary; }
int ref_n_seqs(ref_ctx_t *c) { return c->idx->bns->n_seqs; }

/* ---------------- per-function known-answer entry points ---------------- */

void ref_occ4(ref_ctx_t *c, int n, const uint64_t *k, uint64_t *out)
{
	int i;
	for (i = 0; i < n; ++i) bwt_occ4(c->idx->bwt, k[i], out + 4 * i);
}

void ref_extend(ref_ctx_t *c, int n, const uint64_t *ik3, int is_back, uint64_t *ok12)
{
	int i, j;
	for (i = 0; i < n; ++i) {
		bwtintv_t ik, ok[4];
		ik.x[0] = ik3[3*i]; ik.x[1] = ik3[3*i+1]; ik.x[2] = ik3[3*i+2]; ik.info = 0;
		bwt_extend(c->idx->bwt, &ik, ok, is_back);
		for (j = 0; j < 4; ++j) {
			ok12[12*i + 3*j] = ok[j].x[0]; ok12[12*i + 3*j + 1] = ok[j].x[1]; ok12[12*i + 3*j + 2] = ok[j].x[2];
		}
	}
}

void ref_sa(ref_ctx_t *c, int n, const uint64_t *k, uint64_t *out)
{
	int i;
	for (i = 0; i < n; ++i) out[i] = bwt_sa(c->idx->bwt, k[i]);
}

/* mem_collect_intv (bwamem.c:114): out rows of 4 uint64 {x0,x1,x2,info}; returns count (may exceed cap) */
int ref_collect_intv(ref_ctx_t *c, int len, const uint8_t *seq, uint64_t *out, int cap)
{
	smem_aux_t *a = smem_aux_init();
	int i, n;
	mem_collect_intv(c->opt, c->idx->bwt, len, seq, a);
	n = a->mem.n;
	for (i = 0; i < n && i < cap; ++i) {
		out[4*i] = a->mem.a[i].x[0]; out[4*i+1] = a->mem.a[i].x[1]; out[4*i+2] = a->mem.a[i].x[2]; out[4*i+3] = a->mem.a[i].info;
	}
	smem_aux_destroy(a);
	return n;
}

/* mem_chain (+ optional mem_chain_flt). chain rows: 8 int64 {pos rid n_seeds seed_off w kept first is_alt};
 * seed rows: 4 int64 {rbeg qbeg len score}. frac_rep returned through *frac_rep_bits. */
int ref_chains(ref_ctx_t *c, int len, const uint8_t *seq, int do_flt, int64_t *chains, int cap_c, int64_t *seeds, int cap_s, int *n_seeds_out, uint32_t *frac_rep_bits)
{
	mem_chain_v chn;
	int i, j, ns = 0, n;
	uint8_t *s = (uint8_t*)malloc(len);
	memcpy(s, seq, len);
	chn = mem_chain(c->opt, c->idx->bwt, c->idx->bns, len, s, 0);
	if (do_flt) chn.n = mem_chain_flt(c->opt, chn.n, chn.a);
	n = chn.n;
	*frac_rep_bits = 0;
	for (i = 0; i < n; ++i) {
		mem_chain_t *p = &chn.a[i];
		if (i == 0) memcpy(frac_rep_bits, &p->frac_rep, 4);
		if (i < cap_c) {
			int64_t *r = chains + 8 * i;
			r[0] = p->pos; r[1] = p->rid; r[2] = p->n; r[3] = ns; r[4] = p->w; r[5] = p->kept; r[6] = p->first; r[7] = p->is_alt;
		}
		for (j = 0; j < p->n; ++j, ++ns)
			if (ns < cap_s) {
				int64_t *r = seeds + 4 * ns;
				r[0] = p->seeds[j].rbeg; r[1] = p->seeds[j].qbeg; r[2] = p->seeds[j].len; r[3] = p->seeds[j].score;
			}
		free(p->seeds);
	}
	free(chn.a); free(s);
	*n_seeds_out = ns;
	return n;
}

/* ksw_extend2 (ksw.c:380) with BWA-MEM's fixed scoring; out = {score,qle,tle,gtle,gscore,max_off} */
void ref_ksw_extend2(ref_ctx_t *c, int qlen, const uint8_t *q, int tlen, const uint8_t *t, int w, int end_bonus, int zdrop, int h0, int *out)
{
	const mem_opt_t *o = c->opt;
	out[0] = ksw_extend2(qlen, q, tlen, t, 5, o->mat, o->o_del, o->e_del, o->o_ins, o->e_ins, w, end_bonus, zdrop, h0, &out[1], &out[2], &out[3], &out[4], &out[5]);
}

/* ksw_align2 (ksw.c:343); out = {score,te,qe,score2,te2,tb,qb} */
void ref_ksw_align2(ref_ctx_t *c, int qlen, const uint8_t *q, int tlen, const uint8_t *t, int xtra, int *out)
{
	const mem_opt_t *o = c->opt;
	uint8_t *qq = (uint8_t*)malloc(qlen), *tt = (uint8_t*)malloc(tlen);
	kswr_t r;
	memcpy(qq, q, qlen); memcpy(tt, t, tlen);
	r = ksw_align2(qlen, qq, tlen, tt, 5, o->mat, o->o_del, o->e_del, o->o_ins, o->e_ins, xtra, 0);
	out[0] = r.score; out[1] = r.te; out[2] = r.qe; out[3] = r.score2; out[4] = r.te2; out[5] = r.tb; out[6] = r.qb;
	free(qq); free(tt);
}

/* ksw_global2 (ksw.c:504); returns n_cigar */
int ref_ksw_global2(ref_ctx_t *c, int qlen, const uint8_t *q, int tlen, const uint8_t *t, int w, int *score, uint32_t *cigar, int cap)
{
	const mem_opt_t *o = c->opt;
	int n_cigar = 0, i;
	uint32_t *cg = 0;
	*score = ksw_global2(qlen, q, tlen, t, 5, o->mat, o->o_del, o->e_del, o->o_ins, o->e_ins, w, &n_cigar, &cg);
	for (i = 0; i < n_cigar && i < cap; ++i) cigar[i] = cg[i];
	free(cg);
	return n_cigar;
}

static void put_reg(int64_t *r, const mem_alnreg_t *p)
{
	uint32_t fb;
	memcpy(&fb, &p->frac_rep, 4);
	r[0] = p->rb; r[1] = p->re; r[2] = p->qb; r[3] = p->qe; r[4] = p->rid; r[5] = p->score; r[6] = p->truesc;
	r[7] = p->sub; r[8] = p->alt_sc; r[9] = p->csub; r[10] = p->sub_n; r[11] = p->w; r[12] = p->seedcov;
	r[13] = p->secondary; r[14] = p->secondary_all; r[15] = p->seedlen0; r[16] = p->n_comp; r[17] = p->is_alt;
	r[18] = fb; r[19] = 0;
}

/* mem_align1_core (bwamem.c:1048) on a 2-bit read; returns #regs */
int ref_align1(ref_ctx_t *c, int len, const uint8_t *seq, int64_t *regs, int cap)
{
	mem_alnreg_v v;
	int i, n;
	char *s = (char*)malloc(len + 1);
	memcpy(s, seq, len);
	v = mem_align1_core(c->opt, c->idx->bwt, c->idx->bns, c->idx->pac, len, s, 0);
	n = v.n;
	for (i = 0; i < n && i < cap; ++i) put_reg(regs + (size_t)i * ARX_REG_W, &v.a[i]);
	free(v.a); free(s);
	return n;
}

/* bns_fetch_seq (bntseq.c:421): returns length, writes clamped beg/end/rid */
int64_t ref_fetch_seq(ref_ctx_t *c, int64_t *beg, int64_t mid, int64_t *end, int *rid, uint8_t *out, int64_t cap)
{
	uint8_t *s = bns_fetch_seq(c->idx->bns, c->idx->pac, beg, mid, end, rid);
	int64_t n = *end - *beg, i;
	for (i = 0; i < n && i < cap; ++i) out[i] = s[i];
	free(s);
	return n;
}

/* ---------------- the per-pair path, as the Go bridge drives it ---------------- */

typedef struct {
	mem_alnreg_v r[2];
	mem_aln_t *aln[2];
} pair_res_t;

static void fixed_pes(mem_pestat_t pes[4]) /* gobwa.go:229-237 */
{
	memset(pes, 0, 4 * sizeof(mem_pestat_t));
	pes[0].failed = 1;
	pes[1].low = -35; pes[1].high = 500; pes[1].failed = 0; pes[1].avg = 200.0; pes[1].std = 100.0;
	pes[2].failed = 1; pes[3].failed = 1;
}

static void do_pair(const ref_ctx_t *c, int l1, const uint8_t *s1, int l2, const uint8_t *s2, int score_delta, pair_res_t *res)
{
	const bwaidx_t *idx = c->idx;
	mem_pestat_t pes[4];
	char *q1 = (char*)malloc(l1 + 1), *q2 = (char*)malloc(l2 + 1);
	int i, num, best1 = 0, best2 = 0, n_snap;
	mem_alnreg_t *snap; /* Go keeps pointers into the list being rescued FROM; that list is stable while the OTHER one grows */
	fixed_pes(pes);
	memcpy(q1, s1, l1); memcpy(q2, s2, l2);
	memset(res, 0, sizeof(*res));
	if (l1 > 0) res->r[0] = mem_align1_core(c->opt, idx->bwt, idx->bns, idx->pac, l1, q1, 0);
	if (l2 > 0) res->r[1] = mem_align1_core(c->opt, idx->bwt, idx->bns, idx->pac, l2, q2, 0);
	for (i = 0; i < (int)res->r[0].n; ++i) if (res->r[0].a[i].score > best1) best1 = res->r[0].a[i].score;
	for (i = 0; i < (int)res->r[1].n; ++i) if (res->r[1].a[i].score > best2) best2 = res->r[1].a[i].score;
	/* rescue read1 from read2's hits (gobwa.go:285-300) */
	n_snap = res->r[1].n; snap = res->r[1].a;
	for (i = 0, num = 0; i < n_snap && num < 50 && l1 > 0; ++i)
		if (snap[i].score >= best2 - score_delta) {
			++num;
			mem_matesw(c->opt, idx->bns, idx->pac, pes, &snap[i], l1, (uint8_t*)q1, &res->r[0]);
		}
	/* rescue read2 from the post-rescue read1 list, threshold still the pre-rescue best1 (gobwa.go:302-324) */
	n_snap = res->r[0].n; snap = res->r[0].a;
	for (i = 0, num = 0; i < n_snap && num < 50 && l2 > 0; ++i)
		if (snap[i].score >= best1 - score_delta) {
			++num;
			mem_matesw(c->opt, idx->bns, idx->pac, pes, &snap[i], l2, (uint8_t*)q2, &res->r[1]);
		}
	/* GetAlignments -> GoBwaSmithWaterman -> mem_reg2aln for every reg (aligner.go:1496-1501) */
	for (i = 0; i < 2; ++i) {
		int k, l = i? l2 : l1;
		const char *q = i? q2 : q1;
		res->aln[i] = (mem_aln_t*)calloc(res->r[i].n + 1, sizeof(mem_aln_t));
		for (k = 0; k < (int)res->r[i].n; ++k)
			res->aln[i][k] = mem_reg2aln(c->opt, idx->bns, idx->pac, l, q, &res->r[i].a[k]);
	}
	free(q1); free(q2);
}

/* seqs: concatenated 2-bit (0..4) reads, lens[2*n_pairs], read 2i = R1, 2i+1 = R2. Returns wall seconds. */
double ref_batch_run(ref_ctx_t *c, int64_t n_pairs, const uint8_t *seqs, const int32_t *lens, int score_delta, int n_threads)
{
	pair_res_t *res = (pair_res_t*)calloc(n_pairs, sizeof(pair_res_t));
	int64_t *off = (int64_t*)malloc((2 * n_pairs + 1) * 8), i, nr, nc;
	struct timespec t0, t1;
	off[0] = 0;
	for (i = 0; i < 2 * n_pairs; ++i) off[i + 1] = off[i] + lens[i];
	ref_free_batch(c);
	clock_gettime(CLOCK_MONOTONIC, &t0);
#ifdef _OPENMP
	if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 16)
#endif
	for (i = 0; i < n_pairs; ++i)
		do_pair(c, lens[2*i], seqs + off[2*i], lens[2*i+1], seqs + off[2*i+1], score_delta, &res[i]);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	/* flatten */
	c->n_reads = 2 * n_pairs;
	c->reg_off = (int64_t*)malloc((c->n_reads + 1) * 8);
	nr = nc = 0;
	for (i = 0; i < n_pairs; ++i) {
		int e, k;
		for (e = 0; e < 2; ++e) {
			c->reg_off[2*i + e] = nr;
			nr += res[i].r[e].n;
			for (k = 0; k < (int)res[i].r[e].n; ++k) nc += res[i].aln[e][k].n_cigar;
		}
	}
	c->reg_off[c->n_reads] = nr;
	c->n_regs = nr; c->n_cig = nc;
	c->regs = (int64_t*)malloc((nr + 1) * ARX_REG_W * 8);
	c->alns = (int64_t*)malloc((nr + 1) * ARX_ALN_W * 8);
	c->cigars = (uint32_t*)malloc((nc + 1) * 4);
	nr = nc = 0;
	for (i = 0; i < n_pairs; ++i) {
		int e, k, j;
		for (e = 0; e < 2; ++e) {
			for (k = 0; k < (int)res[i].r[e].n; ++k, ++nr) {
				const mem_aln_t *a = &res[i].aln[e][k];
				int64_t *r = c->alns + nr * ARX_ALN_W;
				put_reg(c->regs + nr * ARX_REG_W, &res[i].r[e].a[k]);
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

/* ---- per-phase split of the CPU time (seed / extend / rescue / CIGAR), for bench.py's cpu_baseline only.  The same
 * reference functions in the same order as do_pair(); mem_align1_core (bwamem.c:1048-1084) is replayed call by call so that the
 * clock can be read between seeding (mem_chain .. mem_flt_chained_seeds) and extension (mem_chain2aln .. mem_sort_dedup_patch).
 * Results are thrown away; out[4] = thread-seconds summed over the threads, out[4] = regions found (cross-check). */
static inline double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static mem_alnreg_v timed_align1(const ref_ctx_t *c, int l_seq, char *seq, double *t_seed, double *t_ext)
{
	const mem_opt_t *opt = c->opt;
	const bwaidx_t *idx = c->idx;
	mem_chain_v chn;
	mem_alnreg_v regs;
	int i;
	double t0 = now_s(), t1;
	chn = mem_chain(opt, idx->bwt, idx->bns, l_seq, (uint8_t*)seq, 0);
	chn.n = mem_chain_flt(opt, chn.n, chn.a);
	mem_flt_chained_seeds(opt, idx->bns, idx->pac, l_seq, (uint8_t*)seq, chn.n, chn.a);
	t1 = now_s(); *t_seed += t1 - t0;
	kv_init(regs);
	for (i = 0; i < (int)chn.n; ++i) {
		mem_chain2aln(opt, idx->bns, idx->pac, l_seq, (uint8_t*)seq, &chn.a[i], &regs);
		free(chn.a[i].seeds);
	}
	free(chn.a);
	regs.n = mem_sort_dedup_patch(opt, idx->bns, idx->pac, (uint8_t*)seq, regs.n, regs.a);
	for (i = 0; i < (int)regs.n; ++i)
		if (regs.a[i].rid >= 0 && idx->bns->anns[regs.a[i].rid].is_alt) regs.a[i].is_alt = 1;
	*t_ext += now_s() - t1;
	return regs;
}

double ref_phase_split(ref_ctx_t *c, int64_t n_pairs, const uint8_t *seqs, const int32_t *lens, int score_delta, int n_threads, double *out)
{
	int64_t *off = (int64_t*)malloc((2 * n_pairs + 1) * 8), i;
	double T[4] = {0, 0, 0, 0}, n_regs = 0, w0, w1;
	off[0] = 0;
	for (i = 0; i < 2 * n_pairs; ++i) off[i + 1] = off[i] + lens[i];
	w0 = now_s();
#ifdef _OPENMP
	if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 16) reduction(+:T[:4], n_regs)
#endif
	for (i = 0; i < n_pairs; ++i) {
		const int l1 = lens[2*i], l2 = lens[2*i+1];
		mem_pestat_t pes[4];
		mem_alnreg_v r[2];
		char *q[2];
		int k, e, num, best[2] = {0, 0}, n_snap;
		mem_alnreg_t *snap;
		double t0;
		fixed_pes(pes);
		q[0] = (char*)malloc(l1 + 1); q[1] = (char*)malloc(l2 + 1);
		memcpy(q[0], seqs + off[2*i], l1); memcpy(q[1], seqs + off[2*i+1], l2);
		memset(r, 0, sizeof r);
		if (l1 > 0) r[0] = timed_align1(c, l1, q[0], &T[0], &T[1]);
		if (l2 > 0) r[1] = timed_align1(c, l2, q[1], &T[0], &T[1]);
		for (e = 0; e < 2; ++e) for (k = 0; k < (int)r[e].n; ++k) if (r[e].a[k].score > best[e]) best[e] = r[e].a[k].score;
		t0 = now_s();
		n_snap = r[1].n; snap = r[1].a;
		for (k = 0, num = 0; k < n_snap && num < 50 && l1 > 0; ++k)
			if (snap[k].score >= best[1] - score_delta) { ++num; mem_matesw(c->opt, c->idx->bns, c->idx->pac, pes, &snap[k], l1, (uint8_t*)q[0], &r[0]); }
		n_snap = r[0].n; snap = r[0].a;
		for (k = 0, num = 0; k < n_snap && num < 50 && l2 > 0; ++k)
			if (snap[k].score >= best[0] - score_delta) { ++num; mem_matesw(c->opt, c->idx->bns, c->idx->pac, pes, &snap[k], l2, (uint8_t*)q[1], &r[1]); }
		T[2] += now_s() - t0;
		t0 = now_s();
		for (e = 0; e < 2; ++e)
			for (k = 0; k < (int)r[e].n; ++k) {
				mem_aln_t a = mem_reg2aln(c->opt, c->idx->bns, c->idx->pac, e ? l2 : l1, q[e], &r[e].a[k]);
				free(a.cigar);
			}
		T[3] += now_s() - t0;
		n_regs += r[0].n + r[1].n;
		free(r[0].a); free(r[1].a); free(q[0]); free(q[1]);
	}
	w1 = now_s();
	for (i = 0; i < 4; ++i) out[i] = T[i];
	out[4] = n_regs;
	free(off);
	return w1 - w0;
}

void ref_batch_get(ref_ctx_t *c, int64_t *n_reads, int64_t *n_regs, int64_t *n_cig, int64_t **reg_off, int64_t **regs, int64_t **alns, uint32_t **cigars)
{
	*n_reads = c->n_reads; *n_regs = c->n_regs; *n_cig = c->n_cig;
	*reg_off = c->reg_off; *regs = c->regs; *alns = c->alns; *cigars = c->cigars;
}
