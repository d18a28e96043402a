/*
 * oracle/arx_oracle_rfa.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the Go half of Arachne's per-barcode path (/root/reference/src/aligner/aligner.go), working on
 * the flat outputs of the BWA half (region rows, alignment rows, CIGARs):
 *   B2  GetChains                         aligner.go:1633-1715   candidate list per read, placeholder for reads without hits
 *   B3  GetAlignments                     aligner.go:1484-1631   CIGAR walk -> mismatches/indels/soft clips, log_alignment_probability, score filter
 *   R1  tagBestAlignments                 aligner.go:1397-1481   best (read, mate) candidate pair, position lists per contig
 *   R2  inferMolecules / markBestAlignmentForReadInMolecule / scrapMolecules   aligner.go:1300-1393, 991-1016
 *   R3  scoreAlignment / isPair           aligner.go:556-581, 1032-1063
 *   R4  fastScore / isActiveMolecule      aligner.go:1109-1250
 *   R5  Optimize / GenerateMove / acceptMove   optimizer.go:15-27, aligner.go:1065-1097, 1261-1298
 *   R6  estimateMapQualities and helpers  aligner.go:643-922
 *
 * PARITY UNPINNED BY THE REFERENCE for this half: the Go tree does not compile, there is no Go toolchain here and the
 * reference holds no test vectors.  What is restated is the intended semantics listed there.  Three
 * things the Go runtime decides are fixed here by rule and documented as such: (1) exact score ties in tagBestAlignments
 * are broken by a md5-seeded math/rand jitter there, by "first candidate pair wins" here; (2) sort.Sort (pdqsort,
 * unstable) orders equal positions arbitrarily, a stable sort is used here; (3) math.Pow/math.Log10 are Go's own
 * implementations, libm's are used here (int(mapq) can differ when the real value is within an ulp of an integer).
 * All score arithmetic is done in integer half-units (every term is a multiple of 0.5 for an integer improper-pair
 * penalty), so summation order is not observable; the mismatch-location multiset of the reference changes no score
 * (its +-2.0 terms are commented out, aligner.go:1168,1178) and is not kept.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "arx_oracle.h"

typedef struct {
	int reg;            /* index into the region rows, -1 for the placeholder of a read without hits */
	int read, mate;     /* barcode-local read ids (2i, 2i+1) */
	int rid;            /* contig, -1 for "" */
	int64_t pos, aend;
	int reversed, score;
	int mismatches, indels, soft_clipped, soft_clipped_length;
	int lap2;           /* log_alignment_probability in half-units */
	int active, is_proper, mol, best_in_mol, active_molecule;
	double sum_move;
	int mapq;
} cand_t;

typedef struct { int first_read_slot, n_potential, n_active, soft_clipped, active_molecule; double confidence; } mol_t;

/* aligner.go:1032-1063 */
static int is_pair(const cand_t *a, const cand_t *b)
{
	const cand_t *f, *r;
	if (a->reversed == b->reversed || a->rid != b->rid) return 0;
	if (a->reversed) { f = b; r = a; } else { f = a; r = b; }
	int64_t dist = r->pos - f->pos;
	return dist >= -35 && dist < 750;
}

/* scoreAlignment (aligner.go:556-581) without the molecule term, in half-units; pen2 = 2 * improper_pair_penalty */
static int pair_score2(const cand_t *a, const cand_t *m, int pen2)
{
	int s = 0;
	if (a) s += a->lap2;
	if (m) s += m->lap2;
	if (!a || !m || !is_pair(a, m)) s += pen2;
	return s;
}

typedef struct {
	cand_t *c; int n_c;
	int *roff;          /* per read: [roff[r], roff[r+1]) = its FILTERED candidates (score >= best - 17), RFA's `alignments` */
	int n_reads;
	mol_t *mol; int n_mol;
	int *pot_read, *pot_off; /* per molecule: reads that have a candidate in it */
	int pen2;
} bc_t;

static int best_for(const bc_t *b, int mol, int read) /* molecule.best_alignment_for_read.Get(read) */
{
	for (int k = b->roff[read]; k < b->roff[read + 1]; ++k)
		if (b->c[k].mol == mol && b->c[k].best_in_mol) return k;
	return -1;
}
static int active_of(const bc_t *b, int read) /* the read's active candidate */
{
	for (int k = b->roff[read]; k < b->roff[read + 1]; ++k) if (b->c[k].active) return k;
	return -1;
}
static int is_active_molecule(const mol_t *m, int change) /* aligner.go:1239-1250 */
{
	double active = (double)(m->n_active + change), potential = (double)m->n_potential;
	if (active <= 4) return 0;
	if (active / potential < 0.1) return 0;
	return 1;
}

/* fastScore (aligner.go:1109-1237) in half-units; moved[] receives (read, sink candidate) pairs acceptMove would apply */
static int fast_score2(const bc_t *b, int S, int T, int *num_out, int *mv_read, int *mv_sink, int *n_mv)
{
	const mol_t *ms = &b->mol[S], *mt = &b->mol[T];
	int change = 0, ach = 0, num = 0, nmv = 0;
	for (int p = b->pot_off[S]; p < b->pot_off[S + 1]; ++p) {
		int read = b->pot_read[p], sa = active_of(b, read);
		if (sa < 0 || b->c[sa].mol != S) continue; /* only the active alignments of the source */
		int ta = best_for(b, T, read);
		if (ta < 0) continue;
		const cand_t *src = &b->c[sa], *snk = &b->c[ta];
		int mate = src->mate, sm = active_of(b, mate);
		int source_has_mate = sm >= 0 && b->c[sm].mol == S;
		int source_pair = source_has_mate && is_pair(src, &b->c[sm]);
		int tm = best_for(b, T, mate);
		int sink_pair = tm >= 0 && is_pair(snk, &b->c[tm]) && source_has_mate;
		if (!source_pair || (source_has_mate && sink_pair)) { if (mv_read) { mv_read[nmv] = read; mv_sink[nmv] = ta; } ++nmv; }
		ach += snk->lap2 - src->lap2;
		if (source_pair && !sink_pair && S != T) ach += b->pen2 / 2;       /* log_unpaired_probability / 2 */
		else if (!source_pair && sink_pair && S != T) ach -= b->pen2 / 2;
		++num;
	}
	int sb = is_active_molecule(ms, 0), sa2 = is_active_molecule(ms, -num);
	if (!sa2 && sb && S != T) change += ms->n_potential;             /* -= len * -0.5 */
	int tb = is_active_molecule(mt, 0), ta2 = is_active_molecule(mt, num);
	if (ta2 && !tb && S != T) change -= mt->n_potential;             /* += len * -0.5 */
	if (ms->n_active - num == 0 && num > 0 && S != T) change += 6;   /* -= -3.0 */
	if (mt->n_active == 0 && num > 0 && S != T) change -= 6;         /* += -3.0 */
	change += ach;
	*num_out = num;
	if (n_mv) *n_mv = nmv;
	return change;
}

static int cmp_pos(const void *x, const void *y) { const int64_t *a = (const int64_t*)x, *b = (const int64_t*)y; return (a[0] > b[0]) - (a[0] < b[0]) ? (a[0] > b[0]) - (a[0] < b[0]) : (a[1] > b[1]) - (a[1] < b[1]); }

/* output row per candidate (ORA_CAND_W int64): reg read pos aend reversed rid score mismatches indels soft_clipped
 * soft_clipped_length lap2 active is_proper mapq molecule_id active_molecule in_filtered */


/* One barcode.  Region/aln rows are the batch-global ones; r0 = first read of the barcode (even), n_reads reads. */
static void rfa_one_barcode(const int64_t *reg_off, const int64_t *regs, const int64_t *alns, const uint32_t *cigars, const int32_t *lens,
                            int64_t r0, int n_reads, int do_rfa, int pen_int, int64_t l_pac, const int64_t *ann_off,
                            const int64_t *cen_start, const int64_t *cen_end,
                            int64_t *cand_rows, int64_t *cand_off_out, int64_t *n_cand_total)
{
	/* B2 + B3: all candidates ("full"), then the filtered lists */
	int n_full = 0;
	for (int r = 0; r < n_reads; ++r) { int n = (int)(reg_off[r0 + r + 1] - reg_off[r0 + r]); n_full += n ? n : 1; }
	cand_t *full = (cand_t*)calloc(n_full + 1, sizeof(cand_t));
	int *foff = (int*)malloc((n_reads + 1) * sizeof(int));
	int k = 0;
	for (int r = 0; r < n_reads; ++r) {
		int64_t g0 = reg_off[r0 + r], g1 = reg_off[r0 + r + 1];
		foff[r] = k;
		if (g0 == g1) { /* placeholder (aligner.go:1664-1676,1700-1711): pos -1, contig "", score 0, empty CIGAR */
			cand_t *c = &full[k++];
			c->reg = -1; c->read = r; c->mate = r ^ 1; c->rid = -1; c->pos = -1; c->aend = 0; c->sum_move = 1.0; c->mol = -1;
			continue;
		}
		for (int64_t g = g0; g < g1; ++g) {
			const int64_t *rg = regs + g * ORA_REG_W, *al = alns + g * ORA_ALN_W;
			cand_t *c = &full[k++];
			int64_t rb = rg[0], re = rg[1], off = ann_off[rg[4]];
			int64_t cpos = rb < l_pac ? rb - off : 2 * l_pac - 1 - rb - off;      /* InterpretAlign, gobwa.go:351-363 */
			int64_t cend = re < l_pac ? re - off : 2 * l_pac - 1 - re - off;
			int indel_len = 0;
			c->reg = (int)g; c->read = r; c->mate = r ^ 1; c->rid = (int)al[1]; c->score = (int)rg[5]; c->sum_move = 1.0; c->mol = -1;
			c->reversed = (int)al[3];
			for (int j = 0; j < (int)al[7]; ++j) { /* order of the walk does not matter for the counts (aligner.go:1529-1560) */
				uint32_t cg = cigars[al[8] + j]; int op = cg & 0xf, len = cg >> 4;
				if (op == 1 || op == 2) { ++c->indels; indel_len += len; }
				else if (op == 3) { ++c->soft_clipped; c->soft_clipped_length += len; }
			}
			c->mismatches = (int)al[6] - indel_len;
			if (c->mismatches < 0) c->mismatches = 0;
			c->pos = cpos; c->aend = cend;
			if (cpos != -1 && c->reversed) { c->pos = cend + 1; c->aend = cpos + 1; }
			c->lap2 = -4 * c->mismatches - 6 * c->indels - (c->soft_clipped > 0 ? 10 * c->soft_clipped + c->soft_clipped_length : 0);
		}
	}
	foff[n_reads] = k;
	/* filtered lists: score >= best - 17 (aligner.go:1490-1495,1624-1627) */
	bc_t b; memset(&b, 0, sizeof b);
	b.n_reads = n_reads; b.pen2 = 2 * pen_int;
	b.c = (cand_t*)calloc(n_full + 1, sizeof(cand_t));
	b.roff = (int*)malloc((n_reads + 1) * sizeof(int));
	int *src_idx = (int*)malloc((n_full + 1) * sizeof(int));
	for (int r = 0; r < n_reads; ++r) {
		int best = 0;
		b.roff[r] = b.n_c;
		for (int i = foff[r]; i < foff[r + 1]; ++i) if (full[i].score > best) best = full[i].score;
		for (int i = foff[r]; i < foff[r + 1]; ++i) if (full[i].score >= best - 17) { src_idx[b.n_c] = i; b.c[b.n_c++] = full[i]; }
	}
	b.roff[n_reads] = b.n_c;
	/* R1 tagBestAlignments: per pair the best (candidate, mate candidate); ties: first pair wins (reference: random jitter) */
	for (int r = 0; r + 1 < n_reads; r += 2) {
		int bs = 0, ba = -1, bm = -1, first = 1;
		for (int i = b.roff[r]; i < b.roff[r + 1]; ++i)
			for (int j = b.roff[r + 1]; j < b.roff[r + 2]; ++j) {
				int s = pair_score2(&b.c[i], &b.c[j], b.pen2);
				if (first || s > bs) { bs = s; ba = i; bm = j; first = 0; }
			}
		b.c[ba].active = 1; b.c[bm].active = 1;
		if (is_pair(&b.c[ba], &b.c[bm])) { b.c[ba].is_proper = 1; b.c[bm].is_proper = 1; }
	}
	if (do_rfa) {
		/* positions per contig in first-seen order, sorted by pos (stable); R2 inferMolecules: split at gaps > 50 kb */
		int64_t *ord = (int64_t*)malloc((b.n_c + 1) * 3 * sizeof(int64_t));
		int *contig_rank = (int*)malloc((b.n_c + 1) * sizeof(int)), n_contigs = 0, *contig_ids = (int*)malloc((b.n_c + 1) * sizeof(int));
		for (int i = 0; i < b.n_c; ++i) {
			int cr = -1;
			for (int q = 0; q < n_contigs; ++q) if (contig_ids[q] == b.c[i].rid) { cr = q; break; }
			if (cr < 0) { cr = n_contigs; contig_ids[n_contigs++] = b.c[i].rid; }
			contig_rank[i] = cr;
		}
		int *mol_of = (int*)malloc((b.n_c + 1) * sizeof(int)), n_mol = 0;
		int *order = (int*)malloc((b.n_c + 1) * sizeof(int)), n_ord = 0;
		for (int q = 0; q < n_contigs; ++q) {
			int m = 0;
			for (int i = 0; i < b.n_c; ++i) if (contig_rank[i] == q) { ord[2 * m] = b.c[i].pos; ord[2 * m + 1] = i; ++m; }
			qsort(ord, m, 2 * sizeof(int64_t), cmp_pos); /* ties by insertion index = stable */
			for (int t = 0; t < m; ++t) {
				int i = (int)ord[2 * t + 1];
				if (t == 0 || ord[2 * t] - ord[2 * (t - 1)] > 50000) ++n_mol;
				mol_of[i] = n_mol - 1;
				order[n_ord++] = i;
			}
		}
		/* markBestAlignmentForReadInMolecule (aligner.go:1340-1393): reads in first-seen order inside each molecule */
		int *has_active = (int*)calloc(n_mol + 1, sizeof(int));
		for (int i = 0; i < b.n_c; ++i) b.c[i].mol = mol_of[i];
		for (int t = 0; t < n_ord; ++t) {
			int i = order[t], m = mol_of[i], r = b.c[i].read, seen = 0;
			for (int u = 0; u < t && !seen; ++u) if (mol_of[order[u]] == m && b.c[order[u]].read == r) seen = 1;
			if (seen) continue;
			/* candidates of read r in molecule m, in position order */
			int best = -1, first = 1; double dummy = 0; (void)dummy;
			int bs = 0;
			for (int u = t; u < n_ord; ++u) {
				int a = order[u];
				if (mol_of[a] != m || b.c[a].read != r) continue;
				int mate_any = 0;
				for (int v = 0; v < n_ord; ++v) {
					int mc = order[v];
					if (mol_of[mc] != m || b.c[mc].read != b.c[a].mate) continue;
					mate_any = 1;
					int s = pair_score2(&b.c[a], &b.c[mc], b.pen2);
					if (first || s > bs) { bs = s; best = a; first = 0; }
				}
				if (!mate_any && (first || b.c[a].lap2 > bs)) { bs = b.c[a].lap2; best = a; first = 0; }
				if (b.c[a].active) has_active[m] = 1;
			}
			b.c[best].best_in_mol = 1;
		}
		/* scrapMolecules: molecules without an active alignment disappear, the rest are renumbered */
		int *renum = (int*)malloc((n_mol + 1) * sizeof(int)), cnt = 0;
		for (int m = 0; m < n_mol; ++m) renum[m] = has_active[m] ? cnt++ : -1;
		for (int i = 0; i < b.n_c; ++i) { b.c[i].mol = renum[mol_of[i]]; if (b.c[i].mol < 0) b.c[i].best_in_mol = 0; }
		b.n_mol = cnt;
		b.mol = (mol_t*)calloc(cnt + 1, sizeof(mol_t));
		b.pot_off = (int*)calloc(cnt + 2, sizeof(int));
		b.pot_read = (int*)malloc((b.n_c + 1) * sizeof(int));
		for (int i = 0; i < b.n_c; ++i) if (b.c[i].mol >= 0 && b.c[i].best_in_mol) ++b.pot_off[b.c[i].mol + 1];
		for (int m = 0; m < cnt; ++m) b.pot_off[m + 1] += b.pot_off[m];
		{ int *fill = (int*)calloc(cnt + 1, sizeof(int));
		  for (int i = 0; i < b.n_c; ++i) if (b.c[i].mol >= 0 && b.c[i].best_in_mol) { int m = b.c[i].mol; b.pot_read[b.pot_off[m] + fill[m]++] = b.c[i].read; }
		  free(fill); }
		for (int m = 0; m < cnt; ++m) b.mol[m].n_potential = b.pot_off[m + 1] - b.pot_off[m];
		for (int i = 0; i < b.n_c; ++i) if (b.c[i].active && b.c[i].mol >= 0) ++b.mol[b.c[i].mol].n_active;
		/* R5 Optimize(obj, 1, 2, 4*M): 2 sweeps of 4*M greedy moves (optimizer.go:15-27; the acceptance closure is never called) */
		if (cnt > 0) {
			int cur = 0, *mvr = (int*)malloc((n_reads + 1) * sizeof(int)), *mvs = (int*)malloc((n_reads + 1) * sizeof(int));
			int *bvr = (int*)malloc((n_reads + 1) * sizeof(int)), *bvs = (int*)malloc((n_reads + 1) * sizeof(int));
			for (int it = 0; it < 2 * 4 * cnt; ++it) {
				int S = cur;
				cur = (cur + 1) % cnt;
				if (b.mol[S].n_active == 0) continue;
				int have = 0, best_sc = 0, best_T = -1, best_n = 0;
				for (int T = 0; T < cnt; ++T) {
					if (T == S) continue;
					int num, nmv, sc = fast_score2(&b, S, T, &num, mvr, mvs, &nmv);
					if (num > 0 && (!have || sc > best_sc || (sc == best_sc && b.mol[T].n_active > b.mol[best_T].n_active))) {
						have = 1; best_sc = sc; best_T = T; best_n = nmv;
						memcpy(bvr, mvr, nmv * sizeof(int)); memcpy(bvs, mvs, nmv * sizeof(int));
					}
				}
				if (have && (best_sc > 0 || (best_sc == 0 && b.mol[best_T].n_active > b.mol[S].n_active))) {
					for (int q = 0; q < best_n; ++q) { /* acceptMove (aligner.go:1261-1298) */
						int sa = active_of(&b, bvr[q]);
						b.c[sa].active = 0; --b.mol[S].n_active;
						int was_in_sink = 0; (void)was_in_sink;
						b.c[bvs[q]].active = 1; ++b.mol[best_T].n_active;
					}
				}
			}
			free(mvr); free(mvs); free(bvr); free(bvs);
		}
		free(ord); free(contig_rank); free(contig_ids); free(mol_of); free(order); free(has_active); free(renum);
	}
	/* R6 estimateMapQualities (aligner.go:797-922) */
	double log_mol_pen = 0.0;
	if (do_rfa) {
		/* method 2: molecule move probability sums (aligner.go:697-720); 10^(x/2) through the same table the device path uses */
		for (int S = 0; S < b.n_mol; ++S)
			for (int T = 0; T < b.n_mol; ++T) {
				if (S == T) continue;
				int num, sc = fast_score2(&b, S, T, &num, 0, 0, 0);
				double p = pow(10.0, 0.5 * sc);
				for (int q = b.pot_off[S]; q < b.pot_off[S + 1]; ++q) {
					int read = b.pot_read[q], sa = active_of(&b, read);
					if (sa >= 0 && b.c[sa].mol == S && best_for(&b, T, read) >= 0) b.c[sa].sum_move += p;
				}
			}
		/* setMoleculeConfidences + updateAlignmentsMoleculeStatus (aligner.go:957-969, 643-680) */
		for (int m = 0; m < b.n_mol; ++m) b.mol[m].confidence = (double)b.mol[m].n_active / (double)b.mol[m].n_potential;
		for (int i = 0; i < b.n_c; ++i) if (b.c[i].active && b.c[i].mol >= 0 && b.c[i].soft_clipped > 0) ++b.mol[b.c[i].mol].soft_clipped;
		for (int i = 0; i < b.n_c; ++i) if (b.c[i].mol >= 0) {
			mol_t *m = &b.mol[b.c[i].mol];
			int act = m->n_active - m->soft_clipped > 4 && m->confidence > 0.1;
			b.c[i].active_molecule = act;
			if (act) m->active_molecule = 1;
		}
		/* calculateLogMoleculePenalty (aligner.go:722-753) with the hard-coded 3.2 Gbp genome */
		if (b.n_mol > 0) {
			double dna = 1000.0;
			for (int m = 0; m < b.n_mol; ++m) {
				int64_t lo = INT64_MAX, hi = -1;
				for (int i = 0; i < b.n_c; ++i) if (b.c[i].active && b.c[i].mol == m) {
					if (b.mol[m].active_molecule) { if (b.c[i].pos > hi) hi = b.c[i].pos; if (b.c[i].pos < lo) lo = b.c[i].pos; }
					else dna += (double)(b.c[i].aend - b.c[i].pos) * 2.0;
				}
				if (b.mol[m].active_molecule && hi >= lo) dna += (double)(hi - lo) + 1000.0;
			}
			log_mol_pen = log10(dna / 3200000000.0 * 0.05);
		}
	}
	{ /* method 1 per read, then the final MAPQ of the active candidate */
		double pen = 0.5 * b.pen2;
		for (int r = 0; r < n_reads; ++r) {
			int n = b.roff[r + 1] - b.roff[r], mr = r ^ 1, ns = 0;
			double *scores = (double*)malloc((n + 2) * sizeof(double));
			/* pseudo-count alignment paired with the best single mate (aligner.go:682-695, 547-554) */
			double best_single = -1.7976931348623157e308;
			for (int j = b.roff[mr]; j < b.roff[mr + 1]; ++j) {
				double s = 0.5 * b.c[j].lap2 + pen; /* scoreAlignment(nil, mate, pen): no molecule term when aln == nil */
				if (s > best_single) best_single = s;
			}
			double pseudo = -10.0 - ((double)lens[r0 + r] - 25.0) * 0.5 + log_mol_pen;
			scores[ns++] = (b.roff[mr + 1] > b.roff[mr]) ? best_single + pseudo : pseudo;
			for (int i = b.roff[r]; i < b.roff[r + 1]; ++i) {
				double bs = -1.7976931348623157e308;
				for (int j = b.roff[mr]; j < b.roff[mr + 1]; ++j) {
					double s = 0.5 * pair_score2(&b.c[i], &b.c[j], b.pen2) + (b.c[i].active_molecule ? 0.0 : log_mol_pen);
					if (s > bs) bs = s;
				}
				scores[ns++] = bs;
			}
			/* sort ascending, sum the top 15 from the largest down (aligner.go:893-898) */
			for (int x = 1; x < ns; ++x) { double t = scores[x]; int y = x; while (y > 0 && scores[y - 1] > t) { scores[y] = scores[y - 1]; --y; } scores[y] = t; }
			double total = 0.0;
			for (int x = ns - 1; x >= 0 && ns - x <= 15; --x) total += pow(10.0, scores[x]);
			int a = active_of(&b, r), am = active_of(&b, mr);
			double sc = 0.5 * pair_score2(&b.c[a], &b.c[am], b.pen2) + (b.c[a].active_molecule ? 0.0 : log_mol_pen);
			double mapq = -10.0 * log10(1.0 - pow(10.0, sc) / total);
			double mmq = -10.0 * log10(1.0 - (1.0 / b.c[a].sum_move));
			mapq = (mapq != mapq || mmq != mmq) ? NAN : (mapq < mmq ? mapq : mmq);
			mapq = (mapq != mapq) ? NAN : (mapq < 60.0 ? mapq : 60.0);
			if (cen_start && b.c[a].rid >= 0 && b.c[a].pos > cen_start[b.c[a].rid] && b.c[a].pos <= cen_end[b.c[a].rid]) mapq = 0.0;
			b.c[a].mapq = (mapq != mapq) ? (int)0x80000000 : (int)mapq;
			free(scores);
		}
	}
	/* emit: one row per candidate of `full`, flags taken from its filtered copy */
	for (int i = 0; i < n_full; ++i) full[i].mol = -1;
	for (int i = 0; i < b.n_c; ++i) { cand_t t = b.c[i]; full[src_idx[i]] = t; full[src_idx[i]].best_in_mol = 2; }
	for (int r = 0; r < n_reads; ++r) cand_off_out[r] = *n_cand_total + foff[r];
	for (int i = 0; i < n_full; ++i) {
		int64_t *o = cand_rows + (*n_cand_total + i) * ORA_CAND_W;
		const cand_t *c = &full[i];
		o[0] = c->reg; o[1] = r0 + c->read; o[2] = c->pos; o[3] = c->aend; o[4] = c->reversed; o[5] = c->rid; o[6] = c->score;
		o[7] = c->mismatches; o[8] = c->indels; o[9] = c->soft_clipped; o[10] = c->soft_clipped_length; o[11] = c->lap2;
		o[12] = c->active; o[13] = c->is_proper; o[14] = c->mapq; o[15] = c->mol; o[16] = c->active_molecule; o[17] = c->best_in_mol == 2;
	}
	*n_cand_total += n_full;
	free(full); free(foff); free(b.c); free(b.roff); free(src_idx); free(b.mol); free(b.pot_off); free(b.pot_read);
}

/* Whole batch.  bc_pair_off[n_barcodes+1]: pair offsets of the barcodes; do_rfa[b] = worthRunningRFA (aligner.go:1018-1030),
 * decided by the caller from the barcode string; cen_start/cen_end per contig or NULL.  cand_rows must hold
 * (n_regs + n_reads) * ORA_CAND_W int64; cand_off[n_reads + 1].  Returns the number of candidate rows. */
int64_t ora_rfa(int64_t n_reads, const int64_t *reg_off, const int64_t *regs, const int64_t *alns, const uint32_t *cigars, const int32_t *lens,
                int n_barcodes, const int64_t *bc_pair_off, const uint8_t *do_rfa, int penalty, int64_t l_pac, const int64_t *ann_off,
                const int64_t *cen_start, const int64_t *cen_end, int64_t *cand_rows, int64_t *cand_off)
{
	int64_t total = 0;
	for (int bidx = 0; bidx < n_barcodes; ++bidx) {
		int64_t r0 = 2 * bc_pair_off[bidx], r1 = 2 * bc_pair_off[bidx + 1];
		rfa_one_barcode(reg_off, regs, alns, cigars, lens, r0, (int)(r1 - r0), do_rfa[bidx], penalty, l_pac, ann_off, cen_start, cen_end, cand_rows, cand_off + r0, &total);
	}
	cand_off[n_reads] = total;
	return total;
}

/* ------------------------------------------------------------------------------------------------------------------
 * The passes between placement and the BAM records, on the rows ora_rfa wrote:
 *   GetAlignments' CIGAR walk            aligner.go:1505-1570   (matches, mismatchLocs, mismatchReadLocs; readmap_s/_e :1620-1623)
 *   GetSeq                                gobwa.go:50-80
 *   markDuplicates                        aligner.go:598-641
 *   CheckSplitReads / GetSplitAlignment   split.go:31-163
 * sort.Sort in GetSplitAlignment is Go's pdqsort: insertion sort (stable) up to 12 elements, which is restated; above
 * that its order among equal scores is implementation-defined, the stable order is used and order_pinned = 0 reports the
 * reads where that choice can be seen in the result.
 * post rows (ORA_POST_W): qb qe matches n_mm mm_off duplicate; split rows (ORA_SPLIT_W): split mapq is_proper
 * n_split_cand order_pinned second_best2 score2.  Returns the number of mismatch locations (-1: mm_cap too small).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct { int read1, reversed, rid, mrid; int64_t pos, mpos; int read; } dup_t;
static int cmp_dup(const void *x, const void *y)
{
	const dup_t *a = (const dup_t*)x, *b = (const dup_t*)y;
	if (a->read1 != b->read1) return a->read1 - b->read1;
	if (a->reversed != b->reversed) return a->reversed - b->reversed;
	if (a->rid != b->rid) return a->rid - b->rid;
	if (a->mrid != b->mrid) return a->mrid - b->mrid;
	if (a->pos != b->pos) return a->pos < b->pos ? -1 : 1;
	if (a->mpos != b->mpos) return a->mpos < b->mpos ? -1 : 1;
	return a->read - b->read;
}
static int row_is_pair(const int64_t *a, const int64_t *b) /* isPair on candidate rows */
{
	const int64_t *f, *r;
	if (a[4] == b[4] || a[5] != b[5]) return 0;
	if (a[4]) { f = b; r = a; } else { f = a; r = b; }
	return r[2] - f[2] >= -35 && r[2] - f[2] < 750;
}
static int row_pair_score2(const int64_t *a, const int64_t *m, int pen2) { return (int)(a[11] + m[11]) + (row_is_pair(a, m) ? 0 : pen2); }

int64_t ora_post(ora_ctx_t *ctx, int64_t n_reads, const int64_t *regs, const int64_t *alns, const uint32_t *cigars, const uint8_t *bases, const int32_t *lens,
                 int n_barcodes, const int64_t *bc_pair_off, int penalty, const int64_t *ann_off, const int64_t *cen_start, const int64_t *cen_end,
                 const int64_t *cand_rows, const int64_t *cand_off, int64_t *post_rows, int64_t *split_rows, int32_t *mm_ref, int32_t *mm_read, int64_t mm_cap)
{
	static const char two_bit_to_seq[4] = {'A','C','G','T'}, two_bit_to_seq_comp[4] = {'T','G','C','A'};
	int64_t n_cands = cand_off[n_reads], n_mm = 0;
	int64_t *base_off = (int64_t*)malloc((n_reads + 1) * sizeof(int64_t));
	base_off[0] = 0;
	for (int64_t r = 0; r < n_reads; ++r) base_off[r + 1] = base_off[r] + lens[r];
	/* the CIGAR walk, one candidate after the other */
	for (int64_t i = 0; i < n_cands; ++i) {
		const int64_t *c = cand_rows + i * ORA_CAND_W;
		int64_t *o = post_rows + i * ORA_POST_W;
		memset(o, 0, ORA_POST_W * sizeof(int64_t));
		o[4] = n_mm;
		if (c[0] < 0) continue; /* placeholder: no chain, empty alignment */
		const int64_t *rg = regs + c[0] * ORA_REG_W, *al = alns + c[0] * ORA_ALN_W;
		int reversed = (int)c[4], l_read = lens[c[1]], n_cigar = (int)al[7];
		int64_t ref_start = c[2], ref_end = c[3]; /* the row already carries the swap of aligner.go:1513-1516 (= :1577-1582) */
		/* GetSeq */
		int64_t L = ref_end - ref_start, offstart = ref_start + ann_off[c[5]], offend = ref_end + ann_off[c[5]];
		int tid = (int)c[5];
		char *ref_seq = (char*)calloc(L > 0 ? L : 1, 1);
		uint8_t *two = (uint8_t*)malloc(L > 0 ? L : 1);
		ora_fetch_seq(ctx, &offstart, (offstart + offend) >> 1, &offend, &tid, two, L);
		for (int64_t x = 0; x < offend - offstart && x < L; ++x) {
			if (reversed) ref_seq[offend - offstart - x - 1] = two_bit_to_seq_comp[two[x]];
			else ref_seq[x] = two_bit_to_seq[two[x]];
		}
		free(two);
		const uint8_t *rd = bases + base_off[c[1]];
		int matches = 0, indel_length = 0, ref_off = 0, read_off = 0, cnt = 0;
		int k = reversed ? n_cigar - 1 : 0, inc = reversed ? -1 : 1;
		for (; k < n_cigar && k >= 0; k += inc) {
			uint32_t cg = cigars[al[8] + k]; int op = cg & 0xf, len = cg >> 4;
			if (op == 0) {
				matches += len;
				for (int m = 0; m < len; ++m) {
					if (ref_off + m >= L) continue;
					if (read_off + m >= l_read) continue; /* the reference panics here */
					if (ref_seq[ref_off + m] != "ACGTN"[rd[read_off + m]]) {
						if (n_mm >= mm_cap) { free(ref_seq); free(base_off); return -1; }
						mm_ref[n_mm] = reversed ? (int32_t)(ref_end - (ref_off + m)) : (int32_t)(ref_off + ref_start + m);
						mm_read[n_mm] = read_off + m;
						++n_mm; ++cnt;
					}
				}
				ref_off += len; read_off += len;
			} else if (op == 1) { indel_length += len; read_off += len; }
			else if (op == 2) { indel_length += len; ref_off += len; }
			else if (op == 3) read_off += len;
		}
		free(ref_seq);
		o[0] = rg[2]; o[1] = rg[3]; o[2] = matches - ((int)al[6] - indel_length); o[3] = cnt;
	}
	/* the active candidate of every read */
	int64_t *act = (int64_t*)malloc((n_reads + 1) * sizeof(int64_t));
	for (int64_t r = 0; r < n_reads; ++r) {
		act[r] = -1;
		for (int64_t i = cand_off[r]; i < cand_off[r + 1]; ++i) if (cand_rows[i * ORA_CAND_W + 12]) { act[r] = i; break; }
	}
	/* markDuplicates, barcode by barcode */
	for (int b = 0; b < n_barcodes; ++b) {
		int64_t r0 = 2 * bc_pair_off[b], r1 = 2 * bc_pair_off[b + 1];
		dup_t *d = (dup_t*)malloc((r1 - r0 + 1) * sizeof(dup_t));
		int n = 0;
		for (int64_t r = r0; r < r1; ++r) {
			const int64_t *a = cand_rows + act[r] * ORA_CAND_W, *m = cand_rows + act[r ^ 1] * ORA_CAND_W;
			dup_t t; t.read1 = !(r & 1); t.reversed = (int)a[4]; t.rid = (int)a[5]; t.mrid = (int)m[5]; t.pos = a[2]; t.mpos = m[2]; t.read = (int)(r - r0);
			d[n++] = t;
		}
		qsort(d, n, sizeof(dup_t), cmp_dup);
		for (int i = 1; i < n; ++i) { dup_t x = d[i - 1]; x.read = d[i].read; if (cmp_dup(&x, &d[i]) == 0) post_rows[act[r0 + d[i].read] * ORA_POST_W + 5] = 1; }
		free(d);
	}
	/* CheckSplitReads */
	for (int64_t r = 0; r < n_reads; ++r) {
		int64_t *o = split_rows + r * ORA_SPLIT_W;
		memset(o, 0, ORA_SPLIT_W * sizeof(int64_t));
		o[0] = -1; o[4] = 1;
		const int64_t *P = cand_rows + act[r] * ORA_CAND_W, *M = cand_rows + act[r ^ 1] * ORA_CAND_W;
		if (P[2] == -1) continue;
		int64_t Ps = post_rows[act[r] * ORA_POST_W], Pe = post_rows[act[r] * ORA_POST_W + 1];
		if (Ps > Pe) { int64_t t = Ps; Ps = Pe; Pe = t; }
		if (Pe - Ps > lens[r] - 15) continue;
		int nc = (int)(cand_off[r + 1] - cand_off[r]), n = 0;
		int64_t *cs = (int64_t*)malloc((nc + 1) * sizeof(int64_t));
		for (int64_t i = cand_off[r]; i < cand_off[r + 1]; ++i) {
			const int64_t *S = cand_rows + i * ORA_CAND_W;
			if (S[12]) continue;
			if (S[2] == -1) continue;
			int64_t Ss = post_rows[i * ORA_POST_W], Se = post_rows[i * ORA_POST_W + 1], overlap;
			if (Ss > Se) { int64_t t = Ss; Ss = Se; Se = t; }
			if ((Ps < Ss && Pe > Se) || (Ss < Ps && Se > Pe)) continue;
			else if (Ps < Ss) overlap = Pe - Ss;
			else overlap = Se - Ps;
			if (overlap < (Se - Ss) / 2) {
				int proper = row_is_pair(S, M);
				if (S[6] >= 36 || proper) cs[n++] = i;
			}
		}
		o[3] = n;
		if (n == 0) { free(cs); continue; }
		/* insertionSort of Go's sort package with Less(i, j) = score[i] > score[j] */
		for (int i = 1; i < n; ++i)
			for (int j = i; j > 0 && cand_rows[cs[j] * ORA_CAND_W + 6] > cand_rows[cs[j - 1] * ORA_CAND_W + 6]; --j) { int64_t t = cs[j]; cs[j] = cs[j - 1]; cs[j - 1] = t; }
		const int64_t *C0 = cand_rows + cs[0] * ORA_CAND_W;
		double mapq;
		int pen2 = 2 * penalty;
		if (n > 1) {
			const int64_t *C1 = cand_rows + cs[1] * ORA_CAND_W;
			mapq = (double)(C0[6] - C1[6]);
			o[5] = row_pair_score2(P, C1, pen2);
		} else {
			mapq = (double)C0[6];
			o[5] = (int)P[11] + pen2 + 2 * (-10) - (lens[r] - 25);
		}
		if (cen_start && C0[5] >= 0 && C0[2] > cen_start[C0[5]] && C0[2] <= cen_end[C0[5]]) mapq = 0.0;
		if (mapq > 60) mapq = 60;
		o[0] = cs[0]; o[1] = (int)mapq; o[2] = row_is_pair(C0, M); o[6] = row_pair_score2(C0, M, pen2);
		if (n > 12) {
			int tie0 = cand_rows[cs[1] * ORA_CAND_W + 6] == C0[6];
			int tie1 = n > 2 && cand_rows[cs[2] * ORA_CAND_W + 6] == cand_rows[cs[1] * ORA_CAND_W + 6];
			o[4] = !(tie0 || tie1);
		}
		free(cs);
	}
	free(act); free(base_off);
	return n_mm;
}
