// dev_rfa.h -- the Go half of the per-barcode path on the device: candidate post-processing (GetChains/GetAlignments),
// best-pair tagging, molecule inference, the RFA joint-placement sweep and the molecule-move probability sums.
// One barcode per workgroup; all state of a barcode lives in its slices of batch-wide pools in HBM (L2-resident while it runs).
//
// Reference: /root/reference/src/aligner/aligner.go -- B2 GetChains :1633, B3 GetAlignments :1484, R1 tagBestAlignments :1397,
// R2 inferMolecules :1300 / markBestAlignmentForReadInMolecule :1340 / scrapMolecules :991, R3 scoreAlignment :556 / isPair :1032,
// R4 fastScore :1109 / isActiveMolecule :1239, R5 optimizer.Optimize (optimizer.go:15) + GenerateMove :1065 + acceptMove :1261,
// R6 moleculeMapqProbabilitySums :697 / updateAlignmentsMoleculeStatus :643 / calculateLogMoleculePenalty :722.
//
// Every score term is a multiple of 0.5 for an integer improper-pair penalty, so scores are int32 half-units and the
// order of summation is not observable.  What stays on the host is floating point only (method-1 normalisation, log10,
// final MAPQ; api_impl.h).  Rules that replace Go-runtime behaviour the reference leaves unpinned (tie jitter, unstable
// sort of equal positions) are the ones stated in oracle/arx_oracle_rfa.c, which this code must match bit for bit.
#pragma once
#include <math.h>
#include "arx_dev.h"

namespace arx {

struct Cand { // one candidate alignment of a read (aligner.go:65-114, the fields the path needs)
	int64_t pos, aend;
	double sum_move;    // 1 + sum over sink molecules of 10^fastScore (method 2 of estimateMapQualities)
	int32_t reg;        // index of the region/alignment record it came from, -1 for the placeholder of a read without hits
	int32_t read;       // batch-global read id
	int32_t rid, reversed, score, mismatches, indels, soft_clipped, soft_clipped_length;
	int32_t lap2;       // log_alignment_probability in half-units
	int32_t active, is_proper, mapq, mol, active_molecule, in_filtered, best_in_mol, pad;
};

struct RfaBarcodeOut { double dna_len; int32_t n_mol, pad; };

constexpr int RFA_P10_HALF = 1400; // table of 10^(x/2) for x in [-1400, 1400]; beyond it pow() underflows to 0 / overflows to inf

ARX_HDI bool cand_is_pair(const Cand &a, const Cand &b) // isPair
{
	if (a.reversed == b.reversed || a.rid != b.rid) return false;
	const int64_t dist = a.reversed ? a.pos - b.pos : b.pos - a.pos; // reverse.pos - forward.pos
	return dist >= -35 && dist < 750;
}
ARX_HDI int cand_pair_score2(const Cand &a, const Cand &m, int pen2) // scoreAlignment without the molecule term
{
	return a.lap2 + m.lap2 + (cand_is_pair(a, m) ? 0 : pen2);
}

// B2 + B3 for one read: one candidate per region (or the placeholder), statistics from the CIGAR, score filter
ARX_DEV void cand_build_read(const IndexView &ix, int read, const Reg *regs, const Aln *alns, const uint32_t *cig, int cig_w, int g0, int n_regs, int c0, Cand *out)
{
	if (n_regs == 0) { // aligner.go:1664-1676,1700-1711
		Cand c = Cand();
		c.pos = -1; c.aend = 0; c.sum_move = 1.0; c.reg = -1; c.read = read; c.rid = -1; c.mol = -1; c.in_filtered = 1;
		out[0] = c;
		return;
	}
	int best = 0;
	for (int i = 0; i < n_regs; ++i) if (regs[g0 + i].score > best) best = regs[g0 + i].score;
	for (int i = 0; i < n_regs; ++i) {
		const Reg &rg = regs[g0 + i];
		const Aln &al = alns[g0 + i];
		Cand c = Cand();
		const int64_t off = ix.ann_off[rg.rid];
		const int64_t cpos = rg.rb < ix.l_pac ? rg.rb - off : 2 * ix.l_pac - 1 - rg.rb - off; // InterpretAlign, gobwa.go:351-363
		const int64_t cend = rg.re < ix.l_pac ? rg.re - off : 2 * ix.l_pac - 1 - rg.re - off;
		int indel_len = 0;
		c.reg = c0 + i; /* index into the dense arrays arx_batch_fetch hands out */ c.read = read; c.rid = al.rid; c.score = rg.score; c.reversed = al.is_rev; c.sum_move = 1.0; c.mol = -1;
		const uint32_t *cg = cig + (size_t)(g0 + i) * cig_w;
		for (int j = 0; j < al.n_cigar; ++j) {
			const int op = cg[j] & 0xf, len = (int)(cg[j] >> 4);
			if (op == 1 || op == 2) { ++c.indels; indel_len += len; }
			else if (op == 3) { ++c.soft_clipped; c.soft_clipped_length += len; }
		}
		c.mismatches = al.NM - indel_len;
		if (c.mismatches < 0) c.mismatches = 0;
		c.pos = cpos; c.aend = cend;
		if (cpos != -1 && c.reversed) { c.pos = cend + 1; c.aend = cpos + 1; }
		c.lap2 = -4 * c.mismatches - 6 * c.indels - (c.soft_clipped > 0 ? 10 * c.soft_clipped + c.soft_clipped_length : 0);
		c.in_filtered = rg.score >= best - 17;
		out[i] = c;
	}
}

// ---- estimateMapQualities' floating-point tail for one read (aligner.go:797-922): method 1 normalises the read's pair
// scores over the 15 largest (plus the pseudo-count of an unseen placement), method 2 comes from sum_move; min, cap at 60.
// Compiled for both sides: the device evaluates it for every read with its own pow/log10; where the value is so close to
// an integer that a few ulp could change int(mapq), the host re-evaluates it with libm (pipeline_rfa.h), so that the
// result is the host-libm one everywhere.  Returns the value before the centromere mask and the int conversion.
ARX_HDI void rfa_top15_push(double *top, int &nt, double v) // keeps the 15 largest, largest first
{
	int at = nt;
	while (at > 0 && top[at - 1] < v) --at;
	if (at >= 15) return;
	const int last = nt < 15 ? nt : 14;
	for (int k = last; k > at; --k) top[k] = top[k - 1];
	top[at] = v;
	if (nt < 15) ++nt;
}
// Best pair score (half-units) of candidate i over the mate's filtered candidates [m_lo, m_hi); false: the mate has none.
// For a fixed candidate i the molecule term is a constant and x -> 0.5 * x + t is monotone in floating point, so the best pair
// score of i is the same expression on the INTEGER maximum of cand_pair_score2 over the mate's filtered candidates.  A read in a
// 200-copy repeat has ~190 x 190 pairs: the mate's fields (32 of a candidate's 96 bytes decide a pair) are read four candidates
// per round trip, their loads independent of each other.
ARX_HDI bool rfa_pair_best2(const Cand *c, int i, int m_lo, int m_hi, int pen2, int *best2_out)
{
	const int64_t ipos = c[i].pos; const int irid = c[i].rid, irev = c[i].reversed, ilap = c[i].lap2;
	bool any = false;
	int best2 = 0;
	for (int j0 = m_lo; j0 < m_hi; j0 += 4) {
		int64_t jpos[4]; int jrid[4], jrev[4], jlap[4], jflt[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) { const Cand &m = c[j0 + u < m_hi ? j0 + u : m_hi - 1]; jpos[u] = m.pos; jrid[u] = m.rid; jrev[u] = m.reversed; jlap[u] = m.lap2; jflt[u] = m.in_filtered; }
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			if (j0 + u >= m_hi || !jflt[u]) continue;
			bool pair = false;
			if (irev != jrev[u] && irid == jrid[u]) { const int64_t dist = irev ? ipos - jpos[u] : jpos[u] - ipos; pair = dist >= -35 && dist < 750; } // cand_is_pair
			const int v = ilap + jlap[u] + (pair ? 0 : pen2);
			if (!any || v > best2) { best2 = v; any = true; }
		}
	}
	*best2_out = best2;
	return any;
}
constexpr int RFA_NO_PAIR = (int)0x80000000;
// pair_best (may be null): rfa_pair_best2 of every candidate, RFA_NO_PAIR for none -- computed one candidate per lane (KMapqPair) so that
// the read's own lane is left with the linear part
ARX_HDI double rfa_mapq_value(const Cand *c, int r_lo, int r_hi, int m_lo, int m_hi, int len_r, double log_mol_pen, int penalty, int *a_out, double *largest,
                              const int32_t *pair_best = nullptr)
{
	const double pen = (double)penalty, NEG = -1.7976931348623157e308;
	const int pen2 = 2 * penalty;
	double top[15];
	int nt = 0, a = -1, am = -1;
	double best_single = NEG;
	for (int j = m_lo; j < m_hi; ++j) {
		if (!c[j].in_filtered) continue;
		const double s = 0.5 * c[j].lap2 + pen;
		if (s > best_single) best_single = s;
	}
	const double pseudo = -10.0 - ((double)len_r - 25.0) * 0.5 + log_mol_pen;
	rfa_top15_push(top, nt, best_single + pseudo);
	for (int j = m_lo; j < m_hi; ++j) if (c[j].in_filtered && c[j].active) am = j; // the last one, as the reference's loop leaves it
	for (int i = r_lo; i < r_hi; ++i) {
		if (!c[i].in_filtered) continue;
		if (c[i].active) a = i;
		int best2 = 0;
		bool any;
		if (pair_best) { best2 = pair_best[i]; any = best2 != RFA_NO_PAIR; }
		else any = rfa_pair_best2(c, i, m_lo, m_hi, pen2, &best2);
		const double bs = any ? 0.5 * best2 + (c[i].active_molecule ? 0.0 : log_mol_pen) : NEG;
		rfa_top15_push(top, nt, bs);
	}
	double total = 0.0;
	for (int x = 0; x < nt; ++x) total += pow(10.0, top[x]);
	const double sc = 0.5 * cand_pair_score2(c[a], c[am], pen2) + (c[a].active_molecule ? 0.0 : log_mol_pen);
	double mapq = -10.0 * log10(1.0 - pow(10.0, sc) / total);
	const double mmq = -10.0 * log10(1.0 - (1.0 / c[a].sum_move));
	mapq = (mapq != mapq || mmq != mmq) ? mapq + mmq : (mapq < mmq ? mapq : mmq); // NaN stays NaN
	*a_out = a; *largest = top[0];
	return mapq;
}
constexpr double RFA_MAPQ_GUARD = 1e-6; // |value - nearest integer| below this goes to the host; device/libm differences are < 1e-8 for values <= 60
ARX_HDI bool rfa_mapq_needs_host(double v, double largest, double guard)
{
	if (v != v || largest < -290.0) return true;       // NaN, or 10^x near the underflow range
	if (v > 60.0 + guard) return false;                // capped
	const double f = v - floor(v);
	return f < guard || f > 1.0 - guard;
}
ARX_HDI int rfa_mapq_final(double v, const Cand &a, const int64_t *cen_start, const int64_t *cen_end)
{
	double mapq = (v != v) ? v : (v < 60.0 ? v : 60.0);
	if (cen_start && a.rid >= 0 && a.pos > cen_start[a.rid] && a.pos <= cen_end[a.rid]) mapq = 0.0;
	return (mapq != mapq) ? (int)0x80000000 : (int)mapq;
}

// ---- one barcode per workgroup
//
// The block handle B (hip_rt.h: HipBlock, 256 lanes; tests/hostsim: a sequential stand-in) provides
//   pfor(n, f)            f(i) for every i in [0, n), lanes striding, then a workgroup barrier
//   single(f)             lane 0 runs f, then a barrier
//   exclusive_scan(in, out, n) -> total      (in != out)
//   sort_kv(keys, vals, P)                   ascending by (key, (uint32)val), P a power of two
//   argmax(n, keyf, &key, &idx)              largest keyf(i) (0 = "no candidate"), ties to the smallest i; result in every lane
// Control flow around these calls is uniform over the workgroup: every value it depends on is read from memory after a barrier.
// Sums that many lanes contribute to are integers (scores are half-units), so their order cannot be observed.
struct RfaScratch {
	int32_t *hdr;                 // [0] filtered candidates, [2..3] DNA length accumulator (int64)
	uint64_t *skey;               // P sort keys
	int32_t *ord;                 // P: filtered candidates in (contig first seen, position, index) order
	int32_t *flag, *excl;         // molecule starts / scans (n_c + 2 each)
	int32_t *spot;                // n_c: molecule a candidate is the best alignment of its read in, or -1
	int32_t *has_active;          // n_c + 2 (pre-scrap molecules)
	int32_t *mol_goff, *mol_nact, *cursor, *grp_read; // surviving molecules: reads with a spot in them
	int32_t *ach, *num, *scv;     // per sink molecule accumulators of one fastScore sweep
	int32_t *mv, *act;            // per read
	int32_t *first_seen;          // n_seqs + 2
	int64_t *m_lo, *m_hi, *m_len; // per molecule
	int32_t *m_soft;
};
ARX_HDI int rfa_pow2ceil(int n) { int p = 1; while (p < n) p <<= 1; return p; }
ARX_HDI int64_t rfa_scratch_words(int n_c, int n_reads, int n_seqs)
{
	const int64_t P = rfa_pow2ceil(n_c), A = n_c + 2;
	int64_t w = 16 + 2 * P + P + 12 * A + 2 * (int64_t)(n_reads + 2) + (n_seqs + 2) + 6 * A + A;
	return (w + 3) & ~(int64_t)3;
}
ARX_DEVI RfaScratch rfa_carve(int32_t *scratch, int n_c, int n_reads, int n_seqs)
{
	const int P = rfa_pow2ceil(n_c), A = n_c + 2;
	RfaScratch s;
	int32_t *p = scratch;
	s.hdr = p; p += 16;
	s.skey = (uint64_t *)p; p += 2 * P;
	s.m_lo = (int64_t *)p; p += 2 * A; s.m_hi = (int64_t *)p; p += 2 * A; s.m_len = (int64_t *)p; p += 2 * A;
	s.ord = p; p += P;
	s.flag = p; p += A; s.excl = p; p += A; s.spot = p; p += A; s.has_active = p; p += A;
	s.mol_goff = p; p += A; s.mol_nact = p; p += A; s.cursor = p; p += A; s.grp_read = p; p += A;
	s.ach = p; p += A; s.num = p; p += A; s.scv = p; p += A; s.m_soft = p; p += A;
	s.mv = p; p += n_reads + 2; s.act = p; p += n_reads + 2;
	s.first_seen = p; p += n_seqs + 2;
	return s;
}

struct RfaView {
	const Cand *c; const int32_t *roff; int roff0, pen2; const RfaScratch *s;
};
ARX_DEVI int rfa_spot_of(const RfaView &v, int read, int T) // molecule.best_alignment_for_read.Get(read): the read's own candidates are few
{
	for (int i = v.roff[read] - v.roff0; i < v.roff[read + 1] - v.roff0; ++i) if (v.s->spot[i] == T) return i;
	return -1;
}
ARX_DEVI bool rfa_mol_active(int nact, int npot) // isActiveMolecule
{
	if ((double)nact <= 4) return false;
	if ((double)nact / (double)npot < 0.1) return false;
	return true;
}
// fastScore's terms that do not depend on single reads (aligner.go:1200-1236), half-units
ARX_DEVI int rfa_molecule_terms(int nact_s, int npot_s, int nact_t, int npot_t, int num)
{
	int change = 0;
	if (!rfa_mol_active(nact_s - num, npot_s) && rfa_mol_active(nact_s, npot_s)) change += npot_s;
	if (rfa_mol_active(nact_t + num, npot_t) && !rfa_mol_active(nact_t, npot_t)) change -= npot_t;
	if (nact_s - num == 0 && num > 0) change += 6;
	if (nact_t == 0 && num > 0) change -= 6;
	return change;
}
// One active read of source molecule S against its spot `ta` in sink T (the loop body of fastScore, aligner.go:1128-1198)
ARX_DEVI int rfa_read_term(const RfaView &v, int S, int T, int read, int sa, int ta, bool *moves)
{
	const Cand *c = v.c;
	const int mate = read ^ 1, sm = v.s->act[mate];
	const bool source_has_mate = c[sm].mol == S;
	const bool source_pair = source_has_mate && cand_is_pair(c[sa], c[sm]);
	const int tm = rfa_spot_of(v, mate, T);
	const bool sink_pair = tm >= 0 && cand_is_pair(c[ta], c[tm]) && source_has_mate;
	*moves = !source_pair || (source_has_mate && sink_pair);
	int d = c[ta].lap2 - c[sa].lap2;
	if (source_pair && !sink_pair) d += v.pen2 / 2;
	else if (!source_pair && sink_pair) d -= v.pen2 / 2;
	return d;
}

template <class B>
#ifndef ARX_RFA_T
#define ARX_RFA_T(k) do {} while (0)   // diagnostics hook: per-phase clock (tools builds)
#endif
ARX_DEV void rfa_barcode(B &blk, Cand *c, const int32_t *roff, int n_reads, int n_c, int read0, int do_rfa, int pen_int, int n_seqs,
                         const double *p10h, int32_t *scratch, RfaBarcodeOut *out)
{
	const RfaScratch s = rfa_carve(scratch, n_c, n_reads, n_seqs);
	const int roff0 = roff[0], pen2 = 2 * pen_int; // roff holds batch-global candidate offsets, c is the barcode's slice
	RfaView v; v.c = c; v.roff = roff; v.roff0 = roff0; v.pen2 = pen2; v.s = &s;
	int32_t *act = s.act;
	ARX_RFA_T(0);
	// R1 tagBestAlignments: per pair the best (candidate, mate candidate) over the filtered lists; exact ties: first pair wins.
	// A read in a repeat has dozens of candidates and the pair loop is quadratic, so the inner loop runs per CANDIDATE (all
	// lanes busy whatever the read), the pick per pair afterwards.
	int32_t *v1 = s.ach, *a1 = s.num; // best mate score / mate candidate of every candidate of a first read (free until the sweeps)
	blk.pfor(n_c, [&](int i) {
		const int r = c[i].read - read0;
		if ((r & 1) || !c[i].in_filtered) return;
		int bs = 0, bm = -1;
		for (int j = roff[r + 1] - roff0; j < roff[r + 2] - roff0; ++j) {
			if (!c[j].in_filtered) continue;
			const int sc = cand_pair_score2(c[i], c[j], pen2);
			if (bm < 0 || sc > bs) { bs = sc; bm = j; }
		}
		v1[i] = bs; a1[i] = bm;
	});
	blk.pfor(n_reads / 2, [&](int pr) {
		const int r = 2 * pr;
		int bs = 0, ba = -1;
		for (int i = roff[r] - roff0; i < roff[r + 1] - roff0; ++i) {
			if (!c[i].in_filtered) continue;
			if (ba < 0 || v1[i] > bs) { bs = v1[i]; ba = i; }
		}
		const int bm = a1[ba];
		c[ba].active = 1; c[bm].active = 1;
		if (cand_is_pair(c[ba], c[bm])) { c[ba].is_proper = 1; c[bm].is_proper = 1; }
		act[r] = ba; act[r + 1] = bm;
	});
	if (!do_rfa) { blk.single([&]() { out->dna_len = 0; out->n_mol = 0; }); return; }

	ARX_RFA_T(1);
	// R2 inferMolecules: filtered candidates, contigs in first-seen order, sorted by position, split at gaps > 50 kb
	const int P = rfa_pow2ceil(n_c);
	blk.pfor(n_seqs + 2 > 16 ? n_seqs + 2 : 16, [&](int q) { if (q < n_seqs + 2) s.first_seen[q] = 0x7fffffff; if (q < 16) s.hdr[q] = 0; });
	blk.pfor(n_c, [&](int i) { s.spot[i] = -1; if (c[i].in_filtered) { ARX_ATOMIC_MIN(&s.first_seen[c[i].rid + 1], i); ARX_ATOMIC_INC(&s.hdr[0]); } });
	blk.pfor(P, [&](int i) {
		const bool f = i < n_c && c[i].in_filtered;
		s.ord[i] = i < n_c ? i : -1;
		s.skey[i] = f ? ((uint64_t)ARX_LOAD_SHARED(&s.first_seen[c[i].rid + 1]) << 36) | (uint64_t)(c[i].pos + 1) : ~(uint64_t)0;
	});
	ARX_RFA_T(2);
	blk.sort_kv(s.skey, s.ord, P);
	ARX_RFA_T(3);
	const int m_c = ARX_LOAD_SHARED(&s.hdr[0]);
	blk.pfor(m_c, [&](int t) {
		const int i = s.ord[t];
		s.flag[t] = t == 0 || c[s.ord[t - 1]].rid != c[i].rid || c[i].pos - c[s.ord[t - 1]].pos > 50000;
	});
	const int n_mol0 = blk.exclusive_scan(s.flag, s.excl, m_c);
	blk.pfor(m_c > n_mol0 ? m_c : n_mol0, [&](int t) {
		if (t < m_c) c[s.ord[t]].mol = s.excl[t] + s.flag[t] - 1;
		if (t < n_mol0) s.has_active[t] = 0;
	});
	// markBestAlignmentForReadInMolecule: per (molecule, read) group the candidate that scores best against the mate's
	// group in the same molecule (alone: by its own probability); ties go to the smaller position rank = index in ord,
	// which for two candidates of one group is the order of (position, candidate index)
	int32_t *gval = s.scv; // per candidate: its best score against the mate's group in its molecule, or its own (free until R6)
	blk.pfor(n_c, [&](int a) { // quadratic part per candidate, as in R1
		if (!c[a].in_filtered) return;
		const int r = c[a].read - read0, m = c[a].mol, mlo = roff[r ^ 1] - roff0, mhi = roff[(r ^ 1) + 1] - roff0;
		if (c[a].active) s.has_active[m] = 1;
		int val = c[a].lap2; bool any = false;
		for (int b = mlo; b < mhi; ++b) {
			if (!c[b].in_filtered || c[b].mol != m) continue;
			const int sc = cand_pair_score2(c[a], c[b], pen2);
			if (!any || sc > val) { val = sc; any = true; }
		}
		gval[a] = val;
	});
	blk.pfor(n_c, [&](int i) { // per candidate, not per read: a read in a repeat has a hundred candidates in as many molecules
		if (!c[i].in_filtered) return;
		const int r = c[i].read - read0, lo = roff[r] - roff0, hi = roff[r + 1] - roff0, m = c[i].mol;
		for (int k = lo; k < i; ++k) if (c[k].in_filtered && c[k].mol == m) return; // not the first of its (read, molecule) group
		int best = -1, bs = 0;
		for (int a = i; a < hi; ++a) {
			if (!c[a].in_filtered || c[a].mol != m) continue;
			const int val = gval[a];
			const bool before = best >= 0 && (c[a].pos < c[best].pos); // a > best in index, so it ranks first only on a smaller position
			if (best < 0 || val > bs || (val == bs && before)) { bs = val; best = a; }
		}
		c[best].best_in_mol = 1;
	});
	// scrapMolecules: molecules holding an active alignment are renumbered in order; the others disappear
	const int cnt = blk.exclusive_scan(s.has_active, s.excl, n_mol0);
	blk.pfor(m_c > cnt + 1 ? m_c : cnt + 1, [&](int t) {
		if (t < m_c) {
			Cand &x = c[s.ord[t]];
			x.mol = s.has_active[x.mol] ? s.excl[x.mol] : -1;
			if (x.mol < 0) x.best_in_mol = 0;
			s.spot[s.ord[t]] = x.best_in_mol ? x.mol : -1;
		}
		if (t <= cnt) { s.cursor[t] = 0; s.mol_nact[t] = 0; }
	});
	blk.pfor(n_c > n_reads ? n_c : n_reads, [&](int i) {
		if (i < n_c && s.spot[i] >= 0) ARX_ATOMIC_INC(&s.cursor[s.spot[i]]);
		if (i < n_reads && c[act[i]].mol >= 0) ARX_ATOMIC_INC(&s.mol_nact[c[act[i]].mol]);
	});
	blk.pfor(cnt + 1, [&](int m) { s.flag[m] = m < cnt ? ARX_LOAD_SHARED(&s.cursor[m]) : 0; });
	blk.exclusive_scan(s.flag, s.mol_goff, cnt + 1); // mol_goff[cnt] = number of (molecule, read) spots
	blk.pfor(cnt, [&](int m) { s.cursor[m] = s.mol_goff[m]; });
	blk.pfor(n_c, [&](int i) { if (s.spot[i] >= 0) s.grp_read[ARX_ATOMIC_ADD(&s.cursor[s.spot[i]], 1)] = c[i].read - read0; });

	ARX_RFA_T(4);
	// one sweep of fastScore(S, *) over the active reads of S: ach/num per sink molecule
	auto sweep = [&](int S) {
		blk.pfor(cnt, [&](int T) { s.ach[T] = 0; s.num[T] = 0; });
		blk.pfor(s.mol_goff[S + 1] - s.mol_goff[S], [&](int g) {
			const int read = s.grp_read[s.mol_goff[S] + g], sa = act[read];
			if (c[sa].mol != S) return;
			for (int ta = roff[read] - roff0; ta < roff[read + 1] - roff0; ++ta) {
				const int T = s.spot[ta];
				if (T < 0 || T == S) continue;
				bool moves;
				const int d = rfa_read_term(v, S, T, read, sa, ta, &moves);
				ARX_ATOMIC_ADD(&s.ach[T], d); ARX_ATOMIC_INC(&s.num[T]);
			}
		});
	};
	auto score_of = [&](int S, int T, int num) {
		return rfa_molecule_terms(ARX_LOAD_SHARED(&s.mol_nact[S]), s.mol_goff[S + 1] - s.mol_goff[S], ARX_LOAD_SHARED(&s.mol_nact[T]),
		                          s.mol_goff[T + 1] - s.mol_goff[T], num) + ARX_LOAD_SHARED(&s.ach[T]);
	};
	// R5 Optimize(obj, 1, 2, 4*M): two sweeps of 4*M greedy moves, sources round robin.  The walk is deterministic, so once
	// M consecutive sources in a row changed nothing the remaining iterations cannot change anything either.
	int idle = 0;
	for (int it = 0; it < 8 * cnt && idle < cnt; ++it) {
		const int S = it % cnt;
		const int nact_s = ARX_LOAD_SHARED(&s.mol_nact[S]);
		if (nact_s == 0) { ++idle; continue; }
		sweep(S);
		uint64_t bkey; int best_T;
		blk.argmax(cnt, [&](int T) -> uint64_t {
			const int num = ARX_LOAD_SHARED(&s.num[T]);
			if (T == S || num == 0) return 0;
			return ((uint64_t)(uint32_t)(score_of(S, T, num) + 0x40000000) << 32) | (uint32_t)ARX_LOAD_SHARED(&s.mol_nact[T]);
		}, &bkey, &best_T);
		const int best_sc = (int)(uint32_t)(bkey >> 32) - 0x40000000;
		if (bkey == 0 || !(best_sc > 0 || (best_sc == 0 && (int)(uint32_t)bkey > nact_s))) { ++idle; continue; }
		idle = 0;
		const int g0 = s.mol_goff[S], ng = s.mol_goff[S + 1] - g0, T = best_T;
		blk.pfor(ng, [&](int g) { // acceptMove works from the state before the move: decide every read first ...
			const int read = s.grp_read[g0 + g], sa = act[read];
			s.mv[read] = -1;
			if (c[sa].mol != S) return;
			const int ta = rfa_spot_of(v, read, T);
			if (ta < 0) return;
			bool moves;
			rfa_read_term(v, S, T, read, sa, ta, &moves);
			if (moves) s.mv[read] = ta;
		});
		blk.pfor(ng, [&](int g) { // ... then apply
			const int read = s.grp_read[g0 + g], ta = s.mv[read];
			if (ta < 0) return;
			c[act[read]].active = 0; c[ta].active = 1; act[read] = ta;
			ARX_ATOMIC_ADD(&s.mol_nact[S], -1); ARX_ATOMIC_INC(&s.mol_nact[T]);
		});
	}
	ARX_RFA_T(5);
	// R6 method 2: sum_move += 10^fastScore(S, T), T ascending, for every active alignment of S that has a spot in T
	for (int S = 0; S < cnt; ++S) {
		if (ARX_LOAD_SHARED(&s.mol_nact[S]) == 0) continue;
		sweep(S);
		blk.pfor(cnt, [&](int T) { const int num = ARX_LOAD_SHARED(&s.num[T]); s.scv[T] = (T != S && num > 0) ? score_of(S, T, num) : 0; });
		blk.pfor(s.mol_goff[S + 1] - s.mol_goff[S], [&](int g) {
			const int read = s.grp_read[s.mol_goff[S] + g], sa = act[read];
			if (c[sa].mol != S) return;
			double sum = c[sa].sum_move;
			for (int last = -1;;) { // the read's other spots in ascending molecule order
				int T = 0x7fffffff;
				for (int i = roff[read] - roff0; i < roff[read + 1] - roff0; ++i) { const int m = s.spot[i]; if (m > last && m != S && m < T) T = m; }
				if (T == 0x7fffffff) break;
				const int sc = s.scv[T];
				sum += sc < -RFA_P10_HALF ? p10h[0] * 0.0 : (sc > RFA_P10_HALF ? p10h[2 * RFA_P10_HALF] * 1e300 * 1e300 : p10h[sc + RFA_P10_HALF]);
				last = T;
			}
			c[sa].sum_move = sum;
		});
	}
	ARX_RFA_T(6);
	// setMoleculeConfidences + updateAlignmentsMoleculeStatus + the DNA length of calculateLogMoleculePenalty (all terms are integers)
	blk.pfor(cnt, [&](int m) { s.m_soft[m] = 0; s.m_lo[m] = 0x7fffffffffffffffLL; s.m_hi[m] = -1; s.m_len[m] = 0; });
	blk.pfor(n_reads, [&](int r) {
		const Cand &a = c[act[r]];
		const int m = a.mol;
		if (m < 0) return;
		if (a.soft_clipped > 0) ARX_ATOMIC_INC(&s.m_soft[m]);
		ARX_ATOMIC_MAX64(&s.m_hi[m], a.pos); ARX_ATOMIC_MIN64(&s.m_lo[m], a.pos);
		ARX_ATOMIC_ADD64(&s.m_len[m], (a.aend - a.pos) * 2);
	});
	int64_t *dna = (int64_t *)(s.hdr + 2);
	blk.pfor(cnt, [&](int m) {
		const int npot = s.mol_goff[m + 1] - s.mol_goff[m], nact = ARX_LOAD_SHARED(&s.mol_nact[m]);
		const int64_t lo = ARX_LOAD_SHARED(&s.m_lo[m]), hi = ARX_LOAD_SHARED(&s.m_hi[m]);
		const bool is_act = nact - ARX_LOAD_SHARED(&s.m_soft[m]) > 4 && (double)nact / (double)npot > 0.1;
		s.flag[m] = is_act;
		if (is_act) { if (hi >= lo) ARX_ATOMIC_ADD64(dna, (hi - lo) + 1000); }
		else ARX_ATOMIC_ADD64(dna, ARX_LOAD_SHARED(&s.m_len[m]));
	});
	blk.pfor(m_c, [&](int t) { Cand &x = c[s.ord[t]]; if (x.mol >= 0) x.active_molecule = s.flag[x.mol]; });
	blk.single([&]() { out->dna_len = cnt > 0 ? 1000.0 + (double)ARX_LOAD_SHARED(dna) : 0.0; out->n_mol = cnt; });
	ARX_RFA_T(7);
}

} // namespace arx
