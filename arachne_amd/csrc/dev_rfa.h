// dev_rfa.h -- the Go half of the per-barcode path on the device: candidate post-processing (GetChains/GetAlignments),
// best-pair tagging, molecule inference, the RFA joint-placement sweep and the molecule-move probability sums.
// One barcode per thread; all state of a barcode lives in its slices of batch-wide pools in HBM.
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
ARX_DEV void cand_build_read(const IndexView &ix, int read, const Reg *regs, const Aln *alns, const uint32_t *cig, int cig_w, int g0, int n_regs, Cand *out)
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
		c.reg = g0 + i; c.read = read; c.rid = al.rid; c.score = rg.score; c.reversed = al.is_rev; c.sum_move = 1.0; c.mol = -1;
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

// ---- per-barcode working view
struct RfaView {
	Cand *c;               // candidates of the barcode (all, filtered or not); read r owns [roff[r], roff[r+1])
	const int32_t *roff;   // barcode-local read -> candidate offset (relative to c)
	int n_reads, n_c, pen2;
	int32_t *act;          // per read: its active candidate
	// molecule tables (after scrap): groups = (molecule, read) pairs sorted by molecule then read
	int32_t *grp_read, *grp_best, *mol_goff; // mol_goff[m]..mol_goff[m+1] = groups of molecule m
	int32_t *mol_nact;
	int n_mol;
};

ARX_DEVI int rfa_best_for(const RfaView &v, int mol, int read) // molecule.best_alignment_for_read.Get(read): binary search in the molecule's groups
{
	int lo = v.mol_goff[mol], hi = v.mol_goff[mol + 1];
	while (lo < hi) { int mid = (lo + hi) >> 1; if (v.grp_read[mid] < read) lo = mid + 1; else hi = mid; }
	return (lo < v.mol_goff[mol + 1] && v.grp_read[lo] == read) ? v.grp_best[lo] : -1;
}
ARX_DEVI bool rfa_mol_active(const RfaView &v, int m, int change) // isActiveMolecule
{
	const double active = (double)(v.mol_nact[m] + change), potential = (double)(v.mol_goff[m + 1] - v.mol_goff[m]);
	if (active <= 4) return false;
	if (active / potential < 0.1) return false;
	return true;
}

// fastScore in half-units.  When mv_read != nullptr the (read, sink candidate) pairs acceptMove would apply are recorded.
ARX_DEV int rfa_fast_score2(const RfaView &v, int S, int T, int *num_out, int32_t *mv_read, int32_t *mv_sink, int *n_mv)
{
	int change = 0, ach = 0, num = 0, nmv = 0;
	for (int g = v.mol_goff[S]; g < v.mol_goff[S + 1]; ++g) {
		const int read = v.grp_read[g], sa = v.act[read];
		if (v.c[sa].mol != S) continue; // only the source's active alignments
		const int ta = rfa_best_for(v, T, read);
		if (ta < 0) continue;
		const Cand &src = v.c[sa], &snk = v.c[ta];
		const int mate = read ^ 1, sm = v.act[mate];
		const bool source_has_mate = v.c[sm].mol == S;
		const bool source_pair = source_has_mate && cand_is_pair(src, v.c[sm]);
		const int tm = rfa_best_for(v, T, mate);
		const bool sink_pair = tm >= 0 && cand_is_pair(snk, v.c[tm]) && source_has_mate;
		if (!source_pair || (source_has_mate && sink_pair)) { if (mv_read) { mv_read[nmv] = read; mv_sink[nmv] = ta; } ++nmv; }
		ach += snk.lap2 - src.lap2;
		if (source_pair && !sink_pair && S != T) ach += v.pen2 / 2;
		else if (!source_pair && sink_pair && S != T) ach -= v.pen2 / 2;
		++num;
	}
	const int npot_s = v.mol_goff[S + 1] - v.mol_goff[S], npot_t = v.mol_goff[T + 1] - v.mol_goff[T];
	if (!rfa_mol_active(v, S, -num) && rfa_mol_active(v, S, 0) && S != T) change += npot_s;
	if (rfa_mol_active(v, T, num) && !rfa_mol_active(v, T, 0) && S != T) change -= npot_t;
	if (v.mol_nact[S] - num == 0 && num > 0 && S != T) change += 6;
	if (v.mol_nact[T] == 0 && num > 0 && S != T) change -= 6;
	*num_out = num;
	if (n_mv) *n_mv = nmv;
	return change + ach;
}

struct SortByContigPos { // positions per contig in first-seen contig order, by pos, ties by candidate order (stable)
	const Cand *c; const int32_t *first_seen; // first_seen[rid + 1] = first candidate index with that contig
	ARX_DEVI bool operator()(int a, int b) const
	{
		const int fa = first_seen[c[a].rid + 1], fb = first_seen[c[b].rid + 1];
		if (fa != fb) return fa < fb;
		if (c[a].pos != c[b].pos) return c[a].pos < c[b].pos;
		return a < b;
	}
};
struct SortByMolRead { // (molecule, read, position rank)
	const Cand *c; const int32_t *rank;
	ARX_DEVI bool operator()(int a, int b) const
	{
		if (c[a].mol != c[b].mol) return c[a].mol < c[b].mol;
		if (c[a].read != c[b].read) return c[a].read < c[b].read;
		return rank[a] < rank[b];
	}
};

// One barcode.  scratch: rfa_scratch_words(n_c, n_reads, n_seqs) int32.  p10h: table of 10^(x/2), index x + RFA_P10_HALF.
ARX_HDI int64_t rfa_scratch_words(int n_c, int n_reads, int n_seqs) { return 7 * (int64_t)n_c + 2 + n_reads + n_seqs + 2; }

ARX_DEV void rfa_barcode(Cand *c, const int32_t *roff, int n_reads, int n_c, int read0, int do_rfa, int pen_int, int n_seqs,
                         const double *p10h, int32_t *scratch, RfaBarcodeOut *out)
{
	RfaView v;
	v.c = c; v.roff = roff; v.n_reads = n_reads; v.n_c = n_c; v.pen2 = 2 * pen_int; v.n_mol = 0;
	int32_t *ord = scratch, *rank = ord + n_c, *grp_read = rank + 2 * n_c + 2, *grp_best = grp_read + n_c, *mvbuf = grp_best + n_c; // rank: 2*n_c+2 (reused for the molecule tables), mvbuf: 2*n_c
	int32_t *act = mvbuf + 2 * n_c, *first_seen = act + n_reads;
	v.act = act;
	out->dna_len = 0; out->n_mol = 0;
	// local read ids: candidates store batch-global reads; inside the barcode use read - read0
	for (int i = 0; i < n_c; ++i) c[i].read -= read0;
	// R1: per pair the best (candidate, mate candidate) over the filtered lists; exact ties: first pair wins
	const int roff0 = roff[0]; // roff holds batch-global candidate offsets, c is the barcode's slice
	for (int r = 0; r + 1 < n_reads; r += 2) {
		int bs = 0, ba = -1, bm = -1;
		for (int i = roff[r] - roff0; i < roff[r + 1] - roff0; ++i) {
			if (!c[i].in_filtered) continue;
			for (int j = roff[r + 1] - roff0; j < roff[r + 2] - roff0; ++j) {
				if (!c[j].in_filtered) continue;
				const int s = cand_pair_score2(c[i], c[j], v.pen2);
				if (ba < 0 || s > bs) { bs = s; ba = i; bm = j; }
			}
		}
		c[ba].active = 1; c[bm].active = 1;
		if (cand_is_pair(c[ba], c[bm])) { c[ba].is_proper = 1; c[bm].is_proper = 1; }
		act[r] = ba; act[r + 1] = bm;
	}
	if (do_rfa) {
		// R2 inferMolecules: filtered candidates, contigs in first-seen order, sorted by position, split at gaps > 50 kb
		for (int s = 0; s < n_seqs + 1; ++s) first_seen[s] = 0x7fffffff;
		int m_c = 0;
		for (int i = 0; i < n_c; ++i) if (c[i].in_filtered) { ord[m_c++] = i; if (first_seen[c[i].rid + 1] > i) first_seen[c[i].rid + 1] = i; }
		SortByContigPos lt1; lt1.c = c; lt1.first_seen = first_seen;
		ks_introsort(m_c, ord, lt1);
		int n_mol0 = 0;
		for (int t = 0; t < m_c; ++t) {
			const int i = ord[t];
			if (t == 0 || c[ord[t - 1]].rid != c[i].rid || c[i].pos - c[ord[t - 1]].pos > 50000) ++n_mol0;
			c[i].mol = n_mol0 - 1;
			rank[i] = t;
		}
		// markBestAlignmentForReadInMolecule: group by (molecule, read); the best candidate of a group against the mate's group
		SortByMolRead lt2; lt2.c = c; lt2.rank = rank;
		ks_introsort(m_c, ord, lt2);
		// has_active per pre-scrap molecule, kept in mvbuf[0..n_mol0)
		int32_t *has_active = mvbuf;
		for (int m = 0; m < n_mol0; ++m) has_active[m] = 0;
		for (int t = 0; t < m_c;) {
			const int m = c[ord[t]].mol, r = c[ord[t]].read;
			int e = t;
			while (e < m_c && c[ord[e]].mol == m && c[ord[e]].read == r) ++e;
			// mate group: same molecule, read r^1 -- adjacent to this group in the sorted order
			int ms = -1, me = -1;
			if ((r & 1) == 0) { if (e < m_c && c[ord[e]].mol == m && c[ord[e]].read == r + 1) { ms = e; me = e; while (me < m_c && c[ord[me]].mol == m && c[ord[me]].read == r + 1) ++me; } }
			else { int b0 = t; while (b0 > 0 && c[ord[b0 - 1]].mol == m && c[ord[b0 - 1]].read == r - 1) --b0; if (b0 < t) { ms = b0; me = t; } }
			int best = -1, bs = 0;
			for (int a = t; a < e; ++a) {
				const int ia = ord[a];
				if (ms >= 0) {
					for (int b = ms; b < me; ++b) { const int s = cand_pair_score2(c[ia], c[ord[b]], v.pen2); if (best < 0 || s > bs) { bs = s; best = ia; } }
				} else if (best < 0 || c[ia].lap2 > bs) { bs = c[ia].lap2; best = ia; }
				if (c[ia].active) has_active[m] = 1;
			}
			c[best].best_in_mol = 1;
			t = e;
		}
		// scrapMolecules: renumber molecules that hold an active alignment; the others disappear
		int cnt = 0;
		for (int m = 0; m < n_mol0; ++m) has_active[m] = has_active[m] ? cnt++ : -1;
		for (int t = 0; t < m_c; ++t) { Cand &x = c[ord[t]]; x.mol = has_active[x.mol]; if (x.mol < 0) x.best_in_mol = 0; }
		// group tables of the surviving molecules (ord is still sorted by old molecule id; renumbering keeps the order)
		int32_t *mol_goff = rank;           // rank[] is free now: reuse for mol_goff (cnt + 1 entries) and mol_nact (cnt entries)
		int32_t *mol_nact = rank + cnt + 1;
		int n_g = 0, cur_m = -1;
		for (int t = 0; t < m_c; ++t) {
			const Cand &x = c[ord[t]];
			if (x.mol < 0 || !x.best_in_mol) continue;
			while (cur_m < x.mol) mol_goff[++cur_m] = n_g;
			grp_read[n_g] = x.read; grp_best[n_g] = ord[t]; ++n_g;
		}
		while (cur_m < cnt) mol_goff[++cur_m] = n_g;
		for (int m = 0; m < cnt; ++m) mol_nact[m] = 0;
		for (int r = 0; r < n_reads; ++r) if (c[act[r]].mol >= 0) ++mol_nact[c[act[r]].mol];
		v.grp_read = grp_read; v.grp_best = grp_best; v.mol_goff = mol_goff; v.mol_nact = mol_nact; v.n_mol = cnt;
		// R5 Optimize(obj, 1, 2, 4*M): two sweeps of 4*M greedy moves, sources round robin
		if (cnt > 0) {
			int32_t *mvr = mvbuf, *mvs = mvbuf + n_c; // candidate moves of the sink being scored (<= n_reads <= n_c entries each)
			int cur = 0;
			for (int it = 0; it < 8 * cnt; ++it) {
				const int S = cur;
				cur = (cur + 1) % cnt;
				if (mol_nact[S] == 0) continue;
				int have = 0, best_sc = 0, best_T = -1;
				for (int T = 0; T < cnt; ++T) {
					if (T == S) continue;
					int num, sc = rfa_fast_score2(v, S, T, &num, 0, 0, 0);
					if (num > 0 && (!have || sc > best_sc || (sc == best_sc && mol_nact[T] > mol_nact[best_T]))) { have = 1; best_sc = sc; best_T = T; }
				}
				if (have && (best_sc > 0 || (best_sc == 0 && mol_nact[best_T] > mol_nact[S]))) {
					int num, nmv;
					rfa_fast_score2(v, S, best_T, &num, mvr, mvs, &nmv); // recompute the winning move's read list, then acceptMove
					for (int q = 0; q < nmv; ++q) {
						const int read = mvr[q];
						c[act[read]].active = 0; --mol_nact[S];
						c[mvs[q]].active = 1; ++mol_nact[best_T];
						act[read] = mvs[q];
					}
				}
			}
		}
		// R6 method 2: sum_move += 10^fastScore(S, T) for every active alignment of S that has a spot in T
		for (int S = 0; S < cnt; ++S)
			for (int T = 0; T < cnt; ++T) {
				if (S == T) continue;
				int num, sc = rfa_fast_score2(v, S, T, &num, 0, 0, 0);
				const double p = sc < -RFA_P10_HALF ? p10h[0] * 0.0 : (sc > RFA_P10_HALF ? p10h[2 * RFA_P10_HALF] * 1e300 * 1e300 : p10h[sc + RFA_P10_HALF]);
				for (int g = mol_goff[S]; g < mol_goff[S + 1]; ++g) {
					const int read = grp_read[g], sa = act[read];
					if (c[sa].mol == S && rfa_best_for(v, T, read) >= 0) c[sa].sum_move += p;
				}
			}
		// setMoleculeConfidences + updateAlignmentsMoleculeStatus + the DNA length of calculateLogMoleculePenalty
		double dna = 1000.0;
		for (int m = 0; m < cnt; ++m) {
			const int npot = mol_goff[m + 1] - mol_goff[m];
			int soft = 0;
			int64_t lo = 0x7fffffffffffffffLL, hi = -1;
			double inactive_len = 0.0;
			for (int g = mol_goff[m]; g < mol_goff[m + 1]; ++g) {
				const Cand &a = c[act[grp_read[g]]];
				if (a.mol != m) continue;
				if (a.soft_clipped > 0) ++soft;
				if (a.pos > hi) hi = a.pos;
				if (a.pos < lo) lo = a.pos;
				inactive_len += (double)(a.aend - a.pos) * 2.0;
			}
			const double conf = (double)mol_nact[m] / (double)npot;
			const bool is_act = mol_nact[m] - soft > 4 && conf > 0.1;
			if (is_act) { if (hi >= lo) dna += (double)(hi - lo) + 1000.0; }
			else dna += inactive_len;
			mol_nact[m] = is_act ? -1 - mol_nact[m] : mol_nact[m]; // flag in the sign; no score is computed after this point
		}
		for (int t = 0; t < m_c; ++t) { Cand &x = c[ord[t]]; if (x.mol >= 0) x.active_molecule = mol_nact[x.mol] < 0; }
		out->dna_len = cnt > 0 ? dna : 0.0;
		out->n_mol = cnt;
	}
	for (int i = 0; i < n_c; ++i) c[i].read += read0;
}

} // namespace arx
