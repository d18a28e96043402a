// dev_post.h -- the per-barcode passes the reference runs on the candidates after placement, moved onto the device outputs
// so that the host never re-fetches the reference sequence:
//   * the CIGAR walk of GetAlignments (aligner.go:1505-1570): matches, mismatch locations in reference and read coordinates,
//     against GetSeq's copy of the reference (gobwa.go:50-80); readmap_s/_e = the region's qb/qe (gobwa.go:368-369, aligner.go:1620-1623)
//   * markDuplicates (aligner.go:598-641)
//   * CheckSplitReads / GetSplitAlignment (split.go:31-163)
// Compiled for the device and for the host test double.
#pragma once
#include "arx_dev.h"
#include "dev_rfa.h"

namespace arx {

struct CandPost { // per candidate
	int32_t qb, qe;       // Alignment.readmap_s / readmap_e (0 for a placeholder)
	int32_t matches;      // Alignment.matches
	int32_t n_mm, mm_off; // its mismatch locations: mm_ref/mm_read[mm_off .. mm_off + n_mm)
	int32_t duplicate;    // Alignment.duplicate (only ever set on active candidates)
};
struct SplitRec { // per read: Alignment.secondary of its active candidate
	int32_t split;         // candidate index (batch-global), -1 for none
	int32_t mapq;          // split.mapq
	int32_t is_proper;     // split.is_proper as GetSplitAlignment leaves it: isPair(split, active mate)
	int32_t n_split_cand;  // candidates that passed the overlap/score test
	int32_t order_pinned;  // 0: more than 12 candidates and a score tie among those that decide the result (see split_read)
	int32_t second_best2;  // split.mapq_data.second_best_score * 2
	int32_t score2;        // split.mapq_data.score * 2
	int32_t pad;
};

// GetSeq (gobwa.go:50-80) as a function of the position in the string it returns: bns_fetch_seq clamps [beg, end) to the
// contig around the midpoint (bntseq.c:421-447) and GetSeq lays the clamped bases at the front of a string of the unclamped
// length (reverse-complemented for reversed = true), leaving zero bytes behind them.  Returns the base code, or -1 for a zero byte.
struct RefString {
	const uint8_t *pac; int64_t cbeg; int32_t clen, reversed;
	ARX_DEVI int at(int k) const
	{
		if (k >= clen) return -1;
		return reversed ? 3 - pac_base(pac, cbeg + clen - 1 - k) : pac_base(pac, cbeg + k);
	}
};
ARX_DEVI RefString ref_string(const IndexView &ix, int rid, int64_t start, int64_t end, int reversed)
{
	const int64_t off = ix.ann_off[rid], far_end = off + ix.ann_len[rid];
	int64_t b = start + off, e = end + off;
	if (b < off) b = off;
	if (e > far_end) e = far_end;
	RefString s;
	s.pac = ix.pac; s.cbeg = b; s.clen = e > b ? (int)(e - b) : 0; s.reversed = reversed;
	return s;
}

// The walk of aligner.go:1529-1570 for one candidate.  EMIT = false counts, EMIT = true also writes the two lists.
template <bool EMIT>
ARX_DEVI int cand_walk(const IndexView &ix, const Cand &c, const Aln &al, const uint32_t *cg, const uint8_t *read, int l_read, int *matches_out,
                       int32_t *mm_ref, int32_t *mm_read)
{
	const int64_t ref_start = c.pos, ref_end = c.aend; // refStart/refEnd: chain.pos/aend, swapped +1 for reversed -- the same swap Cand carries
	const int L = (int)(ref_end - ref_start);
	const RefString rs = ref_string(ix, c.rid, ref_start, ref_end, c.reversed);
	int ref_off = 0, read_off = 0, matches = 0, indel_len = 0, n = 0;
	const int nc = al.n_cigar;
	for (int x = 0; x < nc; ++x) {
		const uint32_t w = cg[c.reversed ? nc - 1 - x : x];
		const int op = w & 0xf, len = (int)(w >> 4);
		if (op == 0) {
			matches += len;
			for (int m = 0; m < len; ++m) {
				if (ref_off + m >= L || read_off + m >= l_read) continue;
				const int rb = rs.at(ref_off + m), qb = read[read_off + m];
				if (rb != qb) { // an N of the read (4) and a zero byte of the string (-1) differ from everything
					if (EMIT) {
						mm_ref[n] = c.reversed ? (int32_t)(ref_end - (ref_off + m)) : (int32_t)(ref_off + ref_start + m);
						mm_read[n] = read_off + m;
					}
					++n;
				}
			}
			ref_off += len; read_off += len;
		} else if (op == 1) { indel_len += len; read_off += len; }
		else if (op == 2) { indel_len += len; ref_off += len; }
		else if (op == 3) read_off += len;
	}
	*matches_out = matches - (al.NM - indel_len); // matches -= mismatches, before the clamp at 0 (aligner.go:1572-1576)
	return n;
}

// ---- markDuplicates: active candidates of one barcode with the same (read1, reversed, contig, pos, mate contig, mate pos); the first
// read in read order keeps its flag clear.  One open-addressing table for the whole batch, the barcode index is part of the key; a
// slot holds the smallest read id seen for its key.
struct DupKey { int64_t pos, mpos; int32_t bc, rid, mrid, bits; };
ARX_DEVI bool dup_key_eq(const DupKey &a, const DupKey &b) { return a.pos == b.pos && a.mpos == b.mpos && a.bc == b.bc && a.rid == b.rid && a.mrid == b.mrid && a.bits == b.bits; }
ARX_DEVI DupKey dup_key(const Cand *cands, const int32_t *act, int r, int bc)
{
	const Cand &a = cands[act[r]], &m = cands[act[r ^ 1]];
	DupKey k;
	k.pos = a.pos; k.mpos = m.pos; k.bc = bc; k.rid = a.rid; k.mrid = m.rid; k.bits = ((r & 1) ^ 1) | a.reversed << 1;
	return k;
}
ARX_DEVI uint32_t dup_hash(const DupKey &k)
{
	uint64_t h = (uint64_t)k.pos * 0x9E3779B97F4A7C15ull;
	h ^= (uint64_t)k.mpos + 0x7F4A7C15ull + (h << 6) + (h >> 2);
	h ^= ((uint64_t)(uint32_t)k.bc << 32 | (uint32_t)(k.rid * 4 + k.bits)) * 0xC2B2AE3D27D4EB4Full;
	h ^= (uint64_t)(uint32_t)k.mrid * 0x165667B19E3779F9ull;
	h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
	return (uint32_t)h;
}
ARX_DEVI void dup_insert(int32_t *table, uint32_t mask, const Cand *cands, const int32_t *act, const int32_t *bc_of, int r)
{
	const DupKey k = dup_key(cands, act, r, bc_of[r]);
	uint32_t s = dup_hash(k) & mask;
	for (;;) {
		int cur = ARX_ATOMIC_CAS(&table[s], -1, r);
		if (cur == -1) return;
		if (dup_key_eq(k, dup_key(cands, act, cur, bc_of[cur]))) { ARX_ATOMIC_MIN(&table[s], r); return; }
		s = (s + 1) & mask;
	}
}
ARX_DEVI int dup_first(const int32_t *table, uint32_t mask, const Cand *cands, const int32_t *act, const int32_t *bc_of, int r)
{
	const DupKey k = dup_key(cands, act, r, bc_of[r]);
	uint32_t s = dup_hash(k) & mask;
	for (;;) {
		const int cur = table[s];
		if (cur < 0) return r; // cannot happen after dup_insert(r)
		if (dup_key_eq(k, dup_key(cands, act, cur, bc_of[cur]))) return cur;
		s = (s + 1) & mask;
	}
}

// ---- GetSplitAlignment (split.go:31-139) for one read.  sort.Sort there is Go's pdqsort: an insertion sort (stable) up to 12
// elements, implementation-defined among equal scores above that.  The stable order is used here; order_pinned = 0 reports the
// reads for which more than 12 candidates tie in a place that decides the result.
ARX_DEVI bool split_candidate(const Cand &S, const CandPost &sp, int Ps, int Pe, const Cand &M) // the loop body of split.go:55-98
{
	if (S.active || S.pos == -1) return false;
	int Ss = sp.qb, Se = sp.qe, overlap;
	if (Ss > Se) { const int t = Ss; Ss = Se; Se = t; }
	if ((Ps < Ss && Pe > Se) || (Ss < Ps && Se > Pe)) return false; // one contains the other
	overlap = Ps < Ss ? Pe - Ss : Se - Ps;
	if (!(overlap < (Se - Ss) / 2)) return false;
	return S.score >= 36 || cand_is_pair(S, M);
}
ARX_DEVI SplitRec split_read(const Cand *cands, const CandPost *post, int lo, int hi, int a, int am, int l_read, int penalty,
                             const int64_t *cen_start, const int64_t *cen_end)
{
	SplitRec o;
	o.split = -1; o.mapq = 0; o.is_proper = 0; o.n_split_cand = 0; o.order_pinned = 1; o.second_best2 = 0; o.score2 = 0; o.pad = 0;
	const Cand &P = cands[a], &M = cands[am];
	if (P.pos == -1) return o;
	int Ps = post[a].qb, Pe = post[a].qe;
	if (Ps > Pe) { const int t = Ps; Ps = Pe; Pe = t; }
	if (Pe - Ps > l_read - 15) return o; // "need at least 28 clipped bases" (the test is 15)
	int c0 = -1, c1 = -1, n = 0;
	for (int i = lo; i < hi; ++i) {
		if (!split_candidate(cands[i], post[i], Ps, Pe, M)) continue;
		++n;
		// first two of a stable sort by descending score: an equal score never overtakes an earlier candidate
		if (c0 < 0) c0 = i;
		else if (cands[i].score > cands[c0].score) { c1 = c0; c0 = i; }
		else if (c1 < 0 || cands[i].score > cands[c1].score) c1 = i;
	}
	o.n_split_cand = n;
	if (n == 0) return o;
	const int pen2 = 2 * penalty;
	const Cand &C = cands[c0];
	double mapq;
	if (n > 1) { mapq = (double)(C.score - cands[c1].score); o.second_best2 = cand_pair_score2(P, cands[c1], pen2); }
	else { mapq = (double)C.score; o.second_best2 = P.lap2 + pen2 - 20 - (l_read - 25); } // scoreAlignment(primary, nil, 0) + psuedoCountAlignmentScore(c, 0)
	if (cen_start && C.rid >= 0 && C.pos > cen_start[C.rid] && C.pos <= cen_end[C.rid]) mapq = 0.0;
	if (mapq > 60) mapq = 60;
	o.split = c0; o.mapq = (int)mapq; o.is_proper = cand_is_pair(C, M);
	o.score2 = cand_pair_score2(C, M, pen2); // scoreAlignment(split, active.mate_alignment, 0)
	if (n > 12) { // beyond the insertion-sort range: pinned only if neither of the two places is tied
		int e0 = 0, e1 = 0;
		for (int i = lo; i < hi; ++i) {
			if (!split_candidate(cands[i], post[i], Ps, Pe, M)) continue;
			e0 += cands[i].score == C.score; e1 += cands[i].score == cands[c1].score;
		}
		o.order_pinned = e0 == 1 && e1 == 1;
	}
	return o;
}

} // namespace arx
