// pipeline_post.h -- the passes between placement and the BAM records, on the candidate records the RFA
// stage left in HBM: GetAlignments' CIGAR walk (aligner.go:1505-1570), markDuplicates (aligner.go:611) and CheckSplitReads
// (split.go:144-163).  Everything here is thread-per-candidate / thread-per-read streaming work over resident arrays.
#pragma once
#include "pipeline_rfa.h"
#include "dev_post.h"

namespace arx {

struct KPostWalk { // one candidate per lane; FILL = false: qb/qe, matches, number of mismatch locations; FILL = true: the lists
	IndexView ix; const Cand *cands; const Reg *regs; const Aln *alns; const uint32_t *cig; const uint8_t *bases; const int32_t *base_off, *lens;
	CandPost *post; int32_t *n_mm; const int32_t *mm_off; int32_t *mm_ref, *mm_read; int fill;
	ARX_DEV void operator()(int i, int) const
	{
		const Cand &c = cands[i];
		if (c.reg < 0) { // placeholder: empty CIGAR, readmap_s/_e stay 0 (aligner.go:1620)
			if (!fill) { CandPost p = CandPost(); post[i] = p; n_mm[i] = 0; }
			else post[i].mm_off = mm_off[i];
			return;
		}
		const Aln &al = alns[c.reg];
		const uint8_t *read = bases + base_off[c.read];
		int matches;
		if (!fill) {
			CandPost p = CandPost();
			p.qb = regs[c.reg].qb; p.qe = regs[c.reg].qe;
			p.n_mm = cand_walk<false>(ix, c, al, cig + al.cigar_off, read, lens[c.read], &matches, nullptr, nullptr);
			p.matches = matches;
			post[i] = p; n_mm[i] = p.n_mm;
		} else {
			post[i].mm_off = mm_off[i];
			if (post[i].n_mm) cand_walk<true>(ix, c, al, cig + al.cigar_off, read, lens[c.read], &matches, mm_ref + mm_off[i], mm_read + mm_off[i]);
		}
	}
};
struct KPostActive { // per read: its active candidate and its barcode
	const Cand *cands; const int32_t *cand_off, *bc_read_off; int n_barcodes; int32_t *act, *bc_of;
	ARX_DEV void operator()(int r, int) const
	{
		int a = cand_off[r];
		for (int i = cand_off[r]; i < cand_off[r + 1]; ++i) if (cands[i].active) { a = i; break; }
		act[r] = a;
		int lo = 0, hi = n_barcodes;
		while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (bc_read_off[mid] <= r) lo = mid; else hi = mid; }
		bc_of[r] = lo;
	}
};
struct KDupInsert {
	const Cand *cands; const int32_t *act, *bc_of; int32_t *table; uint32_t mask;
	ARX_DEV void operator()(int r, int) const { dup_insert(table, mask, cands, act, bc_of, r); }
};
struct KDupMark {
	const Cand *cands; const int32_t *act, *bc_of; const int32_t *table; uint32_t mask; CandPost *post;
	ARX_DEV void operator()(int r, int) const { post[act[r]].duplicate = dup_first(table, mask, cands, act, bc_of, r) != r; }
};
struct KSplit {
	const Cand *cands; const CandPost *post; const int32_t *cand_off, *act, *lens; int penalty; const int64_t *cen_start, *cen_end; SplitRec *out;
	ARX_DEV void operator()(int r, int) const
	{
		out[r] = split_read(cands, post, cand_off[r], cand_off[r + 1], act[r], act[r ^ 1], lens[r], penalty, cen_start, cen_end);
	}
};

struct PostResult { CandPost *d_post = nullptr; SplitRec *d_split = nullptr; int32_t *d_mm_ref = nullptr, *d_mm_read = nullptr; int64_t n_mm = 0; bool done = false; };

template <class RT> struct PostStage {
	static int run(Pipeline<RT> &pipe, const typename Pipeline<RT>::DeviceBatch &b, typename Pipeline<RT>::Work &w, const RfaResult &rfa, PostResult &res)
	{
		RT &rt = pipe.rt;
		const int R = b.n_reads;
		const int64_t NC = rfa.n_cands;
		CandPost *post = rt.template alloc<CandPost>((size_t)NC + 1);
		int32_t *n_mm = rt.template alloc<int32_t>((size_t)NC + 1), *mm_off = rt.template alloc<int32_t>((size_t)NC + 2);
		KPostWalk kw{pipe.ix, rfa.d_cands, w.c_regs, w.c_alns, w.c_cig, b.bases, b.base_off, b.lens, post, n_mm, mm_off, nullptr, nullptr, 0};
		rt.launch_wide("post_walk", (int)NC, kw);
		const int64_t NM = rt.exclusive_scan(n_mm, mm_off, (int)NC);
		int32_t *mm_ref = rt.template alloc<int32_t>((size_t)NM + 1), *mm_read = rt.template alloc<int32_t>((size_t)NM + 1);
		kw.mm_ref = mm_ref; kw.mm_read = mm_read; kw.fill = 1;
		rt.launch_wide("post_fill", (int)NC, kw);
		int32_t *act = rt.template alloc<int32_t>(R + 1), *bc_of = rt.template alloc<int32_t>(R + 1);
		KPostActive ka{rfa.d_cands, rfa.d_cand_off, rfa.d_bc_read_off, rfa.n_barcodes, act, bc_of};
		rt.launch_wide("post_active", R, ka);
		uint32_t cap = 64;
		while (cap < 2u * (uint32_t)R) cap <<= 1;
		int32_t *table = rt.template alloc<int32_t>(cap);
		rt.memset_bytes(table, 0xff, 4 * (size_t)cap);
		KDupInsert ki{rfa.d_cands, act, bc_of, table, cap - 1};
		rt.launch_wide("dup_insert", R, ki);
		KDupMark km{rfa.d_cands, act, bc_of, table, cap - 1, post};
		rt.launch_wide("dup_mark", R, km);
		SplitRec *split = rt.template alloc<SplitRec>(R + 1);
		KSplit ks{rfa.d_cands, post, rfa.d_cand_off, act, b.lens, rfa.penalty, rfa.d_cen_start, rfa.d_cen_end, split};
		rt.launch_wide("split", R, ks);
		res.d_post = post; res.d_split = split; res.d_mm_ref = mm_ref; res.d_mm_read = mm_read; res.n_mm = NM; res.done = true;
		return 0;
	}
	static void fetch(Pipeline<RT> &pipe, const typename Pipeline<RT>::DeviceBatch &b, const RfaResult &rfa, const PostResult &res, CandPost *post, SplitRec *split,
	                  int32_t *mm_ref, int32_t *mm_read)
	{
		RT &rt = pipe.rt;
		if (post) rt.d2h(post, res.d_post, sizeof(CandPost) * (size_t)rfa.n_cands);
		if (split) rt.d2h(split, res.d_split, sizeof(SplitRec) * (size_t)b.n_reads);
		if (mm_ref) rt.d2h(mm_ref, res.d_mm_ref, 4 * (size_t)res.n_mm);
		if (mm_read) rt.d2h(mm_read, res.d_mm_read, 4 * (size_t)res.n_mm);
	}
};

} // namespace arx
