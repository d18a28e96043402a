// pipeline_rfa.h -- the RFA stage on top of a batch that has gone through ARX_STAGE_ALN: candidates per read
// (GetChains/GetAlignments), per-barcode joint placement (tagBestAlignments ... optimizer.Optimize) and the inputs of
// estimateMapQualities on the device; the floating-point tail of estimateMapQualities (aligner.go:797-922) on the host.
#pragma once
#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "pipeline.h"
#include "dev_rfa.h"

namespace arx {

struct KCandCount { // candidates per read: one per region, or the placeholder
	const int32_t *n_regs; int32_t *n_cand;
	ARX_DEV void operator()(int r, int) const { n_cand[r] = n_regs[r] ? n_regs[r] : 1; }
};
struct KCandBuild {
	IndexView ix; const int32_t *preg_off, *n_regs, *c_reg_off, *cand_off; const Reg *pregs; const Aln *alns; const uint32_t *cig; int cig_w; Cand *cands;
	ARX_DEV void operator()(int r, int) const { cand_build_read(ix, r, pregs, alns, cig, cig_w, preg_off[r], n_regs[r], c_reg_off[r], cands + cand_off[r]); }
};
struct KRfa { // one barcode per workgroup
	const int32_t *cand_off; const int32_t *bc_read_off; const uint8_t *do_rfa; const int64_t *scr_off; int pen_int, n_seqs;
	const double *p10h; Cand *cands; int32_t *scratch; RfaBarcodeOut *out;
	template <class B> ARX_DEV void operator()(int b, B &blk) const
	{
		const int r0 = bc_read_off[b], r1 = bc_read_off[b + 1];
		rfa_barcode(blk, cands + cand_off[r0], cand_off + r0, r1 - r0, cand_off[r1] - cand_off[r0], r0, do_rfa[b], pen_int, n_seqs, p10h, scratch + scr_off[b], out + b);
	}
};

struct KMapqPair { // one candidate per lane: its best pair score over the mate's candidates (the quadratic part of estimateMapQualities)
	const Cand *cands; const int32_t *cand_off; int penalty; int32_t *pair_best;
	ARX_DEV void operator()(int i, int) const
	{
		int best2 = 0;
		const int mr = cands[i].read ^ 1;
		pair_best[i] = (cands[i].in_filtered && rfa_pair_best2(cands, i, cand_off[mr], cand_off[mr + 1], 2 * penalty, &best2)) ? best2 : RFA_NO_PAIR;
	}
};
struct KMapq { // one read per lane: MAPQ of its active candidate; values a few ulp could change are queued for the host
	Cand *cands; const int32_t *cand_off; const int32_t *lens; const int32_t *bc_read_off; int n_barcodes; const double *log_mol_pen;
	int penalty; const int64_t *cen_start, *cen_end; double guard; int32_t *flagged, *n_flagged; const int32_t *pair_best;
	ARX_DEV void operator()(int r, int) const
	{
		int lo = 0, hi = n_barcodes;
		while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (bc_read_off[mid] <= r) lo = mid; else hi = mid; }
		const int mr = r ^ 1;
		int a; double largest;
		const double v = rfa_mapq_value(cands, cand_off[r], cand_off[r + 1], cand_off[mr], cand_off[mr + 1], lens[r], log_mol_pen[lo], penalty, &a, &largest, pair_best);
		cands[a].mapq = rfa_mapq_final(v, cands[a], cen_start, cen_end); // the mate's lane reads other fields of this record, never mapq
		if (rfa_mapq_needs_host(v, largest, guard)) flagged[ARX_ATOMIC_INC(n_flagged)] = r;
	}
};
struct KMapqPatch { Cand *cands; const int32_t *idx, *val; ARX_DEV void operator()(int k, int) const { cands[idx[k]].mapq = val[k]; } };

struct RfaResult {
	std::vector<int32_t> cand_off; std::vector<RfaBarcodeOut> bc;
	Cand *d_cands = nullptr; int64_t n_cands = 0; int64_t n_host_mapq = 0;
	// kept in HBM for the passes that follow (pipeline_post.h)
	int32_t *d_cand_off = nullptr, *d_bc_read_off = nullptr; int64_t *d_cen_start = nullptr, *d_cen_end = nullptr; int n_barcodes = 0, penalty = 0;
};

template <class RT> struct RfaStage {
	// bc_pair_off[n_barcodes + 1]: pair offsets of whole barcodes inside the batch; do_rfa[b] = worthRunningRFA (aligner.go:1018-1030)
	static int run(Pipeline<RT> &pipe, const typename Pipeline<RT>::DeviceBatch &b, typename Pipeline<RT>::Work &w, int n_barcodes,
	               const int64_t *bc_pair_off, const uint8_t *do_rfa, int penalty, const int64_t *cen_start, const int64_t *cen_end,
	               const int32_t *lens_host, RfaResult &res)
	{
		RT &rt = pipe.rt;
		const int R = b.n_reads;
		int32_t *n_cand = rt.template alloc<int32_t>(R + 1), *cand_off = rt.template alloc<int32_t>(R + 2);
		KCandCount kc{w.n_regs, n_cand};
		rt.launch_wide("cand_count", R, kc);
		const int64_t NC = rt.exclusive_scan(n_cand, cand_off, R);
		Cand *cands = rt.template alloc<Cand>((size_t)NC + 1);
		KCandBuild kb{pipe.ix, w.preg_off, w.n_regs, w.c_reg_off, cand_off, w.pregs, w.alns, w.cig, w.cig_w, cands};
		rt.launch_wide("cand_build", R, kb);
		res.cand_off.resize(R + 1);
		rt.d2h(res.cand_off.data(), cand_off, 4 * (size_t)(R + 1));
		// per-barcode scratch offsets
		std::vector<int32_t> bro(n_barcodes + 1);
		std::vector<int64_t> so(n_barcodes + 1);
		int64_t tot = 0;
		for (int i = 0; i <= n_barcodes; ++i) bro[i] = (int32_t)(2 * bc_pair_off[i]);
		for (int i = 0; i < n_barcodes; ++i) {
			so[i] = tot;
			tot += rfa_scratch_words(res.cand_off[bro[i + 1]] - res.cand_off[bro[i]], bro[i + 1] - bro[i], pipe.ix.n_seqs);
		}
		so[n_barcodes] = tot;
		std::vector<double> p10(2 * RFA_P10_HALF + 1);
		for (int x = -RFA_P10_HALF; x <= RFA_P10_HALF; ++x) p10[x + RFA_P10_HALF] = std::pow(10.0, 0.5 * x); // same expression as the oracle
		int32_t *d_bro = rt.template alloc<int32_t>(n_barcodes + 1);
		int64_t *d_so = rt.template alloc<int64_t>(n_barcodes + 1);
		uint8_t *d_flags = rt.template alloc<uint8_t>(n_barcodes + 1);
		double *d_p10 = rt.template alloc<double>(p10.size());
		int32_t *d_scr = rt.template alloc<int32_t>((size_t)tot + 1);
		RfaBarcodeOut *d_out = rt.template alloc<RfaBarcodeOut>(n_barcodes + 1);
		rt.h2d(d_bro, bro.data(), 4 * (size_t)(n_barcodes + 1)); rt.h2d(d_so, so.data(), 8 * (size_t)(n_barcodes + 1));
		rt.h2d(d_flags, do_rfa, n_barcodes); rt.h2d(d_p10, p10.data(), 8 * p10.size());
		KRfa kr{cand_off, d_bro, d_flags, d_so, penalty, pipe.ix.n_seqs, d_p10, cands, d_scr, d_out};
		// ARX_RFA_SMALL=1 (experiments): barcodes of TELLseq size in 256-lane workgroups (hip_block.h).  Measured at 4,333 barcodes x 77
		// pairs per batch: 23.3 ms against 7.4 ms with 1,024 lanes for every barcode -- the per-barcode phases are latency chains whose
		// length grows with the work per lane, and ten small workgroups per CU do not make up for it.  Default: off.
		std::vector<uint8_t> small(n_barcodes, 0);
		static const bool rfa_small = getenv("ARX_RFA_SMALL") && atoi(getenv("ARX_RFA_SMALL")) != 0;
		if (rfa_small)
			for (int i = 0; i < n_barcodes; ++i) small[i] = (bro[i + 1] - bro[i] <= 2 * SMALL_LANES && res.cand_off[bro[i + 1]] - res.cand_off[bro[i]] <= SMALL_SORT / 2) ? 1 : 0;
		rt.launch_block("rfa", n_barcodes, kr, rfa_small ? small.data() : nullptr);
#ifdef ARX_RFA_STATS
		{ unsigned long long h[8], z[8] = {0}; hipDeviceSynchronize(); hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rfa_t), sizeof h); hipMemcpyToSymbol(HIP_SYMBOL(g_rfa_t), z, sizeof z);
		  fprintf(stderr, "rfa phases over %d barcodes (ticks): R1 %llu  R2-presort %llu  sort %llu  R2-rest %llu  R5 optimize %llu  R6 %llu  tail %llu\n", n_barcodes, h[1], h[2], h[3], h[4], h[5], h[6], h[7]); }
#endif
		res.bc.resize(n_barcodes);
		rt.d2h(res.bc.data(), d_out, sizeof(RfaBarcodeOut) * (size_t)n_barcodes);
		// calculateLogMoleculePenalty (aligner.go:722-741) with libm on the host: one log10 per barcode
		std::vector<double> lmp(n_barcodes);
		for (int i = 0; i < n_barcodes; ++i) lmp[i] = (do_rfa[i] && res.bc[i].n_mol > 0) ? std::log10(res.bc[i].dna_len / 3200000000.0 * 0.05) : 0.0;
		double *d_lmp = rt.template alloc<double>(n_barcodes + 1);
		rt.h2d(d_lmp, lmp.data(), 8 * (size_t)n_barcodes);
		const int n_seqs = pipe.ix.n_seqs;
		int64_t *d_cs = nullptr, *d_ce = nullptr;
		if (cen_start && cen_end) {
			d_cs = rt.template alloc<int64_t>(n_seqs + 1); d_ce = rt.template alloc<int64_t>(n_seqs + 1);
			rt.h2d(d_cs, cen_start, 8 * (size_t)n_seqs); rt.h2d(d_ce, cen_end, 8 * (size_t)n_seqs);
		}
		int32_t *d_flag = rt.template alloc<int32_t>(R + 1);
		rt.memset0(w.counter, 4);
		const char *ge = getenv("ARX_MAPQ_GUARD"); // tests widen the guard to push every read through the host path
		const double guard = ge ? atof(ge) : RFA_MAPQ_GUARD;
		int32_t *pair_best = rt.template alloc<int32_t>((size_t)NC + 1);
		KMapqPair kp{cands, cand_off, penalty, pair_best};
		rt.launch_wide("mapq_pair", (int)NC, kp);
		KMapq km{cands, cand_off, b.lens, d_bro, n_barcodes, d_lmp, penalty, d_cs, d_ce, guard, d_flag, w.counter, pair_best};
		rt.launch_wide("mapq", R, km);
		const int nf = pipe.read_counter(w);
		res.n_host_mapq = nf;
		if (nf > 0) host_mapq(rt, nf, d_flag, cands, res, bro, lmp, penalty, cen_start, cen_end, lens_host);
		res.d_cands = cands; res.n_cands = NC;
		res.d_cand_off = cand_off; res.d_bc_read_off = d_bro; res.d_cen_start = d_cs; res.d_cen_end = d_ce; res.n_barcodes = n_barcodes; res.penalty = penalty;
		return 0;
	}

	// the reads KMapq queued: same evaluation with the host's libm on their (pair's) candidate records, patched into HBM
	static void host_mapq(RT &rt, int nf, const int32_t *d_flag, Cand *cands, const RfaResult &res, const std::vector<int32_t> &bro, const std::vector<double> &lmp,
	                      int penalty, const int64_t *cen_start, const int64_t *cen_end, const int32_t *lens)
	{
		std::vector<int32_t> fl(nf), pidx(nf), pval(nf);
		rt.d2h(fl.data(), d_flag, 4 * (size_t)nf);
		std::vector<Cand> buf;
		for (int k = 0; k < nf; ++k) {
			const int r = fl[k], p0 = r & ~1, base = res.cand_off[p0], n = res.cand_off[p0 + 2] - base;
			buf.resize(n);
			rt.d2h(buf.data(), cands + base, sizeof(Cand) * (size_t)n);
			const int bi = (int)(std::upper_bound(bro.begin(), bro.end(), r) - bro.begin()) - 1, mr = r ^ 1;
			int a; double largest;
			const double v = rfa_mapq_value(buf.data(), res.cand_off[r] - base, res.cand_off[r + 1] - base, res.cand_off[mr] - base, res.cand_off[mr + 1] - base,
			                                lens[r], lmp[bi], penalty, &a, &largest);
			pidx[k] = base + a; pval[k] = rfa_mapq_final(v, buf[a], cen_start, cen_end);
		}
		int32_t *d_i = rt.template alloc<int32_t>(nf), *d_v = rt.template alloc<int32_t>(nf);
		rt.h2d(d_i, pidx.data(), 4 * (size_t)nf); rt.h2d(d_v, pval.data(), 4 * (size_t)nf);
		KMapqPatch kp{cands, d_i, d_v};
		rt.launch_wide("mapq_patch", nf, kp);
	}

	// candidate records straight into the caller's arrays (arx_batch_rfa_fetch)
	static void fetch(Pipeline<RT> &pipe, const typename Pipeline<RT>::DeviceBatch &b, RfaResult &res, int32_t *cand_off, Cand *cands)
	{
		memcpy(cand_off, res.cand_off.data(), 4 * ((size_t)b.n_reads + 1));
		pipe.rt.d2h(cands, res.d_cands, sizeof(Cand) * (size_t)res.n_cands);
	}
};

} // namespace arx
