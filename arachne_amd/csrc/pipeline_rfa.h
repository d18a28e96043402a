// pipeline_rfa.h -- the RFA stage on top of a batch that has gone through ARX_STAGE_ALN: candidates per read
// (GetChains/GetAlignments), per-barcode joint placement (tagBestAlignments ... optimizer.Optimize) and the inputs of
// estimateMapQualities on the device; the floating-point tail of estimateMapQualities (aligner.go:797-922) on the host.
#pragma once
#include <cmath>
#include <vector>
#include "pipeline.h"
#include "dev_rfa.h"

namespace arx {

struct KCandCount { // candidates per read: one per region, or the placeholder
	const int32_t *n_regs; int32_t *n_cand;
	ARX_DEV void operator()(int r, int) const { n_cand[r] = n_regs[r] ? n_regs[r] : 1; }
};
struct KCandBuild {
	IndexView ix; const int32_t *preg_off, *n_regs, *cand_off; const Reg *pregs; const Aln *alns; const uint32_t *cig; int cig_w; Cand *cands;
	ARX_DEV void operator()(int r, int) const { cand_build_read(ix, r, pregs, alns, cig, cig_w, preg_off[r], n_regs[r], cands + cand_off[r]); }
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

struct RfaResult { std::vector<int32_t> cand_off; std::vector<Cand> cands; std::vector<RfaBarcodeOut> bc; };

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
		rt.launch("cand_count", R, kc);
		const int64_t NC = rt.exclusive_scan(n_cand, cand_off, R);
		Cand *cands = rt.template alloc<Cand>((size_t)NC + 1);
		KCandBuild kb{pipe.ix, w.preg_off, w.n_regs, cand_off, w.pregs, w.alns, w.cig, w.cig_w, cands};
		rt.launch("cand_build", R, kb);
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
		rt.launch_block("rfa", n_barcodes, kr);
		res.cands.resize((size_t)NC); res.bc.resize(n_barcodes);
		rt.d2h(res.cands.data(), cands, sizeof(Cand) * (size_t)NC);
		rt.d2h(res.bc.data(), d_out, sizeof(RfaBarcodeOut) * (size_t)n_barcodes);
		{ // `reg` leaves the device as a slot of the (capacity-sized) region pool; callers index the compact arrays of arx_batch_fetch
			std::vector<int32_t> po(R + 1), nr(R);
			rt.d2h(po.data(), w.preg_off, 4 * (size_t)(R + 1)); rt.d2h(nr.data(), w.n_regs, 4 * (size_t)R);
			int32_t compact = 0;
			for (int r = 0; r < R; ++r) {
				for (int i = res.cand_off[r]; i < res.cand_off[r + 1]; ++i) if (res.cands[i].reg >= 0) res.cands[i].reg = compact + (res.cands[i].reg - po[r]);
				compact += nr[r];
			}
		}
		finalize_mapq(res, bro, do_rfa, penalty, cen_start, cen_end, lens_host);
		return 0;
	}

	// estimateMapQualities' floating-point tail (aligner.go:825-918): method-1 normalisation over the top 15 pair scores,
	// method 2 from sum_move, min, cap at 60, centromere mask, int().  Same operations in the same order as the oracle.
	static void finalize_mapq(RfaResult &res, const std::vector<int32_t> &bro, const uint8_t *do_rfa, int penalty,
	                          const int64_t *cen_start, const int64_t *cen_end, const int32_t *lens)
	{
		const double pen = (double)penalty;
		const int pen2 = 2 * penalty;
		std::vector<double> scores;
		for (size_t bi = 0; bi + 1 < bro.size(); ++bi) {
			const RfaBarcodeOut &bo = res.bc[bi];
			const double log_mol_pen = (do_rfa[bi] && bo.n_mol > 0) ? std::log10(bo.dna_len / 3200000000.0 * 0.05) : 0.0;
			for (int r = bro[bi]; r < bro[bi + 1]; ++r) {
				const int mr = r ^ 1;
				const Cand *c = res.cands.data();
				scores.clear();
				double best_single = -1.7976931348623157e308;
				for (int j = res.cand_off[mr]; j < res.cand_off[mr + 1]; ++j) {
					if (!c[j].in_filtered) continue;
					const double s = 0.5 * c[j].lap2 + pen;
					if (s > best_single) best_single = s;
				}
				const double pseudo = -10.0 - ((double)lens[r] - 25.0) * 0.5 + log_mol_pen;
				scores.push_back(best_single + pseudo);
				int a = -1, am = -1;
				for (int i = res.cand_off[r]; i < res.cand_off[r + 1]; ++i) {
					if (!c[i].in_filtered) continue;
					if (c[i].active) a = i;
					double bs = -1.7976931348623157e308;
					for (int j = res.cand_off[mr]; j < res.cand_off[mr + 1]; ++j) {
						if (!c[j].in_filtered) continue;
						if (c[j].active) am = j;
						const double s = 0.5 * cand_pair_score2(c[i], c[j], pen2) + (c[i].active_molecule ? 0.0 : log_mol_pen);
						if (s > bs) bs = s;
					}
					scores.push_back(bs);
				}
				for (size_t x = 1; x < scores.size(); ++x) { double t = scores[x]; size_t y = x; while (y > 0 && scores[y - 1] > t) { scores[y] = scores[y - 1]; --y; } scores[y] = t; }
				double total = 0.0;
				const int ns = (int)scores.size();
				for (int x = ns - 1; x >= 0 && ns - x <= 15; --x) total += std::pow(10.0, scores[x]);
				const double sc = 0.5 * cand_pair_score2(c[a], c[am], pen2) + (c[a].active_molecule ? 0.0 : log_mol_pen);
				double mapq = -10.0 * std::log10(1.0 - std::pow(10.0, sc) / total);
				const double mmq = -10.0 * std::log10(1.0 - (1.0 / c[a].sum_move));
				mapq = (mapq != mapq || mmq != mmq) ? NAN : (mapq < mmq ? mapq : mmq);
				mapq = (mapq != mapq) ? NAN : (mapq < 60.0 ? mapq : 60.0);
				if (cen_start && c[a].rid >= 0 && c[a].pos > cen_start[c[a].rid] && c[a].pos <= cen_end[c[a].rid]) mapq = 0.0;
				res.cands[a].mapq = (mapq != mapq) ? (int)0x80000000 : (int)mapq;
			}
		}
	}
};

} // namespace arx
