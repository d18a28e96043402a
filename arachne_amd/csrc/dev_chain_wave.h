// dev_chain_wave.h -- chaining and chain filtering of ONE read by a whole 64-lane wavefront (gfx950 only; arx_cold.hip: k_chain_heavy).
//
// A read in a high-copy repeat brings 65-800 seed occurrences and ends with 50-150 chains that all cover the same query span.  With one
// thread per read (dev_chain.h) such a read costs 2-5 ms -- a fifth of it walking the B-tree, the rest weighing the chains, ranking
// them and comparing every chain with every chain kept before it -- and a 667 k-read batch has some 1,800 of them.  Here the read's
// working set sits in LDS and the steps that do not depend on each other are spread over the lanes:
//   * mem_chain's loop itself (bwamem.c:273-307) is a chain of dependent B-tree updates: lane 0 runs chain_build() of dev_chain.h;
//   * the chains' weights (mem_chain_weight): one chain per lane;
//   * ranking by weight: klib's introsort reproduced by the wavefront (w_introsort, dev_regs_wave.h; its order of equal weights is part of
//     the result), lane 0 alone for more than 256 chains;
//   * mem_chain_flt's loop (bwamem.c:340-371): chain i against the chains kept so far, 64 of them per step -- each lane tests one kept
//     chain, ballots give the first kept chain that drops i; `first` is set on the chains the reference's loop would have visited
//     (up to and including that one), exactly as the serial loop does;
//   * lane 0 writes the surviving chains and their seeds.
// Results are those of chain_and_filter() bit for bit.
#pragma once
#include "dev_chain.h"
#include "dev_regs_wave.h" // w_introsort

namespace arx {

// Pools as for chain_and_filter(), all in LDS except cout / sout; every lane of the wavefront calls this with the same arguments.
// Returns (on every lane) the number of chains kept, or -1 on pool exhaustion.
__device__ int w_chain_and_filter(const IndexView &ix, int len, const Biv *intv, int n_intv, const Seed *occ, const int32_t *occ_rid, int n_occ,
                                  int *next, Chain *ctmp, BtNode *nodes, int cap_nodes, int *iscr, Chain *cout, Seed *sout, int sout_base, int *xch /* 4 ints of LDS */)
{
	const int lane = threadIdx.x;
	if (len < OPT_MIN_SEED_LEN || n_occ == 0) return 0;
	int *ord = iscr, *kept_idx = iscr + n_occ;
	int *qb_ = iscr + 2 * n_occ, *qe_ = iscr + 3 * n_occ, *w_ = iscr + 4 * n_occ, *alt_ = iscr + 5 * n_occ, *first_ = iscr + 6 * n_occ;
	if (lane == 0) {
		ARX_CHAIN_T(0);
		BTree bt;
		const int n_ch = chain_build(ix, occ, occ_rid, n_occ, next, ctmp, bt, nodes, cap_nodes, chain_frac_rep(len, intv, n_intv));
		xch[0] = n_ch;
		ARX_CHAIN_T(1);
		if (n_ch > 0) bt_traverse(bt, ord); // chains in key order = the array mem_chain returns
	}
	__syncthreads();
	const int n = xch[0];
	if (n <= 0) return n;
	for (int i = lane; i < n; i += 64) { Chain &c = ctmp[ord[i]]; c.first = -1; c.kept = 0; c.w = chain_weight(c, occ, next); }
	__syncthreads();
	if (lane == 0) ARX_CHAIN_T(2);
	{
		WeightGt gt; gt.c = ctmp;
		if (n <= W_SORT_MAX) w_introsort(n, ord, gt, qb_, qe_); // the rank arrays are filled after the sort
		else { if (lane == 0) ks_introsort(n, ord, gt); __syncthreads(); }
	}
	if (lane == 0) ARX_CHAIN_T(3);
	for (int i = lane; i < n; i += 64) {
		const Chain &c = ctmp[ord[i]];
		qb_[i] = occ[c.head].qbeg; qe_[i] = occ[c.tail].qbeg + occ[c.tail].len; w_[i] = c.w; alt_[i] = c.is_alt; first_[i] = -1;
	}
	if (lane == 0) { ctmp[ord[0]].kept = 3; kept_idx[0] = 0; }
	__syncthreads();
	int n_kept = 1;
	for (int i = 1; i < n; ++i) {
		const int bi = qb_[i], ei = qe_[i], wi = w_[i], alt_i = alt_[i];
		bool large_ovlp = false, dropped = false;
		for (int k0 = 0; k0 < n_kept && !dropped; k0 += 64) {
			const int k = k0 + lane;
			bool ovl = false, drop = false;
			int kj = 0;
			if (k < n_kept) {
				kj = kept_idx[k];
				const int bj = qb_[kj], ej = qe_[kj], wj = w_[kj], aj = alt_[kj];
				const int b_max = bj > bi ? bj : bi, e_min = ej < ei ? ej : ei;
				if (e_min > b_max && (!aj || alt_i)) {
					const int li = ei - bi, lj = ej - bj;
					const int min_l = li < lj ? li : lj;
					if ((float)(e_min - b_max) >= min_l * OPT_MASK_LEVEL && min_l < OPT_MAX_CHAIN_GAP) {
						ovl = true;
						drop = (float)wi < wj * OPT_DROP_RATIO && wj - wi >= OPT_MIN_SEED_LEN << 1;
					}
				}
			}
			const uint64_t dm = __ballot(drop);
			uint64_t om = __ballot(ovl);
			if (dm) { // the serial loop stops at the first kept chain that drops i: what lies behind it is never visited
				const int j = __builtin_ctzll(dm);
				om &= j == 63 ? ~0ull : (1ull << (j + 1)) - 1;
				dropped = true;
			}
			if (om) large_ovlp = true;
			if ((om >> lane & 1) && first_[kj] < 0) first_[kj] = i;
		}
		if (!dropped) {
			if (lane == 0) { kept_idx[n_kept] = i; ctmp[ord[i]].kept = large_ovlp ? 2 : 3; }
			++n_kept;
		}
		__syncthreads();
	}
	for (int i = lane; i < n_kept; i += 64) ctmp[ord[kept_idx[i]]].first = first_[kept_idx[i]];
	__syncthreads();
	for (int i = lane; i < n_kept; i += 64) { const int f = first_[kept_idx[i]]; if (f >= 0) ctmp[ord[f]].kept = 1; }
	__syncthreads();
	if (lane == 0) { ARX_CHAIN_T(4); xch[1] = chain_emit(n, ord, ctmp, occ, next, cout, sout, sout_base); ARX_CHAIN_T(5); }
	__syncthreads();
	return xch[1];
}

} // namespace arx
