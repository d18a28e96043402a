// dev_chain_wave.h -- chaining and chain filtering of ONE read by a whole 64-lane wavefront (gfx950 only; arx_cold.hip: k_chain_heavy).
//
// A read in a high-copy repeat brings 65-800 seed occurrences and ends with 50-150 chains that all cover the same query span.  With one
// thread per read (dev_chain.h) such a read costs 2-5 ms -- a fifth of it walking the B-tree, the rest weighing the chains, ranking
// them and comparing every chain with every chain kept before it -- and a 667 k-read batch has some 1,800 of them.  Here the read's
// working set sits in LDS and the steps that do not depend on each other are spread over the lanes:
//   * mem_chain's loop itself (bwamem.c:273-307) is a chain of dependent B-tree updates: it stays sequential, but every lane walks the tree
//     in step and a node's keys are compared side by side (w_chain_build below);
//   * the chains' weights (mem_chain_weight): one chain per lane;
//   * ranking by weight: klib's introsort reproduced by the wavefront (w_introsort, dev_regs_wave.h; its order of equal weights is part of
//     the result);
//   * mem_chain_flt's loop (bwamem.c:340-371): chain i against the chains kept so far, 64 of them per step -- each lane tests one kept
//     chain, ballots give the first kept chain that drops i; `first` is set on the chains the reference's loop would have visited
//     (up to and including that one), exactly as the serial loop does;
//   * lane 0 writes the surviving chains and their seeds.
// Results are those of chain_and_filter() bit for bit.
#pragma once
#include "dev_chain.h"
#include "dev_regs_wave.h" // w_introsort

namespace arx {

constexpr int CHAIN_WAVE_MAX = 832; // = CHAIN_LDS_OCC of pipeline.h (13 x 64): the most chains a read of the heavy kernel can have

// ---- mem_chain's loop with the wavefront in uniform control flow: every lane follows the same path through the B-tree (kbtree.h semantics as in
// dev_chain.h), a node's nine key positions are compared by nine lanes at once (ballot + popcount instead of nine compare/select chains in
// one lane), the shift that makes room in a leaf is one step, LDS reads are broadcasts; lane 0 does the remaining writes.  One lane alone
// spent ~2 us per occurrence here, almost all of it issuing instructions.
__device__ __forceinline__ int w_bt_new(BTree &b)
{
	if (b.n_nodes >= b.cap_nodes) return -1;
	if (threadIdx.x == 0) { BtNode &x = b.nodes[b.n_nodes]; x.is_internal = 0; x.n = 0; }
	__syncthreads();
	return b.n_nodes++;
}
// A node as the wavefront holds it after ONE trip to LDS: lane l has word l of the node's first 21 words (is_internal, n, key[9], child[10])
// and, for l < 9, the position of key l; single values come out by v_readlane with a wave-uniform index.
struct WNode { int w, k; int64_t kp; }; // k: key l itself (the leaf insert moves key and position of lane l together)
__device__ __forceinline__ WNode w_node_load(const BtNode &x)
{
	const int lane = threadIdx.x;
	WNode v;
	v.w = ((const int *)&x)[lane < 21 ? lane : 0];
	v.k = x.key[lane < BT_MAXK ? lane : 0];
	v.kp = x.kpos[lane < BT_MAXK ? lane : 0];
	return v;
}
__device__ __forceinline__ int w_node_word(const WNode &v, int k) { return __builtin_amdgcn_readlane(v.w, k); }
__device__ __forceinline__ int w_node_n(const WNode &v) { return w_node_word(v, 1); }
__device__ __forceinline__ bool w_node_internal(const WNode &v) { return w_node_word(v, 0) != 0; }
__device__ __forceinline__ int w_node_key(const WNode &v, int j) { return w_node_word(v, 2 + j); }
__device__ __forceinline__ int w_node_child(const WNode &v, int j) { return w_node_word(v, 2 + BT_MAXK + j); }
__device__ __forceinline__ int64_t w_node_kpos(const WNode &v, int j)
{
	return (int64_t)((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)((uint64_t)v.kp >> 32), j) << 32 | (uint32_t)__builtin_amdgcn_readlane((int)v.kp, j));
}
__device__ __forceinline__ int w_bt_getp(const WNode &v, int64_t pos, int *r) // bt_getp_aux
{
	const int lane = threadIdx.x, n = w_node_n(v);
	if (n == 0) return -1;
	int begin = __builtin_popcountll(__ballot(lane < n && v.kp < pos));
	if (begin == n) { *r = 1; return n - 1; }
	const int64_t kb = w_node_kpos(v, begin);
	*r = (pos > kb) - (pos < kb);
	if (*r < 0) --begin;
	return begin;
}
__device__ int w_bt_lower(const BTree &b, int64_t pos) // bt_lower
{
	int xi = b.root, lower = -1, r = 0;
	for (;;) {
		const WNode v = w_node_load(b.nodes[xi]);
		const int i = w_bt_getp(v, pos, &r);
		if (i >= 0 && r == 0) return w_node_key(v, i);
		if (i >= 0) lower = w_node_key(v, i);
		if (!w_node_internal(v)) return lower;
		xi = w_node_child(v, i + 1);
	}
}
__device__ bool w_bt_split(BTree &b, int xi, int i, int yi) // bt_split: lane 0 moves the keys (one split per five insertions)
{
	const int zi = w_bt_new(b);
	if (zi < 0) return false;
	if (threadIdx.x == 0) {
		BtNode &x = b.nodes[xi], &y = b.nodes[yi], &z = b.nodes[zi];
		z.is_internal = y.is_internal;
		z.n = BT_T - 1;
		for (int j = 0; j < BT_T - 1; ++j) { z.key[j] = y.key[BT_T + j]; z.kpos[j] = y.kpos[BT_T + j]; }
		if (y.is_internal) for (int j = 0; j < BT_T; ++j) z.child[j] = y.child[BT_T + j];
		y.n = BT_T - 1;
		for (int j = x.n; j > i; --j) x.child[j + 1] = x.child[j];
		x.child[i + 1] = zi;
		for (int j = x.n - 1; j >= i; --j) { x.key[j + 1] = x.key[j]; x.kpos[j + 1] = x.kpos[j]; }
		x.key[i] = y.key[BT_T - 1]; x.kpos[i] = y.kpos[BT_T - 1];
		++x.n;
	}
	__syncthreads();
	return true;
}
__device__ bool w_bt_put(BTree &b, int ci, int64_t pos) // bt_put
{
	const int lane = threadIdx.x;
	int r;
	++b.n_keys;
	if (b.nodes[b.root].n == BT_MAXK) {
		const int si = w_bt_new(b), old = b.root;
		if (si < 0) return false;
		b.root = si;
		if (lane == 0) { b.nodes[si].is_internal = 1; b.nodes[si].n = 0; b.nodes[si].child[0] = old; }
		__syncthreads();
		if (!w_bt_split(b, si, 0, old)) return false;
	}
	int xi = b.root;
	for (;;) {
		BtNode &x = b.nodes[xi];
		const WNode v = w_node_load(x);
		if (!w_node_internal(v)) {
			const int i = w_bt_getp(v, pos, &r), n = w_node_n(v);
			// keys i+1 .. n-1 one place up: lane l of the loaded node holds key l and its position
			if (lane > i && lane < n) { x.key[lane + 1] = v.k; x.kpos[lane + 1] = v.kp; }
			if (lane == 0) { x.key[i + 1] = ci; x.kpos[i + 1] = pos; x.n = n + 1; }
			__syncthreads();
			return true;
		}
		int i = w_bt_getp(v, pos, &r) + 1;
		const int ch = w_node_child(v, i);
		if (b.nodes[ch].n == BT_MAXK) {
			if (!w_bt_split(b, xi, i, ch)) return false;
			if (pos > x.kpos[i]) ++i;
			xi = x.child[i];
		} else xi = ch;
	}
}
// test_and_merge of dev_chain.h; c in LDS, writes by lane 0
__device__ int w_test_and_merge(int64_t l_pac, Chain &c, const Seed *occ, int *next, int g, int seed_rid)
{
	const Seed p = occ[g], first = occ[c.head], last = occ[c.tail];
	const int64_t qend = last.qbeg + last.len, rend = last.rbeg + last.len;
	if (seed_rid != c.rid) return 0;
	if (p.qbeg >= first.qbeg && p.qbeg + p.len <= qend && p.rbeg >= first.rbeg && p.rbeg + p.len <= rend) return 1; // contained
	if ((last.rbeg < l_pac || first.rbeg < l_pac) && p.rbeg >= l_pac) return 0; // different strand
	const int64_t x = p.qbeg - last.qbeg, y = p.rbeg - last.rbeg;
	if (y >= 0 && x - y <= OPT_W && y - x <= OPT_W && x - last.len < OPT_MAX_CHAIN_GAP && y - last.len < OPT_MAX_CHAIN_GAP) {
		if (threadIdx.x == 0) { next[c.tail] = g; next[g] = -1; c.tail = g; ++c.n; }
		__syncthreads();
		return 1;
	}
	return 0;
}
// chain_build of dev_chain.h
__device__ int w_chain_build(const IndexView &ix, const Seed *occ, const int32_t *occ_rid, int n_occ, int *next, Chain *ctmp, BTree &bt, BtNode *nodes, int cap_nodes, float frac_rep)
{
	bt.nodes = nodes; bt.ch = ctmp; bt.n_nodes = 0; bt.cap_nodes = cap_nodes; bt.n_keys = 0;
	bt.root = w_bt_new(bt);
	int n_ch = 0;
	for (int g = 0; g < n_occ; ++g) {
		const Seed s = occ[g];
		const int rid = occ_rid[g];
		if (rid < 0) continue; // spans contigs or the strand boundary
		bool to_add = true;
		if (bt.n_keys) {
			const int lower = w_bt_lower(bt, s.rbeg);
			if (lower >= 0 && w_test_and_merge(ix.l_pac, ctmp[lower], occ, next, g, rid)) to_add = false;
		}
		if (to_add) {
			const int is_alt = ix.ann_alt[rid] ? 1 : 0;
			if (threadIdx.x == 0) {
				Chain &c = ctmp[n_ch];
				c.pos = s.rbeg; c.rid = rid; c.n = 1; c.head = c.tail = g; next[g] = -1;
				c.is_alt = is_alt;
				c.w = 0; c.kept = 0; c.first = -1; c.seed_off = 0; c.frac_rep = frac_rep;
			}
			__syncthreads();
			if (!w_bt_put(bt, n_ch, s.rbeg)) return -1;
			++n_ch;
		}
	}
	return n_ch;
}

// Pools as for chain_and_filter(), all in LDS except cout / sout; every lane of the wavefront calls this with the same arguments.
// Returns (on every lane) the number of chains kept, or -1 on pool exhaustion.
__device__ int w_chain_and_filter(const IndexView &ix, int len, const Biv *intv, int n_intv, const Seed *occ, const int32_t *occ_rid, int n_occ,
                                  int *next, Chain *ctmp, BtNode *nodes, int cap_nodes, int *iscr, Chain *cout, Seed *sout, int sout_base, int *xch /* 4 ints of LDS */)
{
	const int lane = threadIdx.x;
	if (len < OPT_MIN_SEED_LEN || n_occ == 0) return 0;
	int *ord = iscr, *kept_idx = iscr + n_occ;
	int *qb_ = iscr + 2 * n_occ, *qe_ = iscr + 3 * n_occ, *w_ = iscr + 4 * n_occ, *alt_ = iscr + 5 * n_occ, *first_ = iscr + 6 * n_occ;
	if (lane == 0) ARX_CHAIN_T(0);
	BTree bt;
	const int n_built = w_chain_build(ix, occ, occ_rid, n_occ, next, ctmp, bt, nodes, cap_nodes, chain_frac_rep(len, intv, n_intv));
	if (lane == 0) {
		ARX_CHAIN_T(1);
		xch[0] = n_built;
		if (n_built > 0) bt_traverse(bt, ord); // chains in key order = the array mem_chain returns
	}
	__syncthreads();
	const int n = xch[0];
	if (n <= 0) return n;
	for (int i = lane; i < n; i += 64) { Chain &c = ctmp[ord[i]]; c.first = -1; c.kept = 0; c.w = chain_weight(c, occ, next); }
	__syncthreads();
	if (lane == 0) ARX_CHAIN_T(2);
	{
		WeightGt gt; gt.c = ctmp;
		if (n <= 256) w_introsort<256>(n, ord, gt, qb_, qe_); // the rank arrays are filled after the sort
		else w_introsort<CHAIN_WAVE_MAX>(n, ord, gt, qb_, qe_);
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
