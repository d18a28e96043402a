// dev_chain.h -- seed chaining and chain filtering, one read per thread.
// Agrees with mem_chain / test_and_merge / mem_chain_weight / mem_chain_flt (bwamem.c:190-385) including the
// B-tree lookup semantics of kbtree.h (which duplicate key is found depends on the tree shape) and the unstable
// introsort used to rank chains by weight.
#pragma once
#include "arx_dev.h"

#ifndef ARX_CHAIN_T
#define ARX_CHAIN_T(k) do {} while (0)   // diagnostics hook (tools: per-phase clock of the heavy-read kernel)
#endif

namespace arx {

// B-tree of minimum degree t = 5: KB_DEFAULT_SIZE 512 with a 40-byte key gives t = ((512-4-8)/(8+40)+1)>>1 (kbtree.h:56,388)
constexpr int BT_T = 5, BT_MAXK = 2 * BT_T - 1;
// kpos[j] = Chain::pos of key[j], kept inside the node: a node visit is then one round trip to memory (all key positions of the
// node at once) instead of a binary search of dependent key -> chain -> pos loads -- what a read in a 200-copy repeat spends its
// time on (a thread walks the tree once per seed occurrence, several hundred times)
struct BtNode { int32_t is_internal, n; int32_t key[BT_MAXK]; int32_t child[BT_MAXK + 1]; int64_t kpos[BT_MAXK]; };

struct BTree {
	BtNode *nodes;      // per-read slice of the node pool
	const Chain *ch;    // per-read slice of the chain pool (keys are chain indices ordered by Chain::pos)
	int n_nodes, cap_nodes, root, n_keys;
};

ARX_DEVI int bt_new(BTree &b)
{
	if (b.n_nodes >= b.cap_nodes) return -1;
	BtNode &x = b.nodes[b.n_nodes];
	x.is_internal = 0; x.n = 0;
	return b.n_nodes++;
}

// __kb_getp_aux (kbtree.h:117-131): first key >= pos inside one node; *r = sign(pos - key[result]).  The reference finds that
// key by binary search; the first key >= pos of a sorted node is the same whichever way it is found, here by counting the keys
// below pos (all loads independent of each other).
ARX_DEVI int bt_getp_aux(const BTree &b, int xi, int64_t pos, int *r)
{
	const BtNode &x = b.nodes[xi];
	const int n = x.n;
	if (n == 0) return -1;
	int64_t kp[BT_MAXK];
#pragma unroll
	for (int j = 0; j < BT_MAXK; ++j) kp[j] = x.kpos[j];
	int begin = 0;
#pragma unroll
	for (int j = 0; j < BT_MAXK; ++j) begin += (j < n && kp[j] < pos) ? 1 : 0;
	if (begin == n) { *r = 1; return n - 1; }
	int64_t kb = kp[0];
#pragma unroll
	for (int j = 1; j < BT_MAXK; ++j) if (j == begin) kb = kp[j];
	*r = (pos > kb) - (pos < kb);
	if (*r < 0) --begin;
	return begin;
}

// kb_intervalp (kbtree.h:151-168), `lower` only -- mem_chain ignores `upper` (bwamem.c:289-290)
ARX_DEVI int bt_lower(const BTree &b, int64_t pos)
{
	int xi = b.root, lower = -1, r = 0;
	for (;;) {
		const BtNode &x = b.nodes[xi];
		int i = bt_getp_aux(b, xi, pos, &r);
		if (i >= 0 && r == 0) return x.key[i];
		if (i >= 0) lower = x.key[i];
		if (!x.is_internal) return lower;
		xi = x.child[i + 1];
	}
}

ARX_DEVI bool bt_split(BTree &b, int xi, int i, int yi) // __kb_split (kbtree.h:177-192)
{
	int zi = bt_new(b);
	if (zi < 0) return false;
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
	return true;
}

ARX_DEV bool bt_put(BTree &b, int ci) // kb_putp + __kb_putp_aux (kbtree.h:193-226)
{
	int64_t pos = b.ch[ci].pos;
	int r;
	++b.n_keys;
	if (b.nodes[b.root].n == BT_MAXK) {
		int si = bt_new(b), old = b.root;
		if (si < 0) return false;
		b.root = si; b.nodes[si].is_internal = 1; b.nodes[si].n = 0; b.nodes[si].child[0] = old;
		if (!bt_split(b, si, 0, old)) return false;
	}
	int xi = b.root;
	for (;;) {
		BtNode &x = b.nodes[xi];
		if (!x.is_internal) {
			int i = bt_getp_aux(b, xi, pos, &r);
			for (int j = x.n - 1; j > i; --j) { x.key[j + 1] = x.key[j]; x.kpos[j + 1] = x.kpos[j]; }
			x.key[i + 1] = ci; x.kpos[i + 1] = pos;
			++x.n;
			return true;
		}
		int i = bt_getp_aux(b, xi, pos, &r) + 1;
		if (b.nodes[x.child[i]].n == BT_MAXK) {
			if (!bt_split(b, xi, i, x.child[i])) return false;
			if (pos > x.kpos[i]) ++i;
		}
		xi = x.child[i];
	}
}

// in-order traversal (kbtree.h:352-375) with an explicit stack; the tree height is tiny (<= 8 for 10^5 keys)
ARX_DEV int bt_traverse(const BTree &b, int *out)
{
	int st_x[16], st_i[16], st_p[16], top = 0, n = 0;
	st_x[0] = b.root; st_i[0] = 0; st_p[0] = 0;
	while (top >= 0) {
		const BtNode &x = b.nodes[st_x[top]];
		int i = st_i[top];
		if (st_p[top] == 0) { // first descend into child i
			st_p[top] = 1;
			if (x.is_internal) { ++top; st_x[top] = x.child[i]; st_i[top] = 0; st_p[top] = 0; continue; }
		}
		if (i < x.n) { out[n++] = x.key[i]; st_i[top] = i + 1; st_p[top] = 0; } // then emit key i
		else --top;
	}
	return n;
}

// test_and_merge (bwamem.c:190-211) on a chain whose seeds form a linked list (head = seeds[0], tail = last)
ARX_DEVI int test_and_merge(int64_t l_pac, Chain &c, const Seed *occ, int *next, int g, int seed_rid)
{
	const Seed p = occ[g], first = occ[c.head], last = occ[c.tail];
	int64_t qend = last.qbeg + last.len, rend = last.rbeg + last.len;
	if (seed_rid != c.rid) return 0;
	if (p.qbeg >= first.qbeg && p.qbeg + p.len <= qend && p.rbeg >= first.rbeg && p.rbeg + p.len <= rend) return 1; // contained
	if ((last.rbeg < l_pac || first.rbeg < l_pac) && p.rbeg >= l_pac) return 0; // different strand
	int64_t x = p.qbeg - last.qbeg, y = p.rbeg - last.rbeg;
	if (y >= 0 && x - y <= OPT_W && y - x <= OPT_W && x - last.len < OPT_MAX_CHAIN_GAP && y - last.len < OPT_MAX_CHAIN_GAP) {
		next[c.tail] = g; next[g] = -1; c.tail = g; ++c.n;
		return 1;
	}
	return 0;
}

ARX_DEV int chain_weight(const Chain &c, const Seed *occ, const int *next) // mem_chain_weight (bwamem.c:213-234): both coverages in one walk of the list
{
	int64_t qend = 0, rend = 0;
	int wq = 0, wr = 0;
	for (int g = c.head; g >= 0; g = next[g]) {
		const Seed s = occ[g];
		if (s.qbeg >= qend) wq += s.len;
		else if (s.qbeg + s.len > qend) wq += (int)(s.qbeg + s.len - qend);
		qend = qend > s.qbeg + s.len ? qend : s.qbeg + s.len;
		if (s.rbeg >= rend) wr += s.len;
		else if (s.rbeg + s.len > rend) wr += (int)(s.rbeg + s.len - rend);
		rend = rend > s.rbeg + s.len ? rend : s.rbeg + s.len;
	}
	const int w = wr < wq ? wr : wq;
	return w < 1 << 30 ? w : (1 << 30) - 1;
}

struct WeightGt { const Chain *c; ARX_DEVI bool operator()(int a, int b) const { return c[a].w > c[b].w; } };

// frac_rep: share of the query covered by seeds occurring more than max_occ times (bwamem.c:265-272)
ARX_DEV float chain_frac_rep(int len, const Biv *intv, int n_intv)
{
	int b = 0, e = 0, l_rep = 0;
	for (int i = 0; i < n_intv; ++i) {
		int sb = (int)(intv[i].info >> 32), se = (int)(uint32_t)intv[i].info;
		if (intv[i].s <= (uint64_t)OPT_MAX_OCC) continue;
		if (sb > e) { l_rep += e - b; b = sb; e = se; }
		else e = e > se ? e : se;
	}
	l_rep += e - b;
	return (float)l_rep / len;
}

// mem_chain's loop (bwamem.c:273-307): every occurrence, in order, joins the chain the B-tree finds for it or starts a new one.
// occ_rid[g] = bns_intv2rid of occurrence g (KOccRid: computed for all occurrences of the batch side by side -- two binary searches
// over the contig table that this one-thread loop used to wait for, occurrence after occurrence).  Returns the number of chains, -1 on
// pool exhaustion; bt is left ready for bt_traverse.
ARX_DEV int chain_build(const IndexView &ix, const Seed *occ, const int32_t *occ_rid, int n_occ, int *next, Chain *ctmp, BTree &bt, BtNode *nodes, int cap_nodes, float frac_rep)
{
	bt.nodes = nodes; bt.ch = ctmp; bt.n_nodes = 0; bt.cap_nodes = cap_nodes; bt.n_keys = 0;
	bt.root = bt_new(bt);
	int n_ch = 0;
	for (int g = 0; g < n_occ; ++g) {
		const Seed s = occ[g];
		const int rid = occ_rid[g];
		if (rid < 0) continue; // spans contigs or the strand boundary
		bool to_add = true;
		if (bt.n_keys) {
			int lower = bt_lower(bt, s.rbeg);
			if (lower >= 0 && test_and_merge(ix.l_pac, ctmp[lower], occ, next, g, rid)) to_add = false;
		}
		if (to_add) {
			Chain &c = ctmp[n_ch];
			c.pos = s.rbeg; c.rid = rid; c.n = 1; c.head = c.tail = g; next[g] = -1;
			c.is_alt = ix.ann_alt[rid] ? 1 : 0;
			c.w = 0; c.kept = 0; c.first = -1; c.seed_off = 0; c.frac_rep = frac_rep;
			if (!bt_put(bt, n_ch)) return -1;
			++n_ch;
		}
	}
	return n_ch;
}

// the survivors of the filter with their seeds in list order, compacted (max_chain_extend = 1<<30 never triggers, bwamem.c:373-378)
ARX_DEV int chain_emit(int n, const int *ord, const Chain *ctmp, const Seed *occ, const int *next, Chain *cout, Seed *sout, int sout_base)
{
	int m = 0, so = 0;
	for (int i = 0; i < n; ++i) {
		const Chain &c = ctmp[ord[i]];
		if (c.kept == 0) continue;
		Chain o = c;
		o.seed_off = sout_base + so;
		for (int g = c.head; g >= 0; g = next[g]) sout[so++] = occ[g];
		cout[m++] = o;
	}
	return m;
}

// One read: occurrences [g0, g1) (already located, in interval order) -> filtered chains + their seeds, compacted.
// Pools are per-read slices: ctmp/cout/sout/next have g1-g0 slots, iscr 7*(g1-g0) ints, nodes cap_nodes entries.
// Returns the number of chains kept (mem_chain + mem_chain_flt), or -1 on pool exhaustion.
ARX_DEV int chain_and_filter(const IndexView &ix, int len, const Biv *intv, int n_intv, const Seed *occ, const int32_t *occ_rid, int n_occ,
                             int *next, Chain *ctmp, BtNode *nodes, int cap_nodes, int *iscr, Chain *cout, Seed *sout, int sout_base)
{
	if (len < OPT_MIN_SEED_LEN || n_occ == 0) return 0;
	ARX_CHAIN_T(0);
	BTree bt;
	const int n_ch = chain_build(ix, occ, occ_rid, n_occ, next, ctmp, bt, nodes, cap_nodes, chain_frac_rep(len, intv, n_intv));
	if (n_ch < 0) return -1;
	ARX_CHAIN_T(1);
	if (n_ch == 0) return 0;
	int *ord = iscr, *kept_idx = iscr + n_occ; // iscr: 7 * n_occ ints
	int *qb_ = iscr + 2 * n_occ, *qe_ = iscr + 3 * n_occ, *w_ = iscr + 4 * n_occ, *alt_ = iscr + 5 * n_occ, *first_ = iscr + 6 * n_occ; // by rank in `ord`
	int n = bt_traverse(bt, ord); // chains in key order = the array mem_chain returns
	// mem_chain_flt (bwamem.c:327-385)
	for (int i = 0; i < n; ++i) { Chain &c = ctmp[ord[i]]; c.first = -1; c.kept = 0; c.w = chain_weight(c, occ, next); }
	ARX_CHAIN_T(2);
	WeightGt gt; gt.c = ctmp;
	ks_introsort(n, ord, gt);
	ARX_CHAIN_T(3);
	// The filter compares every chain with every chain kept so far: quadratic for a read in a high-copy repeat (all its chains cover
	// the same query span and weigh the same, so nothing is dropped early).  What a comparison looks at -- query span (chn_beg /
	// chn_end, bwamem.c:325-326), weight, is_alt, `first` -- is laid out by rank once, and the kept chains are visited four at a
	// time: their loads do not depend on each other, the verdicts are then taken in order exactly as the reference's loop would.
	for (int i = 0; i < n; ++i) {
		const Chain &c = ctmp[ord[i]];
		qb_[i] = occ[c.head].qbeg; qe_[i] = occ[c.tail].qbeg + occ[c.tail].len; w_[i] = c.w; alt_[i] = c.is_alt; first_[i] = -1;
	}
	int n_kept = 0;
	ctmp[ord[0]].kept = 3;
	kept_idx[n_kept++] = 0;
	for (int i = 1; i < n; ++i) {
		const int bi = qb_[i], ei = qe_[i], wi = w_[i], alt_i = alt_[i];
		int large_ovlp = 0;
		bool dropped = false;
		for (int k0 = 0; k0 < n_kept && !dropped; k0 += 4) {
			int kj[4], bj[4], ej[4], wj[4], aj[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) kj[u] = kept_idx[k0 + u < n_kept ? k0 + u : n_kept - 1];
#pragma unroll
			for (int u = 0; u < 4; ++u) { bj[u] = qb_[kj[u]]; ej[u] = qe_[kj[u]]; wj[u] = w_[kj[u]]; aj[u] = alt_[kj[u]]; }
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				if (dropped || k0 + u >= n_kept) continue;
				const int b_max = bj[u] > bi ? bj[u] : bi, e_min = ej[u] < ei ? ej[u] : ei;
				if (e_min > b_max && (!aj[u] || alt_i)) {
					const int li = ei - bi, lj = ej[u] - bj[u];
					const int min_l = li < lj ? li : lj;
					if ((float)(e_min - b_max) >= min_l * OPT_MASK_LEVEL && min_l < OPT_MAX_CHAIN_GAP) {
						large_ovlp = 1;
						if (first_[kj[u]] < 0) first_[kj[u]] = i;
						if ((float)wi < wj[u] * OPT_DROP_RATIO && wj[u] - wi >= OPT_MIN_SEED_LEN << 1) dropped = true;
					}
				}
			}
		}
		if (!dropped) { kept_idx[n_kept++] = i; ctmp[ord[i]].kept = large_ovlp ? 2 : 3; }
	}
	for (int i = 0; i < n_kept; ++i) {
		const int f = first_[kept_idx[i]];
		ctmp[ord[kept_idx[i]]].first = f;
		if (f >= 0) ctmp[ord[f]].kept = 1;
	}
	ARX_CHAIN_T(4);
	const int m = chain_emit(n, ord, ctmp, occ, next, cout, sout, sout_base);
	ARX_CHAIN_T(5);
	return m;
}

} // namespace arx
