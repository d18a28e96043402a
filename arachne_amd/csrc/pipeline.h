// pipeline.h -- host orchestration of one batch through the device path, written once against a small runtime
// policy RT (device memory, launches, scan).  The product instantiates it with HipRT (hip_rt.h): every stage below
// runs as a HIP kernel on the MI355X.
//
// Stage order = call order of the reference for one pair (/root/reference/src/gobwa/gobwa.go:226-337,400-415):
//   seed (mem_collect_intv) -> locate (bwt_sa) -> chain+filter (mem_chain, mem_chain_flt)
//   -> extend rounds (mem_chain2aln/ksw_extend2) -> dedup (mem_sort_dedup_patch)          == mem_align1_core x2
//   -> rescue rounds (mem_matesw/ksw_align2)                                             == the two rescue loops
//   -> reg2aln (mem_reg2aln/bwa_gen_cigar2/ksw_global2) for every surviving region.
#pragma once
#include <vector>
#include <string>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include "arx_dev.h"
#include "dev_fm.h"
#include "dev_chain.h"
#include "dev_sw.h"
#include "dev_regs.h"

#ifndef ARX_STAT_BWD
#define ARX_STAT_BWD(n, ext) ((void)(ext)) // the host test double can histogram the backward tasks here
#endif

namespace arx {

ARX_DEVI void raise_err(uint32_t *e, uint32_t bit) { ARX_ATOMIC_OR(e, bit); } // error bits of a batch (Pipeline::d_err)

// ---------------------------------------------------------------- kernel functors (item = work unit, slot = scratch slot)
// ---- first two passes of mem_collect_intv as forward / backward tasks (dev_fm.h).  These functors run one item per thread; the
// HIP runtime drives the same lane programs with persistent lanes instead (hip_fm_coop.h).
struct KSeedFwd1 { // first pass, forward halves: the starts of a read chain through bwt_smem1a's return value
	IndexView ix; const uint8_t *bases; const int32_t *base_off, *lens; SeedPools P; Biv *scratch; int list_cap; int32_t *first1;
	int read0; // the group of reads this launch covers starts here (stage_seed)
	ARX_DEV void operator()(int item, int slot) const
	{
		const int r = read0 + item;
		int len = lens[r], head = -1, last = -1;
		if (len > MAX_READ_LEN) { raise_err(P.err, ERR_READ_TOO_LONG); len = 0; }
		if (len >= OPT_MIN_SEED_LEN) {
			const QBytes q{bases + base_off[r]};
			Biv *list = scratch + (size_t)slot * list_cap;
			for (int x = 0;;) {
				while (x < len && q.at(x) > 3) ++x;
				if (x >= len) break;
				FwdLane<QBytes> ln;
				ln.start(ix, len, q, x, 1, list);
				Biv req = Biv();
				int rc = 0;
				while (ln.advance(&req, &rc)) {
					if (rc >= 0) { ln.consume(extend1(ix, req, 0, rc)); continue; }
					const uint32_t *a = aux_addr(ix, rc, req.k); // text mode (dev_fm.h)
					ln.consume_aux(ix, a[0], a[1], a[2], a[3]);
				}
				const int t = seed_export(P, r, x, 1, list, ln.n);
				if (t < 0) break;
				if (last >= 0) P.tasks[last].next = t; else head = t;
				last = t;
				x = ln.ret();
			}
		}
		first1[r] = head;
	}
};

struct KSeedBwd { // the backward sweep of task t0 + item
	IndexView ix; const uint8_t *bases; const int32_t *base_off, *lens; SeedPools P; int t0;
	bool by_entry = false; // the entry-by-entry form of the sweep (what hip_fm_coop.h k_seed_bwd_e + k_bwd_e_final compute): the host test double's check of its claims
	ARX_DEV void entries(const SeedTask &t, int item) const
	{
		const QBytes q{bases + base_off[t.read]};
		Biv *list = P.pool + t.off, *res = list + t.n, *mem = res + t.n;
		if (t.x == 0) { mem[0] = list[0]; P.tasks[t0 + item].nm = 1; return; }
		for (int j = 0; j < t.n; ++j) { // every interval on its own: how far to the left before it holds fewer than min_intv occurrences
			Biv cur = list[j];
			int i = t.x - 1;
			while (i >= 0 && q.at(i) <= 3) {
				if (ix.sa40 && ix.isa40 && t.min_intv == 1 && cur.s == 1) { // one occurrence left: the text decides (bwd_text_tail's comparison)
					const uint64_t p = p40_load(ix.sa40, cur.k);
					int m = 0;
					while (i - m >= 0 && (uint64_t)m < p && q.at(i - m) <= 3 && q.at(i - m) == ref_base(ix, (int64_t)(p - 1 - (uint64_t)m))) ++m;
					if (m) cur.k = p40_load(ix.isa40, p - (uint64_t)m);
					i -= m;
					break;
				}
				const Biv ok = extend1(ix, cur, 1, q.at(i));
				if (ok.s < (uint64_t)t.min_intv) break;
				const uint64_t info = cur.info;
				cur = ok; cur.info = info; --i;
			}
			cur.info = (uint64_t)(uint32_t)cur.info | (uint64_t)(i + 1) << 32;
			res[j] = cur;
		}
		int nm = 0, mls = 0;
		for (int j = 0; j < t.n; ++j) { // longest first: an SMEM iff it starts left of the last one found
			const int start = (int)(res[j].info >> 32);
			if (nm == 0 || start < mls) { mem[nm++] = res[j]; mls = start; }
		}
		P.tasks[t0 + item].nm = nm;
	}
	ARX_DEV void operator()(int item, int) const
	{
		SeedTask t = P.tasks[t0 + item];
		if (t.n == 0) return;
		if (by_entry) { entries(t, item); return; }
		BwdLane<QBytes> ln;
		ln.start(QBytes{bases + base_off[t.read]}, t, P.pool);
		ln.use_text(ix);
		Biv req = Biv();
		int rc = 0;
		int n_ext = 0;
		while (ln.advance(&req, &rc)) { ln.consume(req, extend1(ix, req, 1, rc)); ++n_ext; }
		ARX_STAT_BWD(t.n, n_ext);
		P.tasks[t0 + item].nm = ln.nm;
	}
};

struct KSeedGather1 {
	const uint8_t *bases; const int32_t *base_off; SeedPools P; const int32_t *first1; Biv *intv; int32_t *n_intv, *first2; int read0;
	ARX_DEV void operator()(int item, int) const
	{
		const int r = read0 + item;
		int ovf = 0, f2 = -1;
		n_intv[r] = seed_gather_pass1(P, r, first1[r], bases + base_off[r], intv + (size_t)r * CAP_INTV, CAP_INTV, &ovf, &f2);
		first2[r] = f2;
		if (ovf) raise_err(P.err, ERR_INTV_OVERFLOW);
	}
};

struct KSeedFwd2 { // forward half of re-seeding task t0 + item; its pool slice is reserved once the list length is known
	IndexView ix; const uint8_t *bases; const int32_t *base_off, *lens; SeedPools P; Biv *scratch; int list_cap; int t0;
	ARX_DEV void operator()(int item, int slot) const
	{
		SeedTask t = P.tasks[t0 + item];
		const QBytes q{bases + base_off[t.read]};
		Biv *list = scratch + (size_t)slot * list_cap;
		FwdLane<QBytes> ln;
		ln.start(ix, lens[t.read], q, t.x, t.min_intv, list);
		Biv req = Biv();
		int rc = 0;
		while (ln.advance(&req, &rc)) ln.consume(extend1(ix, req, 0, rc));
		seed_export_into(P, t0 + item, list, ln.n);
	}
};

struct KSeedGather2 {
	SeedPools P; const int32_t *first2; Biv *intv; int32_t *n_intv; int read0;
	ARX_DEV void operator()(int item, int) const
	{
		const int r = read0 + item;
		int ovf = 0;
		n_intv[r] = seed_gather_pass2(P, first2[r], intv + (size_t)r * CAP_INTV, n_intv[r], CAP_INTV, &ovf);
		if (ovf) raise_err(P.err, ERR_INTV_OVERFLOW);
	}
};

struct KSeedStrat { // pass 3 for one read
	IndexView ix; const uint8_t *bases; const int32_t *base_off, *lens; Biv *strat; int32_t *n_strat;
	ARX_DEV void operator()(int r, int) const
	{
		const int len = lens[r];
		int n = 0;
		if (len >= OPT_MIN_SEED_LEN && len <= MAX_READ_LEN) {
			StratLane<QBytes> ln;
			ln.start(len, QBytes{bases + base_off[r]}, strat + (size_t)r * CAP_STRAT);
			Biv req = Biv();
			int rc = 0;
			while (ln.advance(ix, &req, &rc)) { if (rc < 0) ln.consume_tab(ix, ktab_load(ix, req.k)); else ln.consume(extend1(ix, req, 0, rc)); }
			n = ln.n;
		}
		n_strat[r] = n;
	}
};

struct KSeedMerge { // both interval lists of a read, sorted; the number of seed occurrences they expand to (bwamem.c:273-283: at most max_occ rows per interval)
	Biv *intv; int32_t *n_intv; const Biv *strat; const int32_t *n_strat; int32_t *n_occ; uint32_t *err;
	ARX_DEV void operator()(int r, int) const
	{
		Biv *out = intv + (size_t)r * CAP_INTV;
		int ovf = 0, occ = 0;
		const int n = seed_merge(out, n_intv[r], strat + (size_t)r * CAP_STRAT, n_strat[r], CAP_INTV, &ovf);
		for (int i = 0; i < n; ++i) occ += out[i].s > (uint64_t)OPT_MAX_OCC ? OPT_MAX_OCC : (int)out[i].s;
		if (ovf) raise_err(err, ERR_INTV_OVERFLOW);
		n_intv[r] = n; n_occ[r] = occ;
	}
};

// per read: expand its intervals into seed occurrences (bwamem.c:273-283): at most max_occ rows per interval, evenly
// strided; the SA row goes to Seed::rbeg until KLocate replaces it by the reference position
struct KOccFill {
	const Biv *intv; const int32_t *n_intv, *occ_off; Seed *occ_seed;
	ARX_DEV void operator()(int r, int) const
	{
		const Biv *iv = intv + (size_t)r * CAP_INTV;
		int g = occ_off[r];
		for (int i = 0; i < n_intv[r]; ++i) {
			const Biv p = iv[i];
			const int cnt = p.s > (uint64_t)OPT_MAX_OCC ? OPT_MAX_OCC : (int)p.s;
			const uint64_t step = p.s > (uint64_t)OPT_MAX_OCC ? p.s / OPT_MAX_OCC : 1;
			Seed s;
			s.qbeg = (int32_t)(p.info >> 32);
			s.len = (int32_t)((uint32_t)p.info - (uint32_t)(p.info >> 32));
			for (int k = 0; k < cnt; ++k) { s.rbeg = (int64_t)(p.k + (uint64_t)k * step); occ_seed[g++] = s; }
		}
	}
};

// one seed occurrence: sampled-SA walk (bwt_sa, bwt.c:86-96)
struct KLocate {
	IndexView ix; Seed *occ_seed;
	ARX_DEV void operator()(int g, int) const { occ_seed[g].rbeg = (int64_t)sa_lookup(ix, (uint64_t)occ_seed[g].rbeg); }
};
// contig of every occurrence (bns_intv2rid as mem_chain calls it, bwamem.c:281-283; < 0: spans contigs or the strand boundary)
struct KOccRid {
	IndexView ix; const Seed *occ_seed; int32_t *occ_rid;
	ARX_DEV void operator()(int g, int) const { const Seed s = occ_seed[g]; occ_rid[g] = intv2rid(ix, s.rbeg, s.rbeg + s.len); }
};

// Reads with many seed occurrences (reads in high-copy repeats: 0.1 % of a GRCh38-size batch carries 65-800 of them, against a median
// of 8) are 85 % of the thread-per-read kernel's time -- one lane walking a B-tree and chain lists in HBM, several thousand dependent
// round trips -- while the other 99.9 % finish in a quarter of it.  KChain hands such a read to k_chain_heavy (arx_cold.hip): one
// wavefront per read, the read's working set in LDS.
constexpr int CHAIN_HEAVY_MIN = 64, CHAIN_LDS_OCC = 832; // 832 occurrences x 154 B of working set = 128 KB of a CU's 160 KB LDS
struct KChain {
	IndexView ix; const int32_t *lens; const Biv *intv; const int32_t *n_intv, *occ_off; const Seed *occ_seed; const int32_t *occ_rid;
	int32_t *next; Chain *ctmp; BtNode *nodes; int32_t *iscr; Chain *cout; Seed *sout; int32_t *n_chain; uint32_t *err;
	int32_t *heavy_list, *n_heavy; int32_t heavy_min; // null: every read is chained by its own thread
	// reads between the typical few occurrences and the heavy ones (99 % of a GRCh38-size batch has at most 15, 0.6 % has 16-63): listed and
	// chained by a launch of their own, so that a wavefront of 64 typical reads does not wait for the one read with 40 (null / 0: off)
	int32_t *mid_list = nullptr, *n_mid = nullptr; int32_t mid_min = 0, mid_cap = 0;
	ARX_DEV void operator()(int r, int) const
	{
		const int g0 = occ_off[r], n = occ_off[r + 1] - g0;
		if (heavy_list && n >= heavy_min && n <= CHAIN_LDS_OCC) { heavy_list[ARX_ATOMIC_ADD(n_heavy, 1)] = r; return; }
		if (mid_list && n >= mid_min) { const int at = ARX_ATOMIC_ADD(n_mid, 1); if (at < mid_cap) { mid_list[at] = r; return; } } // (a full list: chained here)
		one(r);
	}
	ARX_DEV void one(int r) const
	{
		const int g0 = occ_off[r], n = occ_off[r + 1] - g0;
		const int node0 = g0 / 3 + 4 * r, node1 = occ_off[r + 1] / 3 + 4 * (r + 1);
		int m = chain_and_filter(ix, lens[r], intv + (size_t)r * CAP_INTV, n_intv[r], occ_seed + g0, occ_rid + g0, n, next + g0, ctmp + g0,
		                         nodes + node0, node1 - node0, iscr + 7 * (size_t)g0, cout + g0, sout + g0, g0);
		if (m < 0) { raise_err(err, ERR_POOL_OVERFLOW); m = 0; }
		n_chain[r] = m;
	}
};

struct KChainMid { // the listed reads of KChain, one thread each; the list's length stays on the device
	KChain f;
	ARX_DEV void operator()(int i, int) const { const int n = *f.n_mid < f.mid_cap ? *f.n_mid : f.mid_cap; if (i < n) f.one(f.mid_list[i]); }
};

// per read: set up one state machine per chain (chain gid = chain_off[read] + index) and remember the chain's read
struct KExtInit {
	IndexView ix; const int32_t *lens, *occ_off, *n_chain, *chain_off; const Chain *chains; const Seed *seeds; int32_t *srt; ExtState *state; int32_t *chain_read;
	ARX_DEV void operator()(int r, int) const
	{
		const int g0 = occ_off[r], c0 = chain_off[r];
		for (int ci = 0; ci < n_chain[r]; ++ci) {
			ExtState st;
			ext_init_chain(ix, lens[r], chains[g0 + ci], seeds, srt, st);
			state[c0 + ci] = st;
			chain_read[c0 + ci] = r;
		}
	}
};

struct KExtStep {
	IndexView ix; const int32_t *base_off, *lens, *occ_off, *chain_off, *chain_read; const Chain *chains; const Seed *seeds; int32_t *srt; Reg *regs;
	ExtState *state; const ExtRes *res; ExtTask *tasks; int32_t *n_tasks; int round;
	int task_stride;                              // tasks + c * task_stride: the list of length class c, n_tasks[c] its length
	const int32_t *act_in; int32_t *act_out;      // chains still extending or waiting (null in the first round: all); n_tasks[EXT_CLASSES] counts act_out
	ARX_DEV void operator()(int item, int) const
	{
		const int gid = act_in ? act_in[item] : item;
		ExtState st = state[gid];
		if (st.phase == PH_DONE) return;
		const int r = chain_read[gid], c0 = chain_off[r], g0 = occ_off[r];
		ExtTask t;
		const ExtRes rs = res[gid]; // only read by a chain that queued a DP in the previous round
		const int what = ext_step(ix, gid, base_off[r], lens[r], chains + g0, gid - c0, seeds, srt, regs, state + c0, round, st, rs, t);
		state[gid] = st;
#if defined(__HIP_DEVICE_COMPILE__)
		// One atomic round trip per wavefront: the lanes count themselves per list with ballots, the first lane of every list reserves for
		// all of them (its own class list and the list of chains still active: two atomics in flight together), the others read the base
		// from it.  (One counter after the other -- seven classes since round 3 -- put seven dependent round trips into every wavefront.)
		const int c = what == EXT_TASK ? ext_class(t.qlen) : -1;
		const bool keep = what != EXT_FINISHED;
		const unsigned long long m_keep = __ballot(keep), lt = (1ull << __lane_id()) - 1ull;
		unsigned long long m_c = 0;
#pragma unroll
		for (int k = 0; k < EXT_CLASSES; ++k) { const unsigned long long mk = __ballot(c == k); m_c = c == k ? mk : m_c; }
		const int lead_c = c >= 0 ? __builtin_ctzll(m_c) : (int)__lane_id(), lead_k = keep ? __builtin_ctzll(m_keep) : (int)__lane_id();
		int base_c = 0, base_k = 0;
		if (c >= 0 && (int)__lane_id() == lead_c) base_c = atomicAdd(n_tasks + c, __popcll(m_c));
		if (keep && (int)__lane_id() == lead_k) base_k = atomicAdd(n_tasks + EXT_CLASSES, __popcll(m_keep));
		base_c = __shfl(base_c, lead_c); base_k = __shfl(base_k, lead_k);
		if (c >= 0) tasks[(size_t)c * task_stride + base_c + __popcll(m_c & lt)] = t;
		if (keep) act_out[base_k + __popcll(m_keep & lt)] = gid;
#else
		if (what == EXT_TASK) {
			const int c = ext_class(t.qlen);
			tasks[(size_t)c * task_stride + claim(n_tasks + c)] = t;
		}
		if (what != EXT_FINISHED) act_out[claim(n_tasks + EXT_CLASSES)] = gid;
#endif
	}
	static ARX_DEVI int claim(int32_t *ctr) { return ARX_ATOMIC_INC(ctr); }
};

// per read: its chains' regions in chain order -> the read's region list (the order mem_chain2aln appends them in)
struct KExtGather {
	const int32_t *occ_off, *n_chain, *chain_off; const Chain *chains; const ExtState *state; const Reg *pool; Reg *regs; int32_t *n_ext;
	ARX_DEV void operator()(int r, int) const
	{
		const int g0 = occ_off[r], c0 = chain_off[r];
		int n = 0;
		for (int ci = 0; ci < n_chain[r]; ++ci) {
			const Reg *src = pool + chains[g0 + ci].seed_off;
			for (int i = 0; i < state[c0 + ci].n_regs; ++i) regs[g0 + n++] = src[i];
		}
		n_ext[r] = n;
	}
};

struct KExtend {
	IndexView ix; const uint8_t *bases; const ExtTask *tasks; ExtRes *res;
	ARX_DEV void operator()(int i, int, uint32_t *row, int stride) const
	{
		const ExtTask t = tasks[i];
		res[t.owner] = ext2_task(ix, bases, t, row, stride);
	}
};

// Reads with long region lists (32+ regions: reads in high-copy repeats) leave the thread-per-read kernel for k_dedup_heavy
// (arx_cold.hip): one wavefront per read, the list in LDS (dev_regs_wave.h: w_sort_dedup).
constexpr int DEDUP_HEAVY_MIN = 32, DEDUP_LDS_REGS = 256;
struct KDedup {
	IndexView ix; const uint8_t *bases; const int32_t *base_off, *lens, *occ_off; const int32_t *n_ext; Reg *regs, *tmp; int32_t *idx;
	int32_t *eh; int eh_words; int32_t *n_core;
	int32_t *clean; // what the rescue stage may assume about the list (matesw_apply): 2 = went through the pass with >= 2 regions and nothing was merged,
	                // so it is a fixed point of the pass mem_matesw repeats; 1 = fewer than two regions; 0 = a merge happened, nothing is known
	int32_t *heavy_list, *n_heavy; int32_t heavy_min; // null: every read by its own thread
	ARX_DEV void operator()(int r, int slot) const
	{
		if (heavy_list && n_ext[r] >= heavy_min && n_ext[r] <= DEDUP_LDS_REGS) { heavy_list[ARX_ATOMIC_ADD(n_heavy, 1)] = r; return; }
		one_thread(r, eh + (size_t)slot * eh_words);
	}
	ARX_DEV void one_thread(int r, int32_t *eh_slot) const
	{
		const int g0 = occ_off[r];
		int n = n_ext[r];
		int32_t patched = 0;
		clean[r] = n >= 2 ? 2 : 1;
		n = sort_dedup_patch(ix, bases + base_off[r], n, regs + g0, tmp + g0, idx + g0, eh_slot, &patched);
		if (patched) clean[r] = 0;
		for (int i = 0; i < n; ++i) { Reg &p = regs[g0 + i]; if (p.rid >= 0 && ix.ann_alt[p.rid]) p.is_alt = 1; }
		n_core[r] = n;
	}
};

// capacity of each read's final region list: core regions + one rescue per eligible anchor of the mate (<= 50)
struct KPairCap {
	const int32_t *n_core; int32_t *cap;
	ARX_DEV void operator()(int p, int) const
	{
		int n0 = n_core[2 * p], n1 = n_core[2 * p + 1];
		int c0 = n0 + (n1 < MAX_RESCUE ? n1 : MAX_RESCUE);
		int c1 = n1 + (c0 < MAX_RESCUE ? c0 : MAX_RESCUE);
		cap[2 * p] = c0 + 1; cap[2 * p + 1] = c1 + 1;
	}
};

// Pairs whose two region lists are long (reads in high-copy repeats: 100-190 regions each) replay their rescue loops with the lists in
// LDS (hip_rt.h: k_rescue_heavy): the replay is one thread walking and shifting 88-byte records, a chain of dependent memory round trips
// that costs 10-14 ms per round from HBM / L2 and a fraction of that from LDS.  heavy[p] marks such a pair (both lists with their spare
// capacity must fit RESCUE_LDS_REGS records); the thread-per-pair kernel skips them.
constexpr int RESCUE_HEAVY_MIN = 48, RESCUE_LDS_REGS = 680; // 680 x 88 B = 58.4 KB; with the 17 KB of the wave's sort scratch two workgroups share a CU's 160 KB
struct KPairInit {
	const int32_t *occ_off, *n_core, *preg_off; const Reg *regs; Reg *pregs; int32_t *n_regs; ResState *state; const int32_t *core_clean;
	const int32_t *cap; uint8_t *heavy; int32_t *heavy_list, *n_heavy; // null: no heavy path
	int32_t heavy_min; // regions of both reads together from which a pair is heavy (RESCUE_HEAVY_MIN; ARX_RESCUE_HEAVY_MIN for tests)
	ARX_DEV void operator()(int p, int) const
	{
		if (heavy) {
			const bool hv = n_core[2 * p] + n_core[2 * p + 1] >= heavy_min && cap[2 * p] + cap[2 * p + 1] <= RESCUE_LDS_REGS;
			heavy[p] = hv ? 1 : 0;
			if (hv) heavy_list[ARX_ATOMIC_ADD(n_heavy, 1)] = p;
		}
		ResState st = ResState();
		for (int e = 0; e < 2; ++e) {
			const int r = 2 * p + e, n = n_core[r];
			st.clean[e] = core_clean[r]; // the list mem_align1_core left is usually already a fixed point of mem_matesw's pass (KDedup)
			int best = 0;
			for (int i = 0; i < n; ++i) { Reg x = regs[occ_off[r] + i]; pregs[preg_off[r] + i] = x; if (x.score > best) best = x.score; }
			n_regs[r] = n; st.best[e] = best;
		}
		st.e = 1; st.i = 0; st.num = 0; st.n_snap = n_core[2 * p + 1]; st.phase = 3; // first loop: anchors = read 2's hits
		state[p] = st;
	}
};

struct KRescueStep {
	IndexView ix; const int32_t *lens, *preg_off; Reg *pregs, *ptmp; int32_t *pidx, *n_regs; ResState *state; const U8Res *res; SwTask *tasks; int32_t *n_tasks, *n_slots;
	int32_t no_ahead;
	int32_t single_base; // result slots [0, single_base): SWs queued ahead; single_base + pair: the pair's single SW
	const uint8_t *heavy; // pairs the LDS kernel takes (null: none)
	ARX_DEV void operator()(int p, int) const
	{
		if (heavy && heavy[p]) return;
		ResState st = state[p];
		if (st.phase == 2) return;
		Reg *rg[2] = { pregs + preg_off[2 * p], pregs + preg_off[2 * p + 1] };
		Reg *tm[2] = { ptmp + preg_off[2 * p], ptmp + preg_off[2 * p + 1] };
		int *ix2[2] = { pidx + preg_off[2 * p], pidx + preg_off[2 * p + 1] };
		int *nr[2] = { n_regs + 2 * p, n_regs + 2 * p + 1 };
		SwEmit em; em.tasks = tasks; em.n_tasks = n_tasks; em.n_slots = n_slots; em.single_slot = single_base + p; em.no_ahead = no_ahead;
		rescue_step(ix, p, lens + 2 * p, rg, nr, tm, ix2, st, res, em);
		state[p] = st;
	}
};

struct KSwU8 {
	IndexView ix; const uint8_t *bases; const int32_t *base_off, *lens; const SwTask *tasks; U8Res *res; uint8_t *scratch; int q_cap, t_cap;
	ARX_DEV void operator()(int i, int slot, uint32_t *row, int stride) const
	{
		const SwTask t = tasks[i];
		const int r = 2 * t.pair + t.o, l_ms = lens[r], tlen = (int)(t.re - t.rb);
		uint8_t *qbuf = scratch + (size_t)slot * (q_cap + 2 * t_cap), *tbuf = qbuf + q_cap, *rowmax = tbuf + t_cap;
		const uint8_t *ms = bases + base_off[r];
		for (int k = 0; k < l_ms; ++k) { int b = ms[k]; qbuf[l_ms - 1 - k] = b < 4 ? 3 - b : 4; } // reverse complement of the mate (bwamem_pair.c:134-137)
		for (int k = 0; k < tlen; ++k) tbuf[k] = (uint8_t)ref_base(ix, t.rb + k);
		res[t.slot] = u8_align(qbuf, l_ms, tbuf, tlen, KSW_XSUBO | KSW_XSTART | (l_ms * OPT_A < 250 ? KSW_XBYTE : 0) | (OPT_MIN_SEED_LEN * OPT_A), row, stride, rowmax); // bwamem_pair.c:150
		ARX_SW_FILTER_CHECK(sw_prefilter_serial(qbuf, l_ms, tbuf, tlen), res[t.slot].score);
	}
};

struct KReg2Aln {
	IndexView ix; const uint8_t *bases; const int32_t *base_off, *lens, *preg_off, *n_regs; int n_reads; const Reg *pregs;
	Aln *alns; uint32_t *cig; int cig_w; int32_t *eh; int eh_words; uint8_t *z; int z_cap; uint32_t *err;
	int32_t *nw_list, *nw_count; int mode; // mode 0: every region slot, gap-free ones finished inline, the rest queued; mode 1: the queued ones
	int32_t *nw_need, *big_list; // per queued region: 64-byte units of traceback matrix; regions too long for the 16-lane kernel (nw_count[1] of them)
	int32_t *class_list; int class_stride; // queue positions by the lane tiling of the first band (nw_count[2 + c] of class c): a wavefront's four groups then run the same code
	ARX_DEV void operator()(int item, int slot) const
	{
		const int g = mode == 2 ? big_list[item] : (mode ? nw_list[item] : item);
		int lo = 0, hi = n_reads;
		while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (preg_off[mid] <= g) lo = mid; else hi = mid; }
		const int r = lo, j = g - preg_off[r];
		if (j >= n_regs[r]) return;
		const Reg ar = pregs[g];
		Aln a = Aln();
		a.cigar_off = g * cig_w;
		const int lq = lens[r];
		if (mode == 0) {
			// both infer_bw() calls give 0 <=> equal lengths and fewer than 12 score units lost: the CIGAR is one M run (bwa.c:141-149)
			const int l1 = ar.qe - ar.qb, l2 = (int)(ar.re - ar.rb);
			const bool gapfree = l1 == l2 && l1 * OPT_A - ar.truesc < ((OPT_O_DEL + OPT_E_DEL - OPT_A) << 1);
			if (!gapfree) {
				if (l1 > NW_Q_CAP || l2 > NW_T_CAP) { big_list[KExtStep::claim(nw_count + 1)] = g; return; }
				const int k = KExtStep::claim(nw_count);
				nw_list[k] = g; nw_need[k] = (int32_t)((reg2aln_z_bound(ar) + 63) >> 6);
				int w0 = reg2aln_w0(ar); w0 = w0 < OPT_W << 2 ? w0 : OPT_W << 2;
				const int c = reg2aln_band_class(ar, w0);
				int at = 0;
				for (int q = 0; q < NW_CLASSES; ++q) if (c == q) at = KExtStep::claim(nw_count + 2 + q);
				class_list[(size_t)c * class_stride + at] = k;
				return;
			}
			if (!reg2aln(ix, lq, bases + base_off[r], ar, (int32_t *)0, (uint8_t *)0, cig + (size_t)g * cig_w, cig_w, a)) raise_err(err, ERR_CIGAR_OVERFLOW);
			alns[g] = a;
			return;
		}
		// worst-case traceback matrix for this region: n_col <= l_query, rows = re - rb
		if ((int64_t)(ar.qe - ar.qb) * (ar.re - ar.rb) > z_cap) { raise_err(err, ERR_POOL_OVERFLOW); a.n_cigar = 0; a.rid = -1; alns[g] = a; return; }
		if (!reg2aln(ix, lq, bases + base_off[r], ar, eh + (size_t)slot * eh_words, z + (size_t)slot * z_cap, cig + (size_t)g * cig_w, cig_w, a))
			raise_err(err, ERR_CIGAR_OVERFLOW);
		alns[g] = a;
	}
};

// results to dense arrays (reads in input order, a read's regions in list order): what arx_batch_fetch hands out
struct KCompactCount {
	const int32_t *preg_off, *n_regs, *c_off; const Aln *alns; int32_t *cig_len;
	ARX_DEV void operator()(int r, int) const { for (int j = 0; j < n_regs[r]; ++j) cig_len[c_off[r] + j] = alns[preg_off[r] + j].n_cigar; }
};
struct KCompact {
	const int32_t *preg_off, *n_regs, *c_off, *cig_off; const Reg *pregs; const Aln *alns; const uint32_t *cig; int cig_w; Reg *o_regs; Aln *o_alns; uint32_t *o_cig;
	ARX_DEV void operator()(int r, int) const
	{
		for (int j = 0; j < n_regs[r]; ++j) {
			const int g = preg_off[r] + j, d = c_off[r] + j;
			Aln a = alns[g];
			const uint32_t *c = cig + (size_t)g * cig_w;
			a.cigar_off = cig_off[d];
			for (int k = 0; k < a.n_cigar; ++k) o_cig[a.cigar_off + k] = c[k];
			o_regs[d] = pregs[g]; o_alns[d] = a;
		}
	}
};

// ---------------------------------------------------------------- batch containers
struct BatchResult { // statistics of the run (the results themselves stay on the device until arx_batch_fetch)
	int ext_rounds = 0, rescue_rounds = 0;
	int64_t n_occ = 0, n_ext_tasks = 0, n_sw_tasks = 0;
};

template <class RT> class Pipeline {
public:
	RT &rt;
	IndexView ix;
	bool trace = getenv("ARX_TRACE") != nullptr; // per-round progress on stderr
	int seed_group_reads = getenv("ARX_SEED_GROUP") ? atoi(getenv("ARX_SEED_GROUP")) : 0; // reads per pass through the first two seeding passes (0: the whole batch at once; groups shrink the interval pool from 12 KB to 12 KB x group / batch per read at the price of under-filled forward launches: 0 / 360 k / 180 k / 90 k reads -> 7.0 / 9.5 / 11.4 / 14.1 ms of seed_fwd per 667 k-read batch, seed_bwd unchanged)
	int seed_tasks_per_read = getenv("ARX_SEED_TASKS") ? atoi(getenv("ARX_SEED_TASKS")) : 12;   // seeding tasks per read (all three passes), same rule
	int seed_pool_per_read = getenv("ARX_SEED_POOL") ? atoi(getenv("ARX_SEED_POOL")) : 384; // interval-pool entries per read (3 per forward-list entry); an overflow is reported, never silent
	explicit Pipeline(RT &rt_, const IndexView &ix_) : rt(rt_), ix(ix_) {}

	// device-resident input of one batch; the buffers are kept by the handle and reused by arx_batch_reset (cap_*: what they hold)
	struct DeviceBatch { uint8_t *bases = 0; int32_t *base_off = 0, *lens = 0; int n_reads = 0; int64_t n_bases = 0; int max_len = 0; int64_t cap_bases = 0; int cap_reads = 0; };

	// Reads go to HBM through the runtime's staging (pinned host memory on the GPU, one async copy per array on the batch's stream).
	// An existing batch keeps its buffers when the new reads fit: a steady-state caller allocates nothing (the reference recycles
	// its buffers per work unit the same way, aligner.go:234 ReturnBuffer, gobwa.go:107-126 Arena).
	void upload_into(DeviceBatch &b, const uint8_t *bases, const int32_t *lens, int n_reads)
	{
		int64_t tot = 0; int mx = 0;
		for (int i = 0; i < n_reads; ++i) { tot += lens[i]; if (lens[i] > mx) mx = lens[i]; }
		if (tot + 16 > b.cap_bases) { rt.pfree(b.bases); b.cap_bases = tot + tot / 8 + 64; b.bases = rt.template palloc<uint8_t>((size_t)b.cap_bases); }
		if (n_reads + 1 > b.cap_reads) {
			rt.pfree(b.base_off); rt.pfree(b.lens);
			b.cap_reads = n_reads + n_reads / 8 + 8;
			b.base_off = rt.template palloc<int32_t>((size_t)b.cap_reads); b.lens = rt.template palloc<int32_t>((size_t)b.cap_reads);
		}
		b.n_reads = n_reads; b.n_bases = tot; b.max_len = mx;
		const bool direct = tot > 0 && RT::host_pinned(bases); // page-locked by the caller (arx_host_register): the bases go as they lie, no staging copy
		uint8_t *st = (uint8_t *)rt.stage((direct ? 0 : (size_t)tot) + 8 * ((size_t)n_reads + 2) + 8);
		int32_t *st_off = (int32_t *)(st + (direct ? 0 : (((size_t)tot + 7) & ~(size_t)7))), *st_len = st_off + n_reads + 1;
		if (!direct) memcpy(st, bases, (size_t)tot);
		int64_t run = 0;
		for (int i = 0; i < n_reads; ++i) { st_off[i] = (int32_t)run; run += lens[i]; }
		st_off[n_reads] = (int32_t)run;
		memcpy(st_len, lens, sizeof(int32_t) * (size_t)n_reads);
		if (direct) rt.h2d_pinned(b.bases, bases, (size_t)tot); else rt.h2d_staged(b.bases, st, (size_t)tot);
		rt.h2d_staged(b.base_off, st_off, sizeof(int32_t) * ((size_t)n_reads + 1));
		rt.h2d_staged(b.lens, st_len, sizeof(int32_t) * (size_t)n_reads);
	}
	// The same from arrays that are in device memory already (a batch received from another GPU over RCCL: arachne_amd/shard.py): device-to-device
	// copies on the batch's stream, the base offsets by a scan on the device; the read lengths come back to the host once (arx_batch_rfa's
	// host-libm guard needs them).  n_bases / max_len: what the sender's header says; checked against the scan.  The caller's buffers must be
	// complete when this is called (its own stream synchronised) and may be reused when it returns.
	bool upload_from_device(DeviceBatch &b, const uint8_t *d_bases, const int32_t *d_lens, int n_reads, int64_t n_bases, int max_len, std::vector<int32_t> &lens_host)
	{
		if (n_bases + 16 > b.cap_bases) { rt.pfree(b.bases); b.cap_bases = n_bases + n_bases / 8 + 64; b.bases = rt.template palloc<uint8_t>((size_t)b.cap_bases); }
		if (n_reads + 1 > b.cap_reads) {
			rt.pfree(b.base_off); rt.pfree(b.lens);
			b.cap_reads = n_reads + n_reads / 8 + 8;
			b.base_off = rt.template palloc<int32_t>((size_t)b.cap_reads); b.lens = rt.template palloc<int32_t>((size_t)b.cap_reads);
		}
		b.n_reads = n_reads; b.n_bases = n_bases; b.max_len = max_len;
		rt.d2d(b.bases, d_bases, (size_t)n_bases);
		rt.d2d(b.lens, d_lens, sizeof(int32_t) * (size_t)n_reads);
		const int64_t tot = rt.exclusive_scan(b.lens, b.base_off, n_reads);
		lens_host.resize((size_t)n_reads);
		rt.d2h(lens_host.data(), b.lens, sizeof(int32_t) * (size_t)n_reads);
		int mx = 0;
		for (int i = 0; i < n_reads; ++i) { if (lens_host[i] < 0 || lens_host[i] > MAX_READ_LEN) return false; if (lens_host[i] > mx) mx = lens_host[i]; }
		b.max_len = mx;
		return tot == n_bases;
	}
	DeviceBatch upload(const uint8_t *bases, const int32_t *lens, int n_reads)
	{
		DeviceBatch b;
		upload_into(b, bases, lens, n_reads);
		return b;
	}
	void release(DeviceBatch &b) { rt.pfree(b.bases); rt.pfree(b.base_off); rt.pfree(b.lens); b = DeviceBatch(); }

	// everything that stays on the device between the stages of one batch
	struct Work {
		Biv *intv = 0, *smem_scr = 0; int32_t *n_intv = 0, *n_occ = 0, *occ_off = 0; Seed *occ_seed = 0; int32_t *occ_rid = 0, *core_clean = 0;
		int32_t *next = 0, *iscr = 0, *n_chain = 0, *srt = 0, *idx = 0, *n_core = 0; Chain *ctmp = 0, *cout = 0; BtNode *nodes = 0; Seed *sout = 0;
		Reg *regs = 0, *rtmp = 0; ExtState *est = 0; ExtTask *etask = 0; ExtRes *eres = 0; int32_t *counter = 0; uint32_t *err = 0;
		int32_t *eh = 0; int32_t *cap = 0, *preg_off = 0, *n_regs = 0, *pidx = 0; Reg *pregs = 0, *ptmp = 0; ResState *rst = 0; SwTask *stask = 0; U8Res *sres = 0;
		uint8_t *sw_scr = 0, *z = 0; Aln *alns = 0; uint32_t *cig = 0; int32_t *nw_list = 0;
		int32_t *c_reg_off = 0; Reg *c_regs = 0; Aln *c_alns = 0; uint32_t *c_cig = 0; int64_t c_n_regs = 0, c_n_cig = 0; // dense results
		int64_t T = 0, P = 0; int cig_w = 0;
	};

	void free_work(Work &w)
	{
		void *ptrs[] = { w.intv, w.smem_scr, w.n_intv, w.n_occ, w.occ_off, w.occ_seed, w.occ_rid, w.next, w.iscr, w.n_chain, w.srt, w.idx, w.n_core, w.ctmp, w.cout,
		                 w.nodes, w.sout, w.regs, w.rtmp, w.est, w.etask, w.eres, w.counter, w.err, w.eh, w.cap, w.preg_off, w.n_regs, w.pidx, w.pregs,
		                 w.ptmp, w.rst, w.stask, w.sres, w.sw_scr, w.z, w.alns, w.cig, w.nw_list };
		for (void *p : ptrs) if (p) rt.free(p);
		rt.arena_reset();
		w = Work();
	}

	uint32_t read_err(Work &w) { uint32_t e = 0; rt.d2h(&e, w.err, 4); return e; }
	int32_t read_counter(Work &w) { int32_t c = 0; rt.d2h(&c, w.counter, 4); return c; }

	// ---- stage 1+2: seeding and locate.  Leaves intervals and located seeds on the device.
	int stage_seed(const DeviceBatch &b, Work &w)
	{
		const int R = b.n_reads, slots = rt.max_seed_slots() > rt.max_slots() ? rt.max_seed_slots() : rt.max_slots(), list_cap = 2 * (b.max_len + 2); // two forward lists per resident lane (hip_fm_coop.h: FwdProg1)
		rt.set_seed_read_len(b.max_len);
		w.err = rt.template alloc<uint32_t>(4); rt.memset0(w.err, 16);
		w.counter = rt.template alloc<int32_t>(4);
		w.intv = rt.template alloc<Biv>((size_t)R * CAP_INTV);
		w.n_intv = rt.template alloc<int32_t>(R + 1); w.n_occ = rt.template alloc<int32_t>(R + 1); w.occ_off = rt.template alloc<int32_t>(R + 2);
		// Everything below this mark lives for the seeding passes only -- the interval pool (12 KB per read), the task array, the forward
		// lists, the packed reads, the third pass's intervals: ~28 GB of a 1 M-pair batch's ~65 GB -- and is handed back to the arena when
		// the passes are through (the later stages of the same batch bump-allocate over it; stream order makes that safe), so that three
		// 1 M-pair batches in flight fit beside the index and the k-mer table (round 3).
		const auto seed_mark = rt.arena_mark();
		rt.seed_prepare(b.bases, b.base_off, b.lens, R);
		w.smem_scr = rt.template alloc<Biv>((size_t)slots * list_cap);
		Biv *strat = rt.template alloc<Biv>((size_t)R * CAP_STRAT);
		int32_t *n_strat = rt.template alloc<int32_t>(R + 1);
		// first two passes: forward chains -> backward tasks -> gather + re-seeding tasks -> their forward and backward halves -> gather,
		// optionally one GROUP of reads after the other through the same (group-sized) interval pool and task array (ARX_SEED_GROUP).
		// Tried in round 2 because scattered 64-byte reads run at 52 G blocks/s while a kernel's footprint stays within ~3.5 GiB and at
		// 28-34 G/s beyond (tools/calib_random.hip: the reach of the address translation caches) and the GRCh38 Occ table alone is 2.9 GiB;
		// measured: the pool's footprint is NOT what holds the backward sweeps back (no change), so the default is one group.
		const int GR = seed_group_reads > 0 ? seed_group_reads : R;
		const int Rg_max = GR < R ? GR : R;
		SeedPools P;
		// per read, on average: 12 tasks and seed_pool_per_read pool entries for reads of up to 150 bases, in proportion for longer ones
		// (a 255-base read that matches nowhere yields a first-pass task every ~12 bases on a small genome); an overflow is reported
		const int64_t len_scale = b.max_len > 150 ? (b.max_len + 149) / 150 : 1, tasks_per_read = (int64_t)seed_tasks_per_read * len_scale;
		P.pool_cap = (int64_t)Rg_max * seed_pool_per_read * len_scale; P.task_cap = (int32_t)(Rg_max * tasks_per_read < 0x7fffffff ? Rg_max * tasks_per_read : 0x7fffffff);
		P.pool = rt.template alloc<Biv>((size_t)P.pool_cap + 1); P.tasks = rt.template alloc<SeedTask>((size_t)P.task_cap + 1);
		P.cursors = rt.template alloc<int32_t>(2); P.err = w.err;
		int32_t *first1 = rt.template alloc<int32_t>(R + 1), *first2 = rt.template alloc<int32_t>(R + 1);
		for (int g0 = 0; g0 < R; g0 += GR) {
			const int Rg = R - g0 < GR ? R - g0 : GR;
			int32_t cur[2];
			rt.memset0(P.cursors, 8);
			KSeedFwd1 kf{ix, b.bases, b.base_off, b.lens, P, w.smem_scr, list_cap, first1, g0};
			rt.run_seed_fwd1("seed_fwd", Rg, kf, w.counter);
			rt.d2h(cur, P.cursors, 8);
			const int n1 = cur[1] < P.task_cap ? cur[1] : P.task_cap;
			if (getenv("ARX_SEED_DUMP")) { // diagnostics: the first-pass tasks of the first reads and their forward lists
				for (int r = 0; r < 3 && r < Rg; ++r) {
					int32_t t = 0;
					rt.d2h(&t, first1 + g0 + r, 4);
					while (t >= 0 && t < n1) {
						SeedTask k; rt.d2h(&k, P.tasks + t, sizeof k);
						fprintf(stderr, "[arx seed dump] read %d task %d x %d n %d:", r, t, k.x, k.n);
						for (int e = 0; e < k.n && e < 24; ++e) { Biv v; rt.d2h(&v, P.pool + k.off + e, sizeof v); fprintf(stderr, " (s %llu end %d)", (unsigned long long)v.s, (int)(uint32_t)v.info); }
						fprintf(stderr, "\n");
						t = k.next;
					}
				}
			}
			KSeedBwd kb{ix, b.bases, b.base_off, b.lens, P, 0};
			kb.by_entry = getenv("ARX_SEED_BWD_ENTRY") != nullptr; // (only the one-thread form looks at it: the host test double, ARX_SW_SIMPLE)
			rt.run_seed_bwd("seed_bwd", n1, kb, w.counter);
			KSeedGather1 kg1{b.bases, b.base_off, P, first1, w.intv, w.n_intv, first2, g0};
			rt.launch_wide("seed_gather", Rg, kg1);
			rt.d2h(cur, P.cursors, 8);
			const int n2 = cur[1] < P.task_cap ? cur[1] : P.task_cap;
			KSeedFwd2 kf2{ix, b.bases, b.base_off, b.lens, P, w.smem_scr, list_cap, n1};
			rt.run_seed_fwd2("seed_fwd", n2 - n1, kf2, w.counter);
			kb.t0 = n1;
			rt.run_seed_bwd("seed_bwd", n2 - n1, kb, w.counter);
			KSeedGather2 kg2{P, first2, w.intv, w.n_intv, g0};
			rt.launch_wide("seed_gather", Rg, kg2);
		}
		KSeedStrat k3{ix, b.bases, b.base_off, b.lens, strat, n_strat};
		rt.run_seed_strat("seed_strat", R, k3, w.counter);
		KSeedMerge km{w.intv, w.n_intv, strat, n_strat, w.n_occ, w.err};
		rt.launch_wide("seed_merge", R, km);
		int64_t total = rt.exclusive_scan(w.n_occ, w.occ_off, R); // (waits for the stream: the seeding kernels are through)
		rt.arena_rewind(seed_mark); // HipRT: the seeding passes' memory goes back to the arena (w.smem_scr dangles from here on: nothing reads it); the test double keeps it
		if (total >= (int64_t)1 << 30) return -2; // keep 32-bit pool indices; the caller splits the batch
		w.T = total;
		w.occ_seed = rt.template alloc<Seed>(w.T + 1); w.occ_rid = rt.template alloc<int32_t>(w.T + 1);
		if (w.T) {
			KOccFill kf{w.intv, w.n_intv, w.occ_off, w.occ_seed};
			rt.launch_wide("occ_fill", R, kf);
			KLocate kl{ix, w.occ_seed};
			rt.run_locate("locate", (int)w.T, kl, w.counter);
			KOccRid kr{ix, w.occ_seed, w.occ_rid};
			rt.launch_wide("occ_rid", (int)w.T, kr);
		}
		return 0;
	}

	// ---- stage 3: chaining and chain filtering
	void stage_chain(const DeviceBatch &b, Work &w)
	{
		const int R = b.n_reads; const size_t T = (size_t)w.T + 1;
		w.next = rt.template alloc<int32_t>(T); w.ctmp = rt.template alloc<Chain>(T); w.cout = rt.template alloc<Chain>(T);
		w.nodes = rt.template alloc<BtNode>(T / 3 + 4 * (size_t)R + 8); w.iscr = rt.template alloc<int32_t>(7 * T + 8); w.sout = rt.template alloc<Seed>(T);
		w.n_chain = rt.template alloc<int32_t>(R + 1);
		KChain k{ix, b.lens, w.intv, w.n_intv, w.occ_off, w.occ_seed, w.occ_rid, w.next, w.ctmp, w.nodes, w.iscr, w.cout, w.sout, w.n_chain, w.err, nullptr, nullptr,
		         getenv("ARX_CHAIN_HEAVY_MIN") ? atoi(getenv("ARX_CHAIN_HEAVY_MIN")) : CHAIN_HEAVY_MIN};
		if (rt.chain_heavy_ok()) { k.heavy_list = rt.template alloc<int32_t>(R + 4); k.n_heavy = k.heavy_list + R; rt.memset0(k.n_heavy, 16); }
		const int mid_min = getenv("ARX_CHAIN_MID_MIN") ? atoi(getenv("ARX_CHAIN_MID_MIN")) : 16; // 0: no launch of their own for the reads in between
		const int mid_cap = R / 4 + 64; // the list holds a quarter of the reads; a read beyond that is chained where it is found
		if (mid_min > 0 && k.heavy_list) { k.mid_list = rt.template alloc<int32_t>((size_t)mid_cap + 4); k.n_mid = k.mid_list + mid_cap; k.mid_min = mid_min; k.mid_cap = mid_cap; rt.memset0(k.n_mid, 16); }
		rt.launch_wide("chain", R, k);
		if (k.mid_list) { KChainMid km{k}; rt.launch_wide("chain", mid_cap, km); } // (the list's length is not on the host: threads beyond it return at once)
		if (k.heavy_list) rt.run_chain_heavy("chain_heavy", R, k); // the list's length stays on the device: no host round trip
	}

	// ---- stage 4: extension rounds, then de-duplication -> core regions of every read
	void stage_extend(const DeviceBatch &b, Work &w, BatchResult &out)
	{
		const int R = b.n_reads; const size_t T = (size_t)w.T + 1; const int slots = rt.max_slots();
		w.srt = rt.template alloc<int32_t>(T); w.regs = rt.template alloc<Reg>(T); w.rtmp = rt.template alloc<Reg>(T); w.idx = rt.template alloc<int32_t>(T);
		w.n_core = rt.template alloc<int32_t>(R + 1); w.core_clean = rt.template alloc<int32_t>(R + 1);
		const int eh_words = 2 * (b.max_len + 2);
		w.eh = rt.template alloc<int32_t>((size_t)slots * eh_words);
		// one state machine per chain (dev_regs.h): chain gids by a scan of the per-read chain counts
		int32_t *chain_off = rt.template alloc<int32_t>(R + 2);
		const int NCH = (int)rt.exclusive_scan(w.n_chain, chain_off, R);
		w.est = rt.template alloc<ExtState>((size_t)NCH + 1); w.eres = rt.template alloc<ExtRes>((size_t)NCH + 1);
		w.etask = rt.template alloc<ExtTask>((size_t)EXT_CLASSES * (NCH + 1));
		int32_t *chain_read = rt.template alloc<int32_t>((size_t)NCH + 1);
		int32_t *act[2] = { rt.template alloc<int32_t>((size_t)NCH + 1), rt.template alloc<int32_t>((size_t)NCH + 1) };
		int32_t *ecnt = rt.template alloc<int32_t>(EXT_CLASSES + 1);
		int32_t *n_ext = rt.template alloc<int32_t>(R + 1);
		Reg *pool = w.rtmp; // the chains' regions while they are extended; gathered into w.regs in chain order afterwards
		KExtInit ki{ix, b.lens, w.occ_off, w.n_chain, chain_off, w.cout, w.sout, w.srt, w.est, chain_read};
		rt.launch_wide("ext_init", R, ki);
		int n_act = NCH;
		for (int round = 2; n_act > 0; ++round) {
			rt.memset0(ecnt, 4 * (EXT_CLASSES + 1));
			KExtStep ks{ix, b.base_off, b.lens, w.occ_off, chain_off, chain_read, w.cout, w.sout, w.srt, pool, w.est, w.eres, w.etask, ecnt, round,
			            NCH + 1, round == 2 ? nullptr : act[round & 1], act[(round + 1) & 1]};
			rt.launch_wide("ext_step", n_act, ks);
			int32_t cnt[EXT_CLASSES + 1];
			rt.d2h(cnt, ecnt, 4 * (EXT_CLASSES + 1));
			n_act = cnt[EXT_CLASSES];
			int nt = 0;
			for (int c = 0; c < EXT_CLASSES; ++c) nt += cnt[c];
			if (trace) { fprintf(stderr, "[arx] ext round %d: %d chains active, %d DPs\n", round, n_act, nt); fflush(stderr); }
			if (round > w.T + R + 8) { uint32_t e = ERR_INTERNAL; rt.h2d(w.err, &e, 4); break; } // cannot happen: every round retires a DP or a chain
			if (nt == 0) continue;
			out.n_ext_tasks += nt; ++out.ext_rounds;
			KExtend ke{ix, b.bases, w.etask, w.eres};
			rt.run_extend("extend", cnt, NCH + 1, ke);
		}
		KExtGather kg{w.occ_off, w.n_chain, chain_off, w.cout, w.est, pool, w.regs, n_ext};
		rt.launch_wide("ext_gather", R, kg);
		if (trace) { fprintf(stderr, "[arx] dedup\n"); fflush(stderr); }
		KDedup kd{ix, b.bases, b.base_off, b.lens, w.occ_off, n_ext, w.regs, w.rtmp, w.idx, w.eh, eh_words, w.n_core, w.core_clean, nullptr, nullptr,
		          getenv("ARX_DEDUP_HEAVY_MIN") ? atoi(getenv("ARX_DEDUP_HEAVY_MIN")) : DEDUP_HEAVY_MIN};
		if (rt.dedup_heavy_ok()) { kd.heavy_list = rt.template alloc<int32_t>(R + 4); kd.n_heavy = kd.heavy_list + R; rt.memset0(kd.n_heavy, 16); }
		rt.launch_cold("dedup", R, kd);
		if (kd.heavy_list) rt.run_dedup_heavy("dedup_heavy", R, kd);
	}

	// ---- stage 5: mate rescue rounds
	void stage_rescue(const DeviceBatch &b, Work &w, BatchResult &out)
	{
		const int R = b.n_reads, NP = R / 2, slots = rt.max_slots();
		w.cap = rt.template alloc<int32_t>(R + 1); w.preg_off = rt.template alloc<int32_t>(R + 2); w.n_regs = rt.template alloc<int32_t>(R + 1);
		KPairCap kc{w.n_core, w.cap};
		rt.launch_wide("pair_cap", NP, kc);
		w.P = rt.exclusive_scan(w.cap, w.preg_off, R);
		const size_t P = (size_t)w.P + 1;
		w.pregs = rt.template alloc<Reg>(P); w.ptmp = rt.template alloc<Reg>(P); w.pidx = rt.template alloc<int32_t>(P);
		// a loop queues at most min(n_regs, MAX_RESCUE) SWs per pair, so both loops stay below 2P result slots; plus one single SW per pair
		w.rst = rt.template alloc<ResState>(NP + 1); w.stask = rt.template alloc<SwTask>(P + NP + 1); w.sres = rt.template alloc<U8Res>(2 * P + NP + 1);
		int32_t *n_slots = rt.template alloc<int32_t>(2); // [0] result slots handed out, [1] single SWs
		rt.memset0(n_slots, 8);
		const int q_cap = (b.max_len + 15) & ~15, t_cap = (PES_HIGH - PES_LOW + 2 * b.max_len + 31) & ~15;
		w.sw_scr = rt.template alloc<uint8_t>((size_t)slots * (q_cap + 2 * t_cap));
		uint8_t *hv = nullptr; int32_t *hv_list = nullptr, *n_hv = nullptr;
		if (rt.rescue_heavy_ok()) { hv = rt.template alloc<uint8_t>(NP + 8); hv_list = rt.template alloc<int32_t>(NP + 1); n_hv = rt.template alloc<int32_t>(2); rt.memset0(n_hv, 8); }
		KPairInit ki{w.occ_off, w.n_core, w.preg_off, w.regs, w.pregs, w.n_regs, w.rst, w.core_clean, w.cap, hv, hv_list, n_hv, getenv("ARX_RESCUE_HEAVY_MIN") ? atoi(getenv("ARX_RESCUE_HEAVY_MIN")) : RESCUE_HEAVY_MIN};
		rt.launch_wide("pair_init", NP, ki);
		int n_heavy = 0;
		if (hv) rt.d2h(&n_heavy, n_hv, 4);
		for (int round = 0;; ++round) {
			rt.memset0(w.counter, 4);
			KRescueStep ks{ix, b.lens, w.preg_off, w.pregs, w.ptmp, w.pidx, w.n_regs, w.rst, w.sres, w.stask, w.counter, n_slots, getenv("ARX_RESCUE_NO_AHEAD") ? 1 : 0, (int32_t)(2 * w.P), hv};
			if (n_heavy > 0) rt.run_rescue_heavy("rescue_heavy", n_heavy, hv_list, ks); // on the side stream: a few wavefronts' worth of work
			rt.launch_cold("rescue_step", NP, ks);
			rt.aux_join();
			int nt = read_counter(w);
			if (trace) { fprintf(stderr, "[arx] rescue round %d: %d tasks\n", round, nt); fflush(stderr); }
			if (nt == 0) {
				if (trace) { int32_t ns[2]; rt.d2h(ns, n_slots, 8); fprintf(stderr, "[arx] rescue: %d SWs queued ahead, %d single\n", ns[0], ns[1]); fflush(stderr); }
				break;
			}
			if (round > 2 * MAX_RESCUE + 4) { uint32_t e = ERR_INTERNAL; rt.h2d(w.err, &e, 4); break; }
			out.n_sw_tasks += nt; ++out.rescue_rounds;
			KSwU8 kw{ix, b.bases, b.base_off, b.lens, w.stask, w.sres, w.sw_scr, q_cap, t_cap};
			rt.run_sw_u8("sw_u8", nt, kw, b.max_len);
		}
	}

	// ---- stage 6: CIGAR for every region; retried with wider CIGAR slots if one overflows
	int stage_reg2aln(const DeviceBatch &b, Work &w)
	{
		const int slots = rt.max_slots_small(); // only the gapped minority needs DP scratch
		const int eh_words = 2 * (b.max_len + 2);
		const int z_cap = b.max_len * (2 * b.max_len + 64);
		const size_t P = (size_t)w.P + 1;
		w.z = rt.template alloc<uint8_t>((size_t)slots * z_cap);
		rt.free(w.eh); w.eh = rt.template alloc<int32_t>((size_t)slots * eh_words);
		w.alns = rt.template alloc<Aln>(P);
		w.nw_list = rt.template alloc<int32_t>(P);
		int32_t *nw_need = rt.template alloc<int32_t>(P + 1), *nw_zoff = rt.template alloc<int32_t>(P + 2), *big_list = rt.template alloc<int32_t>(P);
		int32_t *cnt2 = rt.template alloc<int32_t>(2 + NW_CLASSES), *class_list = rt.template alloc<int32_t>((size_t)NW_CLASSES * P + 1);
		for (w.cig_w = 16;; w.cig_w *= 2) {
			if (w.cig) rt.free(w.cig);
			w.cig = rt.template alloc<uint32_t>(P * w.cig_w);
			rt.memset0(cnt2, 4 * (2 + NW_CLASSES));
			KReg2Aln k{ix, b.bases, b.base_off, b.lens, w.preg_off, w.n_regs, b.n_reads, w.pregs, w.alns, w.cig, w.cig_w, w.eh, eh_words, w.z, z_cap, w.err,
			           w.nw_list, cnt2, 0, nw_need, big_list, class_list, (int)P};
			rt.launch("reg2aln", (int)w.P, k);
			int32_t n2[2 + NW_CLASSES];
			rt.d2h(n2, cnt2, 4 * (2 + NW_CLASSES));
			if (n2[0] > 0) { // gapped regions: every one gets its own slice of traceback matrix, sized from its band
				const int64_t units = rt.exclusive_scan(nw_need, nw_zoff, n2[0]);
				uint8_t *zbuf = rt.template alloc<uint8_t>((size_t)units * 64 + 64);
				k.mode = 1;
				rt.run_reg2aln_nw("reg2aln_nw", n2[0], n2 + 2, k, zbuf, nw_zoff);
			}
			k.mode = 2;
			rt.launch_small("reg2aln_nw_big", n2[1], k);
			rt.merge_sort_fail(w.err);
			uint32_t e = read_err(w);
			if (!(e & ERR_CIGAR_OVERFLOW)) { compact(b, w); return (int)e; }
			if (w.cig_w >= 1024) return (int)e;
			e &= ~ERR_CIGAR_OVERFLOW; rt.h2d(w.err, &e, 4);
		}
	}

	// ---- whole path for one batch (n_reads even: read 2i / 2i+1 are mates).  Returns 0 or an error bit set.
	int run(const DeviceBatch &b, BatchResult &out, Work &w)
	{
		int rc = stage_seed(b, w);
		if (rc) return rc;
		stage_chain(b, w);
		stage_extend(b, w, out);
		stage_rescue(b, w, out);
		rc = stage_reg2aln(b, w);
		out.n_occ = w.T;
		return rc;
	}

	// copy the final region lists and alignment records to the host, compacted
	// dense result arrays on the device: two scans (regions per read, CIGAR words per region) and one copy kernel
	void compact(const DeviceBatch &b, Work &w)
	{
		const int R = b.n_reads;
		w.c_reg_off = rt.template alloc<int32_t>(R + 2);
		w.c_n_regs = rt.exclusive_scan(w.n_regs, w.c_reg_off, R);
		const size_t NR = (size_t)w.c_n_regs;
		int32_t *cig_len = rt.template alloc<int32_t>(NR + 1), *cig_off = rt.template alloc<int32_t>(NR + 2);
		KCompactCount kc{w.preg_off, w.n_regs, w.c_reg_off, w.alns, cig_len};
		rt.launch_wide("compact_count", R, kc);
		w.c_n_cig = NR ? rt.exclusive_scan(cig_len, cig_off, (int)NR) : 0;
		w.c_regs = rt.template alloc<Reg>(NR + 1); w.c_alns = rt.template alloc<Aln>(NR + 1); w.c_cig = rt.template alloc<uint32_t>((size_t)w.c_n_cig + 1);
		KCompact kk{w.preg_off, w.n_regs, w.c_reg_off, cig_off, w.pregs, w.alns, w.cig, w.cig_w, w.c_regs, w.c_alns, w.c_cig};
		rt.launch_wide("compact", R, kk);
	}
	// straight into the caller's arrays (sized from arx_batch_counts)
	void fetch(const DeviceBatch &b, Work &w, int32_t *reg_off, Reg *regs, Aln *alns, uint32_t *cigars)
	{
		rt.d2h(reg_off, w.c_reg_off, 4 * ((size_t)b.n_reads + 1));
		rt.d2h(regs, w.c_regs, sizeof(Reg) * (size_t)w.c_n_regs);
		rt.d2h(alns, w.c_alns, sizeof(Aln) * (size_t)w.c_n_regs);
		rt.d2h(cigars, w.c_cig, 4 * (size_t)w.c_n_cig);
	}
};

} // namespace arx
