// dev_regs.h -- alignment regions: the per-read extension state machine (mem_chain2aln), region de-duplication
// (mem_sort_dedup_patch / mem_patch_reg), the per-pair mate-rescue state machine (GoBwaMemMateSW driving mem_matesw)
// and region -> alignment record (mem_reg2aln).
//
// The reference extends seeds one after another because each decision looks at the regions found so far.  Here every
// read (pair) is a small state machine: a "step" kernel advances it to its next dynamic-programming task, a dense task
// kernel runs all pending tasks of the batch, and the next step consumes the results.
#pragma once
#include "arx_dev.h"
#include "dev_sw.h"

#ifndef ARX_STAT_RESCUE
#define ARX_STAT_RESCUE(pair, n, inserts, clean) ((void)0) // the host test double can count rescue work here
#endif
#ifndef ARX_RESCUE_FAST   // the host test double can switch dedup_insert() off, or check it against the general path
#define ARX_RESCUE_FAST 1
#define ARX_RESCUE_CROSSCHECK_BEGIN(ma, n, b) ((void)0)
#define ARX_RESCUE_CROSSCHECK_END(ix, ma, m) ((void)0)
#endif

namespace arx {

// ------------------------------------------------------------------------------------------------
// mem_chain2aln (bwamem.c:632-786) as a state machine
// ------------------------------------------------------------------------------------------------
// One state machine per CHAIN.  The reference walks a read's chains one after the other because the "is this seed already
// explained" test (bwamem.c:671-706) looks at every region found so far.  A region of chain c lies inside c's reference
// window [rmax0, rmax1) (the extensions never leave it), so it can only contain a seed that lies inside that window: a
// later chain none of whose seeds does is independent of c and runs side by side with it; one that has such a seed waits
// until c is finished and then sees c's regions.  The regions of a chain go to the slots of its own seeds; they are
// concatenated in chain order afterwards (KExtGather), which is the order the reference produces them in.
enum { PH_PICK = 0, PH_LEFT = 1, PH_RIGHT = 2, PH_DONE = 3, PH_WAIT = 4 };
enum { EXT_FINISHED = 0, EXT_TASK = 1, EXT_WAITING = 2 };

struct ExtState {
	int64_t rmax0, rmax1;
	Reg a;                 // region being built
	int32_t k, phase, band_try, n_regs, seed, sc0, aw0, aw1, prev, rstart;
	int32_t done_round;    // 0: running; otherwise the round it finished in (rounds count from 2; 1 = empty chain, finished at set-up)
	int32_t dep_cursor;    // earlier chains below this index are known not to hold it back
};

struct SeedKeyLt { // ascending (score<<32 | index) with score == len (bwamem.c:663-665); keys are unique
	const Seed *s;
	ARX_DEVI bool operator()(int a, int b) const { return s[a].len < s[b].len || (s[a].len == s[b].len && a < b); }
};

// set up the chain: reference window (bwamem.c:642-660) and the seed visiting order
ARX_DEV void ext_begin_chain(const IndexView &ix, int l_query, const Chain &c, const Seed *seeds, int *srt, ExtState &st)
{
	int64_t l_pac = ix.l_pac, r0 = l_pac << 1, r1 = 0;
	for (int i = 0; i < c.n; ++i) {
		const Seed t = seeds[i];
		int64_t b = t.rbeg - (t.qbeg + cal_max_gap(t.qbeg));
		int64_t e = t.rbeg + t.len + ((l_query - t.qbeg - t.len) + cal_max_gap(l_query - t.qbeg - t.len));
		r0 = r0 < b ? r0 : b;
		r1 = r1 > e ? r1 : e;
	}
	r0 = r0 > 0 ? r0 : 0;
	r1 = r1 < l_pac << 1 ? r1 : l_pac << 1;
	if (r0 < l_pac && l_pac < r1) {
		if (seeds[0].rbeg < l_pac) r1 = l_pac; else r0 = l_pac;
	}
	int rid;
	fetch_clamp(ix, &r0, seeds[0].rbeg, &r1, &rid);
	st.rmax0 = r0; st.rmax1 = r1;
	for (int i = 0; i < c.n; ++i) srt[i] = i;
	SeedKeyLt lt; lt.s = seeds;
	ks_introsort(c.n, srt, lt);
	st.k = c.n - 1;
}

// Does region p explain seed s (the body of the loop at bwamem.c:671-689)?
ARX_DEVI bool reg_explains_seed(const Reg &p, const Seed &s, int l_query)
{
	int64_t rd;
	int qd, w, max_gap;
	if (s.rbeg < p.rb || s.rbeg + s.len > p.re || s.qbeg < p.qb || s.qbeg + s.len > p.qe) return false;
	if (s.len - p.seedlen0 > .1 * l_query) return false;
	qd = s.qbeg - p.qb; rd = s.rbeg - p.rb;
	max_gap = cal_max_gap(qd < rd ? qd : (int)rd);
	w = max_gap < p.w ? max_gap : p.w;
	if (qd - rd < w && rd - qd < w) return true;
	qd = p.qe - (s.qbeg + s.len); rd = p.re - (s.rbeg + s.len);
	max_gap = cal_max_gap(qd < rd ? qd : (int)rd);
	w = max_gap < p.w ? max_gap : p.w;
	return qd - rd < w && rd - qd < w;
}

// Is seed s (srt position k of chain ci) already explained by an earlier region of this read?  (bwamem.c:671-706)
// Earlier regions = those of the earlier chains that finished before this round, and this chain's own.
ARX_DEV bool ext_seed_skipped(int l_query, const Chain *chains, int ci, const Seed *seeds, const int *srt, int k, const Reg *reg_pool,
                              const ExtState *states, int round, const Reg *own, int n_own)
{
	const Chain &c = chains[ci];
	const Seed s = seeds[srt[k]];
	bool hit = false;
	for (int c2 = 0; c2 < ci && !hit; ++c2) {
		const int dr = states[c2].done_round;
		if (dr <= 0 || dr >= round) continue; // still running: independent of this chain (see the dependency scan), its regions cannot contain s
		const Reg *av = reg_pool + chains[c2].seed_off;
		const int n_av = states[c2].n_regs;
		for (int i = 0; i < n_av; ++i) if (reg_explains_seed(av[i], s, l_query)) { hit = true; break; }
	}
	for (int i = 0; i < n_own && !hit; ++i) if (reg_explains_seed(own[i], s, l_query)) hit = true;
	if (!hit) return false;
	int i;
	for (i = k + 1; i < c.n; ++i) { // an extended, overlapping, off-diagonal seed keeps this one alive
		if (srt[i] < 0) continue;
		const Seed t = seeds[srt[i]];
		if (t.len < s.len * .95) continue;
		if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) break;
		if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) break;
	}
	return i == c.n;
}

ARX_DEVI void ext_make_task(ExtTask &t, int owner, int read_base, int l_query, const Seed &s, const ExtState &st, bool left)
{
	t.owner = owner;
	if (left) {
		t.qoff = read_base + s.qbeg - 1; t.qdir = -1; t.qlen = s.qbeg;
		t.tpos = s.rbeg - 1; t.tdir = -1; t.tlen = (int)(s.rbeg - st.rmax0);
		t.w = OPT_W << st.band_try; t.h0 = s.len * OPT_A;
	} else {
		int qe = s.qbeg + s.len;
		t.qoff = read_base + qe; t.qdir = 1; t.qlen = l_query - qe;
		t.tpos = s.rbeg + s.len; t.tdir = 1; t.tlen = (int)(st.rmax1 - (s.rbeg + s.len));
		t.w = OPT_W << st.band_try; t.h0 = st.sc0;
	}
}

// Set up chain ci of a read (bwamem.c:642-665): window, seed order; empty chains are finished at once.
ARX_DEV void ext_init_chain(const IndexView &ix, int l_query, const Chain &c, const Seed *seed_pool, int *srt_pool, ExtState &st)
{
	st = ExtState();
	st.k = -1; st.n_regs = 0; st.dep_cursor = 0;
	if (c.n == 0) { st.phase = PH_DONE; st.done_round = 1; return; }
	ext_begin_chain(ix, l_query, c, seed_pool + c.seed_off, srt_pool + c.seed_off, st);
	st.phase = PH_WAIT; st.done_round = 0;
}

// Advance chain ci until it needs a DP (EXT_TASK, task filled), has to wait for an earlier chain (EXT_WAITING) or has no
// seed left (EXT_FINISHED).  chains/states: the read's slices; reg_pool/seed_pool/srt_pool: batch pools indexed by
// Chain::seed_off; res: the result of the task emitted by the previous call; round: number of this step (>= 2).
ARX_DEV int ext_step(const IndexView &ix, int owner, int read_base, int l_query, const Chain *chains, int ci,
                     const Seed *seed_pool, int *srt_pool, Reg *reg_pool, const ExtState *states, int round, ExtState &st, const ExtRes &res, ExtTask &task)
{
	const Chain *c = &chains[ci];
	const Seed *seeds = seed_pool + c->seed_off;
	int *srt = srt_pool + c->seed_off;
	Reg *av = reg_pool + c->seed_off;
	if (st.phase == PH_DONE) return EXT_FINISHED;
	if (st.phase == PH_WAIT) { // an earlier, unfinished chain whose window holds one of this chain's seeds?
		for (int c2 = st.dep_cursor; c2 < ci; ++c2) {
			const int dr = states[c2].done_round;
			if (dr > 0 && dr < round) continue;
			const int64_t w0 = states[c2].rmax0, w1 = states[c2].rmax1;
			bool inside = false;
			for (int i = 0; i < c->n && !inside; ++i) inside = seeds[i].rbeg >= w0 && seeds[i].rbeg + seeds[i].len <= w1;
			if (inside) { st.dep_cursor = c2; return EXT_WAITING; }
		}
		st.dep_cursor = ci;
		st.phase = PH_PICK;
	}
	for (;;) {
		if (st.phase == PH_LEFT) {
			const Seed s = seeds[st.seed];
			Reg &a = st.a;
			a.score = res.score;
			st.aw0 = OPT_W << st.band_try;
			if (!(a.score == st.prev || res.max_off < (st.aw0 >> 1) + (st.aw0 >> 2)) && st.band_try + 1 < OPT_MAX_BAND_TRY) {
				st.prev = a.score; ++st.band_try;
				ext_make_task(task, owner, read_base, l_query, s, st, true);
				return EXT_TASK;
			}
			if (res.gscore <= 0 || res.gscore <= a.score - OPT_PEN_CLIP5) { a.qb = s.qbeg - res.qle; a.rb = s.rbeg - res.tle; a.truesc = a.score; }
			else { a.qb = 0; a.rb = s.rbeg - res.gtle; a.truesc = res.gscore; }
			st.phase = PH_RIGHT; st.rstart = 0;
		}
		if (st.phase == PH_RIGHT) {
			const Seed s = seeds[st.seed];
			Reg &a = st.a;
			bool finish = false;
			if (!st.rstart) {
				if (s.qbeg + s.len != l_query) {
					st.sc0 = a.score; st.prev = a.score; st.band_try = 0; st.rstart = 1;
					ext_make_task(task, owner, read_base, l_query, s, st, false);
					return EXT_TASK;
				}
				a.qe = l_query; a.re = s.rbeg + s.len;
				finish = true;
			} else {
				a.score = res.score;
				st.aw1 = OPT_W << st.band_try;
				if (!(a.score == st.prev || res.max_off < (st.aw1 >> 1) + (st.aw1 >> 2)) && st.band_try + 1 < OPT_MAX_BAND_TRY) {
					st.prev = a.score; ++st.band_try;
					ext_make_task(task, owner, read_base, l_query, s, st, false);
					return EXT_TASK;
				}
				int qe = s.qbeg + s.len;
				if (res.gscore <= 0 || res.gscore <= a.score - OPT_PEN_CLIP3) { a.qe = qe + res.qle; a.re = s.rbeg + s.len + res.tle; a.truesc += a.score - st.sc0; }
				else { a.qe = l_query; a.re = s.rbeg + s.len + res.gtle; a.truesc += res.gscore - st.sc0; }
				finish = true;
			}
			if (finish) {
				a.seedcov = 0;
				for (int i = 0; i < c->n; ++i) {
					const Seed t = seeds[i];
					if (t.qbeg >= a.qb && t.qbeg + t.len <= a.qe && t.rbeg >= a.rb && t.rbeg + t.len <= a.re) a.seedcov += t.len;
				}
				a.w = st.aw0 > st.aw1 ? st.aw0 : st.aw1;
				a.seedlen0 = s.len;
				a.frac_rep = c->frac_rep;
				av[st.n_regs++] = a;
				--st.k;
				st.phase = PH_PICK;
			}
		}
		// PH_PICK
		if (st.k < 0) { st.phase = PH_DONE; st.done_round = round; return EXT_FINISHED; }
		if (ext_seed_skipped(l_query, chains, ci, seeds, srt, st.k, reg_pool, states, round, av, st.n_regs)) { srt[st.k] = -1; --st.k; continue; }
		{
			st.seed = srt[st.k];
			const Seed s = seeds[st.seed];
			Reg &a = st.a;
			a = Reg();
			a.rb = a.re = 0; a.qb = a.qe = 0; a.sub = a.alt_sc = a.csub = a.sub_n = a.seedcov = a.secondary = a.secondary_all = a.n_comp = a.is_alt = 0; a.pad = 0;
			a.w = st.aw0 = st.aw1 = OPT_W;
			a.score = a.truesc = -1;
			a.rid = c->rid;
			a.seedlen0 = 0; a.frac_rep = 0.f;
			if (s.qbeg) {
				st.phase = PH_LEFT; st.band_try = 0; st.prev = -1;
				ext_make_task(task, owner, read_base, l_query, s, st, true);
				return EXT_TASK;
			}
			a.score = a.truesc = s.len * OPT_A; a.qb = 0; a.rb = s.rbeg;
			st.phase = PH_RIGHT; st.rstart = 0;
		}
	}
}

// ------------------------------------------------------------------------------------------------
// mem_sort_dedup_patch (bwamem.c:437-489) + mem_patch_reg (bwamem.c:406-435)
// ------------------------------------------------------------------------------------------------
struct RegReLt { const Reg *r; ARX_DEVI bool operator()(int a, int b) const { return r[a].re < r[b].re; } };
struct RegScoreLt {
	const Reg *r;
	ARX_DEVI bool operator()(int x, int y) const
	{
		const Reg &a = r[x], &b = r[y];
		return a.score > b.score || (a.score == b.score && (a.rb < b.rb || (a.rb == b.rb && a.qb < b.qb)));
	}
};

template <class LT>
ARX_DEV void permute_regs(int n, Reg *a, Reg *tmp, int *idx, LT lt)
{
	for (int i = 0; i < n; ++i) idx[i] = i;
	ks_introsort(n, idx, lt);
	for (int i = 0; i < n; ++i) tmp[i] = a[idx[i]];
	for (int i = 0; i < n; ++i) a[i] = tmp[i];
}

// query == nullptr disables patching (the mate-rescue call site passes bns = pac = query = 0, bwamem_pair.c:175)
ARX_DEV int patch_reg(const IndexView &ix, const uint8_t *query, const Reg &a, const Reg &b, int32_t *eh, int *w_)
{
	if (query == 0) return 0;
	if (a.rb < ix.l_pac && b.rb >= ix.l_pac) return 0;
	if (a.qb >= b.qb || a.qe >= b.qe || a.re >= b.re) return 0;
	int w = (int)((a.re - b.rb) - (a.qe - b.qb));
	w = w > 0 ? w : -w;
	double r = (double)(a.re - b.rb) / (double)(b.re - a.rb) - (double)(a.qe - b.qb) / (double)(b.qe - a.qb);
	r = r > 0. ? r : -r;
	if (a.re < b.rb || a.qe < b.qb) { if (w > OPT_W << 1 || r >= (double)0.05f) return 0; }
	else if (w > OPT_W << 2 || r >= (double)(0.05f * 2)) return 0;
	w += a.w + b.w;
	w = w < OPT_W << 2 ? w : OPT_W << 2;
	int score = 0, nc, nm;
	gen_cigar2(ix, w, query, a.qb, b.qe, a.rb, b.re, eh, 0, false, 0, 0, &score, &nc, &nm);
	int q_s = (int)((double)(b.qe - a.qb) / ((b.qe - b.qb) + (a.qe - a.qb)) * (b.score + a.score) + .499);
	int r_s = (int)((double)(b.re - a.rb) / (double)((b.re - b.rb) + (a.re - a.rb)) * (b.score + a.score) + .499);
	if ((double)score / (q_s > r_s ? q_s : r_s) < (double)0.90f) return 0;
	*w_ = w;
	return score;
}

#ifndef ARX_DEDUP_ATTR
#define ARX_DEDUP_ATTR
#endif
ARX_DEV ARX_DEDUP_ATTR int sort_dedup_patch(const IndexView &ix, const uint8_t *query, int n, Reg *a, Reg *tmp, int *idx, int32_t *eh, int32_t *patched = nullptr)
{ // *patched is set when two regions were merged (mem_patch_reg): only then can the result fail to be a fixed point of the pass
	int m, i, j;
	if (n <= 1) return n;
	RegReLt lt1; lt1.r = a;
	permute_regs(n, a, tmp, idx, lt1);
	for (i = 0; i < n; ++i) a[i].n_comp = 1;
	for (i = 1; i < n; ++i) {
		// structured form of the reference's loop (bwamem.c:443-473): `live` replaces its continue/break exits
		bool live = !(a[i].rid != a[i - 1].rid || a[i].rb >= a[i - 1].re + OPT_MAX_CHAIN_GAP);
		for (j = i - 1; live && j >= 0; --j) {
			Reg p = a[i], q = a[j];
			if (!(p.rid == q.rid && p.rb < q.re + OPT_MAX_CHAIN_GAP)) { live = false; }
			else if (q.qe != q.qb) { // a[j] has not been excluded
				int64_t orr = q.re - p.rb;
				int64_t oq = q.qb < p.qb ? q.qe - p.qb : p.qe - q.qb;
				int64_t mr = q.re - q.rb < p.re - p.rb ? q.re - q.rb : p.re - p.rb;
				int64_t mq = q.qe - q.qb < p.qe - p.qb ? q.qe - q.qb : p.qe - p.qb;
				int score = 0, w = 0;
				if ((float)orr > OPT_MASK_LEVEL_REDUN * (float)mr && (float)oq > OPT_MASK_LEVEL_REDUN * (float)mq) {
					if (p.score < q.score) { a[i].qe = p.qb; live = false; }
					else a[j].qe = q.qb;
				} else if (q.rb < p.rb && (score = patch_reg(ix, query, q, p, eh, &w)) > 0) {
					p.n_comp += q.n_comp + 1;
					p.seedcov = p.seedcov > q.seedcov ? p.seedcov : q.seedcov;
					p.sub = p.sub > q.sub ? p.sub : q.sub;
					p.csub = p.csub > q.csub ? p.csub : q.csub;
					p.qb = q.qb; p.rb = q.rb;
					p.truesc = p.score = score;
					p.w = w;
					a[i] = p;
					a[j].qb = q.qe;
					if (patched) *patched = 1;
				}
			}
		}
	}
	m = 0;
	for (i = 0; i < n; ++i)
		if (a[i].qe > a[i].qb) { if (m != i) a[m] = a[i]; ++m; }
	n = m;
	RegScoreLt lt2; lt2.r = a;
	permute_regs(n, a, tmp, idx, lt2);
	for (i = 1; i < n; ++i)
		if (a[i].score == a[i - 1].score && a[i].rb == a[i - 1].rb && a[i].qb == a[i - 1].qb) a[i].qe = a[i].qb;
	m = n > 0 ? 1 : 0;
	for (i = 1; i < n; ++i)
		if (a[i].qe > a[i].qb) { if (m != i) a[m] = a[i]; ++m; }
	return m;
}

// ------------------------------------------------------------------------------------------------
// Mate rescue: GoBwaMemMateSW's two loops (/root/reference/src/gobwa/gobwa.go:285-324) around mem_matesw
// (bwamem_pair.c:111-180) with the fixed insert model (only FR valid, [-35, 500]).
// ------------------------------------------------------------------------------------------------
// The SW of one (anchor, mate) rescue depends only on the anchor region and the mate's sequence; what depends on earlier
// rescues of the same loop is whether it is skipped (a mate region already sits in the window) and how its result is merged.
// So each loop is run in two steps: every SW the loop could need, given the mate list as it stands when the loop starts, is
// queued at once (rescue_enumerate) and computed in one launch; then the loop is replayed in order with the results at hand.
// A rescue can only add skips, except when the merge drops the region that caused one; the replay then asks for that
// single SW the slow way (phase 1) -- rare, but it keeps the result exactly the sequential one.
struct ResState {
	int64_t rb, re;        // clamped window of the pending SW
	uint64_t spec_mask;    // bit k: the k-th anchor above the score threshold had its SW queued ahead
	int32_t clean[2];      // clean[o]: read o's list is a fixed point of mem_sort_dedup_patch (see matesw_apply)
	int32_t e, i, num, n_snap, best[2], phase, spec_off; // phase 0: replaying, 1: waiting for a single SW, 2: finished, 3: loop `e` not enumerated yet
	int32_t mask_fresh, pad; // 1: the mate list is still the one spec_mask was computed against (nothing applied since): the replay trusts the mask
};
struct SwTask { int64_t rb, re; int32_t pair, o, slot, pad; };
struct SwEmit { SwTask *tasks; int32_t *n_tasks, *n_slots; int32_t single_slot, no_ahead; }; // no_ahead (tests): queue nothing ahead, every SW takes the single path; n_tasks: this round's queue; n_slots: result slots handed out so far (they live until the stage ends); single_slot: this pair's phase-1 result

ARX_DEVI int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist) // mem_infer_dir (bwamem_pair.c:23-31)
{
	int r1 = (b1 >= l_pac), r2 = (b2 >= l_pac);
	int64_t p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
	*dist = p2 > b1 ? p2 - b1 : b1 - p2;
	return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

// The redundancy test of mem_sort_dedup_patch (bwamem.c:452-459) for p = the region later in `re` order, q = the earlier one
ARX_DEVI bool regs_redundant(const Reg &p, const Reg &q)
{
	const int64_t orr = q.re - p.rb;
	const int64_t oq = q.qb < p.qb ? q.qe - p.qb : p.qe - q.qb;
	const int64_t mr = q.re - q.rb < p.re - p.rb ? q.re - q.rb : p.re - p.rb;
	const int64_t mq = q.qe - q.qb < p.qe - p.qb ? q.qe - q.qb : p.qe - p.qb;
	return (float)orr > OPT_MASK_LEVEL_REDUN * (float)mr && (float)oq > OPT_MASK_LEVEL_REDUN * (float)mq;
}

// mem_sort_dedup_patch(list + b) when `list` (n >= 1 regions) is already a fixed point of it (sorted by (score desc, rb,
// qb), pairwise non-redundant, no patching since query = 0): only pairs with b can interact, so the two sorts and the
// quadratic pass collapse to one scan.  In `re` order b first meets the earlier regions of its contig within
// max_chain_gap, latest first (a redundant one with the smaller score goes; if that is b the scan stops), then every later
// region meets b the same way.  Regions of one contig are contiguous in `re` order, which makes the set of regions b meets
// independent of the others.  Returns the new length, or -1 when the outcome would depend on how introsort orders equal
// keys (a region with the same `re`, or the same (score, rb, qb)): the caller then takes the general path.
ARX_DEV int dedup_insert_long(const Reg &b_in, Reg *ma, int n, Reg *tmp, int *idx)
{
	Reg b = b_in;
	b.n_comp = 1;
	for (int i = 0; i < n; ++i) {
		if (ma[i].re == b.re) return -1;
		if (ma[i].score == b.score && ma[i].rb == b.rb && ma[i].qb == b.qb) return -1;
	}
	// earlier regions b meets, by decreasing re (insertion sort of the few candidates into idx)
	int nv = 0;
	for (int i = 0; i < n; ++i) {
		const Reg &q = ma[i];
		if (q.re < b.re && q.rid == b.rid && b.rb < q.re + OPT_MAX_CHAIN_GAP) {
			int at = nv++;
			while (at > 0 && ma[idx[at - 1]].re < q.re) { idx[at] = idx[at - 1]; --at; }
			idx[at] = i;
		}
	}
	bool b_gone = false;
	int n_gone = 0;
	for (int t = 0; t < nv && !b_gone; ++t) {
		Reg &q = ma[idx[t]];
		if (!regs_redundant(b, q)) continue;
		if (b.score < q.score) b_gone = true;
		else { q.qe = q.qb; ++n_gone; }
	}
	// later regions meet b, by increasing re; once b is gone nothing else can change
	if (!b_gone) {
		nv = 0;
		for (int i = 0; i < n; ++i) {
			const Reg &p = ma[i];
			if (p.re > b.re && p.rid == b.rid && p.rb < b.re + OPT_MAX_CHAIN_GAP) {
				int at = nv++;
				while (at > 0 && ma[idx[at - 1]].re > p.re) { idx[at] = idx[at - 1]; --at; }
				idx[at] = i;
			}
		}
		for (int t = 0; t < nv && !b_gone; ++t) {
			Reg &p = ma[idx[t]];
			if (!regs_redundant(p, b)) continue;
			if (p.score < b.score) { p.qe = p.qb; ++n_gone; }
			else b_gone = true;
		}
	}
	// survivors in the final order: the old ones keep theirs, b goes in front of the first one that sorts after it
	if (n_gone == 0) {
		for (int i = 0; i < n; ++i) ma[i].n_comp = 1;
		if (b_gone) return n;
		int at = 0;
		while (at < n && (ma[at].score > b.score || (ma[at].score == b.score && (ma[at].rb < b.rb || (ma[at].rb == b.rb && ma[at].qb < b.qb))))) ++at;
		for (int i = n; i > at; --i) ma[i] = ma[i - 1];
		ma[at] = b;
		return n + 1;
	}
	int m = 0;
	bool placed = b_gone;
	for (int i = 0; i < n; ++i) {
		const Reg &x = ma[i];
		if (!(x.qe > x.qb)) continue;
		if (!placed && !(x.score > b.score || (x.score == b.score && (x.rb < b.rb || (x.rb == b.rb && x.qb < b.qb))))) { tmp[m++] = b; placed = true; }
		tmp[m] = x; tmp[m].n_comp = 1; ++m;
	}
	if (!placed) tmp[m++] = b;
	for (int i = 0; i < m; ++i) ma[i] = tmp[i];
	return m;
}

// The same, for lists of up to 256 regions, arranged so that one thread keeps several loads in flight: the scan reads only the first
// 32 bytes of a region (rb, re, qb, qe, rid, score), four regions per round trip, and notes in bit masks which regions are redundant
// with b on either side; the order-dependent part of the reference's loop (who is looked at first, where it stops) is settled afterwards
// from the masks: on the earlier side b meets regions by decreasing `re` and dies at the first redundant one that scores higher (the
// "stopper"), the redundant ones met before it go; on the later side by increasing `re`, the stopper being the first redundant region
// that does not score lower.  Equal `re` keeps list order on both sides, as the insertion sorts of dedup_insert_long() do.
struct RegHead { int64_t rb, re; int32_t qb, qe, rid, score; }; // the first 32 bytes of Reg
ARX_DEV int dedup_insert(const Reg &b_in, Reg *ma, int n, Reg *tmp, int *idx)
{
	if (n > 256) return dedup_insert_long(b_in, ma, n, tmp, idx);
	Reg b = b_in;
	b.n_comp = 1;
	uint64_t E[4] = {0, 0, 0, 0}, L[4] = {0, 0, 0, 0};
	int s1 = -1, s2 = -1, at = n; // stoppers (earlier / later side), first region that does not sort before b
	int64_t s1_re = 0, s2_re = 0;
	for (int j0 = 0; j0 < n; j0 += 4) {
		RegHead h[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) h[u] = *(const RegHead *)(ma + (j0 + u < n ? j0 + u : n - 1));
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const int j = j0 + u;
			if (j >= n) continue;
			const RegHead &q = h[u];
			if (q.re == b.re) return -1;
			if (q.score == b.score && q.rb == b.rb && q.qb == b.qb) return -1;
			if (j < at && !(q.score > b.score || (q.score == b.score && (q.rb < b.rb || (q.rb == b.rb && q.qb < b.qb))))) at = j;
			if (q.rid != b.rid) continue;
			if (q.re < b.re) {
				if (!(b.rb < q.re + OPT_MAX_CHAIN_GAP)) continue;
				// regs_redundant(p = b, q)
				const int64_t orr = q.re - b.rb, oq = q.qb < b.qb ? q.qe - b.qb : b.qe - q.qb;
				const int64_t mr = q.re - q.rb < b.re - b.rb ? q.re - q.rb : b.re - b.rb, mq = q.qe - q.qb < b.qe - b.qb ? q.qe - q.qb : b.qe - b.qb;
				if (!((float)orr > OPT_MASK_LEVEL_REDUN * (float)mr && (float)oq > OPT_MASK_LEVEL_REDUN * (float)mq)) continue;
				E[j >> 6] |= 1ull << (j & 63);
				if (b.score < q.score && (s1 < 0 || q.re > s1_re)) { s1 = j; s1_re = q.re; } // equal re: the smaller index stays
			} else {
				if (!(q.rb < b.re + OPT_MAX_CHAIN_GAP)) continue;
				// regs_redundant(p = q, q = b)
				const int64_t orr = b.re - q.rb, oq = b.qb < q.qb ? b.qe - q.qb : q.qe - b.qb;
				const int64_t mr = b.re - b.rb < q.re - q.rb ? b.re - b.rb : q.re - q.rb, mq = b.qe - b.qb < q.qe - q.qb ? b.qe - b.qb : q.qe - q.qb;
				if (!((float)orr > OPT_MASK_LEVEL_REDUN * (float)mr && (float)oq > OPT_MASK_LEVEL_REDUN * (float)mq)) continue;
				L[j >> 6] |= 1ull << (j & 63);
				if (!(q.score < b.score) && (s2 < 0 || q.re < s2_re)) { s2 = j; s2_re = q.re; }
			}
		}
	}
	bool b_gone = s1 >= 0;
	uint64_t G[4] = {0, 0, 0, 0}; // regions that go
	int n_gone = 0;
	if (E[0] | E[1] | E[2] | E[3]) {
		for (int wd = 0; wd < 4; ++wd)
			for (uint64_t m = E[wd]; m; m &= m - 1) {
				const int j = wd * 64 + __builtin_ctzll(m);
				if (s1 >= 0) { const int64_t re = ma[j].re; if (!(re > s1_re || (re == s1_re && j < s1))) continue; }
				G[wd] |= 1ull << (j & 63); ++n_gone;
			}
	}
	if (!b_gone) {
		b_gone = s2 >= 0;
		if (L[0] | L[1] | L[2] | L[3]) {
			for (int wd = 0; wd < 4; ++wd)
				for (uint64_t m = L[wd]; m; m &= m - 1) {
					const int j = wd * 64 + __builtin_ctzll(m);
					if (s2 >= 0) { const int64_t re = ma[j].re; if (!(re < s2_re || (re == s2_re && j < s2))) continue; }
					G[wd] |= 1ull << (j & 63); ++n_gone;
				}
		}
	}
	if (n_gone == 0) {
		for (int i = 0; i < n; ++i) ma[i].n_comp = 1;
		if (b_gone) return n;
		// make room at `at`: four regions per round trip, from the top
		int i = n;
		for (; i - 4 >= at; i -= 4) {
			const Reg r0 = ma[i - 1], r1 = ma[i - 2], r2 = ma[i - 3], r3 = ma[i - 4];
			ma[i] = r0; ma[i - 1] = r1; ma[i - 2] = r2; ma[i - 3] = r3;
		}
		for (; i > at; --i) ma[i] = ma[i - 1];
		ma[at] = b;
		return n + 1;
	}
	for (int wd = 0; wd < 4; ++wd)
		for (uint64_t m = G[wd]; m; m &= m - 1) { Reg &x = ma[wd * 64 + __builtin_ctzll(m)]; x.qe = x.qb; } // as the reference marks them
	int m = 0;
	bool placed = b_gone;
	for (int i = 0; i < n; ++i) {
		const Reg &x = ma[i];
		if (!(x.qe > x.qb)) continue;
		if (!placed && !(x.score > b.score || (x.score == b.score && (x.rb < b.rb || (x.rb == b.rb && x.qb < b.qb))))) { tmp[m++] = b; placed = true; }
		tmp[m] = x; tmp[m].n_comp = 1; ++m;
	}
	if (!placed) tmp[m++] = b;
	for (int i = 0; i < m; ++i) ma[i] = tmp[i];
	return m;
}

// Insert the rescued region and re-sort (bwamem_pair.c:150-176).  ma has room for one more entry.
// mem_sort_dedup_patch runs after every SW of the loop, also after one that added nothing (bwamem_pair.c:175).  Without
// patching (query = 0 here) it is idempotent on its own output: the survivors of the redundancy pass were all compared with
// each other and found distinct (the overlap test does not depend on which of two equal-`re` regions comes first), the final
// order is the strict order by (score, rb, qb), and n_comp is 1 throughout.  So once the list has been through it (*clean)
// and until something is inserted again the call is skipped, and an insertion into such a list takes dedup_insert().
ARX_DEV int matesw_apply(const IndexView &ix, const Reg &a, int l_ms, const U8Res &aln, int64_t rb, Reg *ma, int n_ma, Reg *tmp, int *idx, int32_t *clean, int32_t *fresh)
{
	const int64_t l_pac = ix.l_pac;
	const bool inserts = aln.score >= OPT_MIN_SEED_LEN && aln.qb >= 0;
	ARX_STAT_RESCUE(pair_id_for_stats, n_ma, inserts, *clean);
	if (!inserts && *clean) return n_ma;
	*fresh = 0; // the list may change from here on
	if (inserts) { // is_rev == 1 for the FR orientation
		Reg b = Reg();
		b.rb = b.re = 0; b.truesc = b.sub = b.alt_sc = b.sub_n = b.w = b.secondary_all = b.seedlen0 = b.n_comp = 0; b.frac_rep = 0.f; b.pad = 0;
		b.rid = a.rid;
		b.is_alt = a.is_alt;
		b.qb = l_ms - (aln.qe + 1);
		b.qe = l_ms - aln.qb;
		b.rb = (l_pac << 1) - (rb + aln.te + 1);
		b.re = (l_pac << 1) - (rb + aln.tb);
		b.score = aln.score;
		b.csub = aln.score2;
		b.secondary = -1;
		b.seedcov = (int)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
		if (*clean == 2 && n_ma >= 1 && ARX_RESCUE_FAST) { // the list went through a full pass with at least two regions
			ARX_RESCUE_CROSSCHECK_BEGIN(ma, n_ma, b);
			const int m = dedup_insert(b, ma, n_ma, tmp, idx);
			ARX_RESCUE_CROSSCHECK_END(ix, ma, m);
			if (m >= 0) return m;
		}
		int i, at;
		++n_ma;
		for (i = 0; i < n_ma - 1; ++i) if (ma[i].score < b.score) break;
		at = i;
		for (i = n_ma - 1; i > at; --i) ma[i] = ma[i - 1];
		ma[at] = b;
	}
	*clean = n_ma >= 2 ? 2 : 1; // with fewer than two regions the pass returns at once and leaves n_comp as it is
	return sort_dedup_patch(ix, 0, n_ma, ma, tmp, idx, 0);
}

// Window of the FR rescue of anchor `a` (bwamem_pair.c:126-133 with r = 1: mate reversed, larger coordinate)
ARX_DEVI bool rescue_window(const IndexView &ix, const Reg &a, int l_ms, int64_t *prb, int64_t *pre)
{
	int64_t rb = a.rb + PES_LOW - l_ms, re = a.rb + PES_HIGH;
	int rid = -1;
	if (rb < 0) rb = 0;
	if (re > ix.l_pac << 1) re = ix.l_pac << 1;
	if (rb < re) fetch_clamp(ix, &rb, (rb + re) >> 1, &re, &rid);
	*prb = rb; *pre = re;
	return a.rid == rid && re - rb >= OPT_MIN_SEED_LEN;
}
ARX_DEVI bool rescue_skipped(const IndexView &ix, const Reg &a, const Reg *ma, int n_ma) // bwamem_pair.c:118-124
{
	// eight list entries per round trip to memory: the loads of a group do not depend on each other (a pair in a 200-copy repeat
	// walks 50 anchors over a 200-entry mate list, one thread)
	for (int j0 = 0; j0 < n_ma; j0 += 8) {
		int64_t rb[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) rb[u] = ma[j0 + u < n_ma ? j0 + u : n_ma - 1].rb;
		bool hit = false;
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			int64_t dist;
			const int r = infer_dir(ix.l_pac, a.rb, rb[u], &dist);
			hit = hit || (r == 1 && dist >= PES_LOW && dist <= PES_HIGH); // entries past the end repeat the last one: same answer
		}
		if (hit) return true;
	}
	return false;
}

// Every SW loop st.e would run against the mate list as it is now.  out == nullptr: count only and compute *mask; otherwise *mask is
// the result of the counting pass (nothing changed in between) and says which anchors get a task.
ARX_DEV int rescue_enumerate(const IndexView &ix, int pair, const int *lens2, Reg *const regs[2], int *const n_regs[2], const ResState &st, uint64_t *mask,
                             SwTask *out, int slot0)
{
	const int e = st.e, o = 1 - e, l_ms = lens2[o];
	int num = 0, cnt = 0;
	const uint64_t known = *mask;
	if (!out) *mask = 0;
	if (l_ms <= 0) return 0;
	for (int i = 0; i < st.n_snap && num < MAX_RESCUE; ++i) {
		const Reg a = regs[e][i];
		if (a.score < st.best[e] - 25) continue;
		const int k = num++;
		if (out) { if (!(known >> k & 1)) continue; }
		else if (rescue_skipped(ix, a, regs[o], *n_regs[o])) continue;
		int64_t rb, re;
		if (!rescue_window(ix, a, l_ms, &rb, &re)) continue;
		if (!out) *mask |= (uint64_t)1 << k;
		if (out) { SwTask t; t.rb = rb; t.re = re; t.pair = pair; t.o = o; t.slot = slot0 + cnt; t.pad = 0; out[cnt] = t; }
		++cnt;
	}
	return cnt;
}

// Advance one pair until it has queued SWs whose results it needs (true) or both rescue loops are finished (false).
// regs[e]/n_regs[e]: the two reads' region lists (with spare capacity); sres: SW results by slot.
ARX_DEV bool rescue_step(const IndexView &ix, int pair, const int *lens2, Reg *const regs[2], int *const n_regs[2], Reg *const tmp[2], int *const idx[2],
                         ResState &st, const U8Res *sres, const SwEmit &emit)
{
	for (;;) {
		if (st.phase == 2) return false;
		if (st.phase == 3) { // start of a loop: queue its SWs
			uint64_t mask;
			int cnt = emit.no_ahead ? 0 : rescue_enumerate(ix, pair, lens2, regs, n_regs, st, &mask, nullptr, 0);
			if (emit.no_ahead) mask = 0;
			st.spec_mask = mask; st.spec_off = 0; st.i = 0; st.num = 0; st.phase = 0;
			st.mask_fresh = emit.no_ahead ? 0 : 1;
			if (cnt > 0) {
				st.spec_off = ARX_ATOMIC_ADD(emit.n_slots, cnt);
				rescue_enumerate(ix, pair, lens2, regs, n_regs, st, &mask, emit.tasks + ARX_ATOMIC_ADD(emit.n_tasks, cnt), st.spec_off);
				return true;
			}
		}
		const int e = st.e, o = 1 - e;
		if (st.phase == 1) { // the single SW came back: ma = the other read's list
			*n_regs[o] = matesw_apply(ix, regs[e][st.i], lens2[o], sres[emit.single_slot], st.rb, regs[o], *n_regs[o], tmp[o], idx[o], &st.clean[o], &st.mask_fresh);
			st.phase = 0; ++st.i;
		}
		if (st.i >= st.n_snap || st.num >= MAX_RESCUE || lens2[o] <= 0) {
			if (e == 1) { st.e = 0; st.n_snap = *n_regs[0]; st.phase = 3; continue; } // second loop walks the POST-rescue read-1 list
			st.phase = 2;
			return false;
		}
		const Reg a = regs[e][st.i];
		if (a.score < st.best[e] - 25) { ++st.i; continue; } // threshold stays the PRE-rescue best (gobwa.go:302-312)
		const int k = st.num++;
		// while the mate list is the one the mask was computed against, the mask is the answer of both tests below
		if (st.mask_fresh && !(st.spec_mask >> k & 1)) { ++st.i; continue; }
		if (!st.mask_fresh && rescue_skipped(ix, a, regs[o], *n_regs[o])) { ++st.i; continue; }
		int64_t rb, re;
		if (!rescue_window(ix, a, lens2[o], &rb, &re)) { ++st.i; continue; } // nothing aligned: ma stays as it is
		if (st.spec_mask >> k & 1) {
			const int slot = st.spec_off + __builtin_popcountll(st.spec_mask & (((uint64_t)1 << k) - 1));
			*n_regs[o] = matesw_apply(ix, a, lens2[o], sres[slot], rb, regs[o], *n_regs[o], tmp[o], idx[o], &st.clean[o], &st.mask_fresh);
			++st.i;
			continue;
		}
		st.rb = rb; st.re = re; st.phase = 1;
		SwTask t; t.rb = rb; t.re = re; t.pair = pair; t.o = o; t.slot = emit.single_slot; t.pad = 0;
		emit.tasks[ARX_ATOMIC_ADD(emit.n_tasks, 1)] = t;
		ARX_ATOMIC_INC(emit.n_slots + 1); // statistics: SWs that could not be queued ahead
		return true;
	}
}

// ------------------------------------------------------------------------------------------------
// mem_reg2aln (bwamem.c:1086-1156); mapq is not produced (Arachne never reads it)
// ------------------------------------------------------------------------------------------------
ARX_DEVI int infer_bw(int l1, int l2, int score, int a, int q, int r) // bwamem.c:792-799 with r == 1
{
	if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
	int w = (l1 < l2 ? l1 : l2) * a - score - q + 2;
	if (w < iabs(l1 - l2)) w = iabs(l1 - l2);
	return w;
}

// Band mem_reg2aln starts from (bwamem.c:1098-1103), and the bytes of traceback matrix its widest retry can need
ARX_DEVI int reg2aln_w0(const Reg &ar)
{
	const int l1 = ar.qe - ar.qb, l2 = (int)(ar.re - ar.rb);
	int tmp = infer_bw(l1, l2, ar.truesc, OPT_A, OPT_O_DEL, OPT_E_DEL);
	int w2 = infer_bw(l1, l2, ar.truesc, OPT_A, OPT_O_INS, OPT_E_INS);
	w2 = w2 > tmp ? w2 : tmp;
	if (w2 > OPT_W) w2 = w2 < ar.w ? w2 : ar.w;
	return w2;
}
// lane tiling (columns per lane: 1, 2, 4, 8 or 16 -> class 0..4) the 16-lane CIGAR kernel takes for band w_
ARX_DEVI int reg2aln_band_class(const Reg &ar, int w_)
{
	const int l1 = ar.qe - ar.qb, l2 = (int)(ar.re - ar.rb);
	int max_gap = ((l1 + 1) >> 1) - 5;
	max_gap = max_gap > 1 ? max_gap : 1;
	int w = (max_gap + iabs(l2 - l1) + 1) >> 1;
	w = w < w_ ? w : w_;
	const int min_w = iabs(l2 - l1) + 3;
	w = w > min_w ? w : min_w;
	const int n_col = l1 < 2 * w + 1 ? l1 : 2 * w + 1;
	return n_col <= 16 ? 0 : n_col <= 32 ? 1 : n_col <= 64 ? 2 : n_col <= 128 ? 3 : 4;
}
constexpr int NW_CLASSES = 5;
ARX_DEVI int64_t reg2aln_z_bound(const Reg &ar)
{
	const int l1 = ar.qe - ar.qb, l2 = (int)(ar.re - ar.rb);
	if (l1 <= 0 || l2 <= 0) return 0;
	int w_ = reg2aln_w0(ar);
	w_ = w_ < 0 ? 0 : w_;
	w_ = (w_ << 2) < (OPT_W << 2) ? (w_ << 2) : (OPT_W << 2); // at most two doublings, capped (bwamem.c:1104-1113)
	int max_gap = ((l1 + 1) >> 1) - 5;
	max_gap = max_gap > 1 ? max_gap : 1;
	int w = (max_gap + iabs(l2 - l1) + 1) >> 1;
	w = w < w_ ? w : w_;
	const int min_w = iabs(l2 - l1) + 3;
	w = w > min_w ? w : min_w;
	const int n_col = l1 < 2 * w + 1 ? l1 : 2 * w + 1;
	const int stride = n_col <= 16 ? 16 : n_col <= 32 ? 32 : n_col <= 64 ? 64 : n_col <= 128 ? 128 : 256; // rows are padded to the lane tiling of the 16-lane kernel
	return (int64_t)stride * l2;
}
constexpr int NW_Q_CAP = 256, NW_T_CAP = 1024; // what the 16-lane CIGAR kernel stages per region; larger regions take the one-thread path

// cg: cap words of output CIGAR for this region; returns false when cap is too small (caller retries with a larger slot)
ARX_DEV bool reg2aln(const IndexView &ix, int l_query, const uint8_t *query, const Reg &ar, int32_t *eh, uint8_t *z, uint32_t *cg, int cap, Aln &a)
{
	int i, w2, tmp, NM = -1, score = 0, is_rev, last_sc = -(1 << 30), n_cigar = 0;
	const int qb = ar.qb, qe = ar.qe;
	const int64_t rb = ar.rb, re = ar.re;
	a.flag = ar.secondary >= 0 ? 0x100 : 0;
	tmp = infer_bw(qe - qb, (int)(re - rb), ar.truesc, OPT_A, OPT_O_DEL, OPT_E_DEL);
	w2 = infer_bw(qe - qb, (int)(re - rb), ar.truesc, OPT_A, OPT_O_INS, OPT_E_INS);
	w2 = w2 > tmp ? w2 : tmp;
	if (w2 > OPT_W) w2 = w2 < ar.w ? w2 : ar.w;
	i = 0;
	do {
		w2 = w2 < OPT_W << 2 ? w2 : OPT_W << 2;
		gen_cigar2(ix, w2, query, qb, qe, rb, re, eh, z, true, cg + 1, cap - 2, &score, &n_cigar, &NM); // room for both clips
		if (n_cigar > cap - 2) return false;
		if (score == last_sc || w2 == OPT_W << 2) break;
		last_sc = score;
		w2 <<= 1;
	} while (++i < 3 && score < ar.truesc - OPT_A);
	a.NM = NM;
	int64_t pos = depos(ix, rb < ix.l_pac ? rb : re - 1, &is_rev);
	a.is_rev = is_rev;
	uint32_t *c0 = cg + 1;
	if (n_cigar > 0) { // squeeze out a leading or trailing deletion
		if ((c0[0] & 0xf) == 2) { pos += c0[0] >> 4; --n_cigar; ++c0; }
		else if ((c0[n_cigar - 1] & 0xf) == 2) --n_cigar;
	}
	if (qb != 0 || qe != l_query) {
		int clip5 = is_rev ? l_query - qe : qb, clip3 = is_rev ? qb : l_query - qe;
		if (clip5) { --c0; c0[0] = (uint32_t)clip5 << 4 | 3; ++n_cigar; }
		if (clip3) c0[n_cigar++] = (uint32_t)clip3 << 4 | 3;
	}
	if (c0 != cg) for (i = 0; i < n_cigar; ++i) cg[i] = c0[i]; // c0 >= cg, forward copy is safe
	a.n_cigar = n_cigar;
	a.rid = pos2rid(ix, pos);
	a.pos = pos - ix.ann_off[a.rid];
	a.score = ar.score; a.sub = ar.sub > ar.csub ? ar.sub : ar.csub;
	a.is_alt = ar.is_alt; a.alt_sc = ar.alt_sc;
	return true;
}

} // namespace arx
