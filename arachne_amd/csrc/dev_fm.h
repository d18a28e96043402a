// dev_fm.h -- FM-index seeding on the device: Occ, bidirectional extension, SMEM search,
// the three seeding passes of BWA-MEM and sampled-SA locate.
// Agrees with bwt.c:53-115,169-274,289-379 and bwamem.c:114-162 of the reference.
#pragma once
#include "arx_dev.h"

namespace arx {

// ---- Occ block arithmetic.  The index files keep BWA's interleaved layout (per 128 symbols: 4 x u64 cumulative counts,
// then 8 x u32 of 16 symbols each, most significant first).  In HBM the same 64 bytes are re-packed when the index is
// opened (occ_repack_block) so that a count touches ONE 64-bit word instead of four:
//   words 0-3  low 32 bits of the four cumulative counts (symbols before the block)
//   word  4    bits 32-39 of the four counts, one byte per symbol (2^40 symbols: any genome BWA can index)
//   words 5-7  counts of the four symbols inside the block before symbol 32, 64 and 96, one byte per symbol
//   words 8-15 the 128 symbols as before
// A query for the first n symbols of the block takes the checkpoint of quarter (n-1)/32 and popcounts the quarter's word.
struct alignas(16) Q16 { uint32_t x, y, z, w; };
struct OccHead { Q16 lo, hx; }; // words 0-3, words 4-7: two 16-byte loads

ARX_HDI void occ_repack_block(uint32_t *blk) // in place, from BWA's layout; host side of the index upload
{
	uint64_t cum[4];
	for (int c = 0; c < 4; ++c) cum[c] = (uint64_t)blk[2 * c] | (uint64_t)blk[2 * c + 1] << 32; // little-endian u64
	uint32_t ck[3] = {0, 0, 0}, run[4] = {0, 0, 0, 0};
	for (int wd = 0; wd < 6; ++wd) {
		const uint32_t x = blk[8 + wd];
		for (int t = 0; t < 16; ++t) ++run[(x >> (30 - 2 * t)) & 3];
		if (wd & 1) ck[wd >> 1] = run[0] | run[1] << 8 | run[2] << 16 | run[3] << 24;
	}
	uint32_t hi = 0;
	for (int c = 0; c < 4; ++c) { blk[c] = (uint32_t)cum[c]; hi |= (uint32_t)((cum[c] >> 32) & 0xff) << (8 * c); }
	blk[4] = hi; blk[5] = ck[0]; blk[6] = ck[1]; blk[7] = ck[2];
}

ARX_DEVI OccHead load_head(const uint32_t *blk)
{
	const Q16 *p = (const Q16 *)blk; // 64-byte aligned
	OccHead h; h.lo = p[0]; h.hx = p[1];
	return h;
}
ARX_DEVI uint64_t load_quarter(const uint32_t *blk, int q) // symbols 32q .. 32q+31, symbol i of the quarter at bits 63-2i..62-2i
{
	struct alignas(8) D8 { uint32_t a, b; };
	const D8 d = *(const D8 *)(blk + 8 + 2 * q);
	return (uint64_t)d.a << 32 | d.b;
}
ARX_DEVI uint64_t head_cum(const OccHead &h, int c) // compile-time c in the callers' unrolled loops
{
	const uint32_t lo = c == 0 ? h.lo.x : c == 1 ? h.lo.y : c == 2 ? h.lo.z : h.lo.w;
	return (uint64_t)lo | (uint64_t)((h.hx.x >> (8 * c)) & 0xff) << 32;
}
ARX_DEVI uint32_t head_ck(const OccHead &h, int q) { return q == 0 ? 0u : q == 1 ? h.hx.y : q == 2 ? h.hx.z : h.hx.w; }

// counts of the four symbols among the first n (1..128) symbols of the block, added to the cumulative counts
ARX_DEVI void block_occ4(const uint32_t *blk, const OccHead &h, int n, uint64_t cnt[4])
{
	const int q = (n - 1) >> 5, nj = n - 32 * q; // nj in 1..32
	const uint64_t w = load_quarter(blk, q);
	const uint64_t keep = ~0ull << (64 - 2 * nj);
	const uint64_t lo = w & 0x5555555555555555ull & keep, hi = (w >> 1) & 0x5555555555555555ull & keep;
	const uint32_t p3 = (uint32_t)__builtin_popcountll(hi & lo);
	const uint32_t c2 = (uint32_t)__builtin_popcountll(hi) - p3, c1 = (uint32_t)__builtin_popcountll(lo) - p3;
	const uint32_t c0 = (uint32_t)nj - c1 - c2 - p3;
	const uint32_t ck = head_ck(h, q);
	cnt[0] = head_cum(h, 0) + (ck & 0xff) + c0;
	cnt[1] = head_cum(h, 1) + ((ck >> 8) & 0xff) + c1;
	cnt[2] = head_cum(h, 2) + ((ck >> 16) & 0xff) + c2;
	cnt[3] = head_cum(h, 3) + (ck >> 24) + p3;
}

// bwt_occ4 (bwt.c:169-187): counts in B[0..k] of the $-removed BWT.  Touches exactly one 64-byte block.
ARX_DEVI void occ4(const IndexView &ix, uint64_t k, uint64_t cnt[4])
{
	if (k == (uint64_t)-1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
	k -= (k >= ix.primary);
	const uint32_t *blk = ix.bwt + ((k >> 7) << 4);
	block_occ4(blk, load_head(blk), (int)(k & 127) + 1, cnt);
}

// bwt_2occ4 (bwt.c:189-217): both ends of an interval; when they fall into the same 64-byte block its head is fetched once
ARX_DEVI void occ4_pair(const IndexView &ix, uint64_t k, uint64_t l, uint64_t ck[4], uint64_t cl[4])
{
	if (k == (uint64_t)-1 || l == (uint64_t)-1) { occ4(ix, k, ck); occ4(ix, l, cl); return; }
	k -= (k >= ix.primary); l -= (l >= ix.primary);
	const uint32_t *bk = ix.bwt + ((k >> 7) << 4), *bl = ix.bwt + ((l >> 7) << 4);
	const OccHead hk = load_head(bk);
	OccHead hl = hk;
	if (bl != bk) hl = load_head(bl);
	block_occ4(bk, hk, (int)(k & 127) + 1, ck);
	block_occ4(bl, hl, (int)(l & 127) + 1, cl);
}

// symbol at position pos (0..127) of the block and its count among the block's first pos+1 symbols plus the cumulative count
ARX_DEVI uint64_t block_occ1_at(const uint32_t *blk, const OccHead &h, int pos, int *sym)
{
	const int q = pos >> 5, o = pos & 31;
	const uint64_t w = load_quarter(blk, q);
	const int c = (int)(w >> (62 - 2 * o)) & 3;
	const uint64_t keep = ~0ull << (62 - 2 * o);
	const uint64_t flip_lo = (c & 1) ? 0 : ~0ull, flip_hi = (c & 2) ? 0 : ~0ull;
	const uint32_t r = (uint32_t)__builtin_popcountll((w ^ flip_lo) & ((w >> 1) ^ flip_hi) & 0x5555555555555555ull & keep);
	const uint32_t ck = head_ck(h, q);
	const uint32_t lo = c == 0 ? h.lo.x : c == 1 ? h.lo.y : c == 2 ? h.lo.z : h.lo.w;
	*sym = c;
	return ((uint64_t)lo | (uint64_t)((h.hx.x >> (8 * c)) & 0xff) << 32) + ((ck >> (8 * c)) & 0xff) + r;
}

// bwt_extend (bwt.c:262-274), returning only the child for symbol c -- the SMEM search never looks at the other three.
// is_back = 1 extends x[0] (k) with the cumulative sizes fixing x[1] (l); is_back = 0 the other way round.
ARX_DEVI uint64_t sel4(const uint64_t v[4], int c) { return c == 0 ? v[0] : c == 1 ? v[1] : c == 2 ? v[2] : v[3]; } // keeps v[] in registers (a dynamic index would spill it)
ARX_DEVI Biv extend1(const IndexView &ix, const Biv &ik, int is_back, int c)
{
	uint64_t a = is_back ? ik.k : ik.l, b = is_back ? ik.l : ik.k;
	uint64_t tk[4], tl[4];
	occ4_pair(ix, a - 1, a - 1 + ik.s, tk, tl);
	uint64_t s3 = tl[3] - tk[3], s2 = tl[2] - tk[2], s1 = tl[1] - tk[1];
	uint64_t x = b + (a <= ix.primary && a + ik.s - 1 >= ix.primary); // position of child 3 on the other strand
	if (c < 3) x += s3;
	if (c < 2) x += s2;
	if (c < 1) x += s1;
	Biv ok;
	const uint64_t tkc = sel4(tk, c), tlc = sel4(tl, c);
	const uint64_t l2c = c == 0 ? ix.L2[0] : c == 1 ? ix.L2[1] : c == 2 ? ix.L2[2] : ix.L2[3];
	uint64_t na = l2c + 1 + tkc;
	ok.s = tlc - tkc;
	if (is_back) { ok.k = na; ok.l = x; } else { ok.l = na; ok.k = x; }
	ok.info = 0;
	return ok;
}

ARX_DEVI Biv set_intv(const IndexView &ix, int c) // bwt_set_intv (bwt.h:78)
{
	Biv ik;
	ik.k = ix.L2[c] + 1; ik.s = ix.L2[c + 1] - ix.L2[c]; ik.l = ix.L2[3 - c] + 1; ik.info = 0;
	return ik;
}

// one LF step of bwt_invPsi (bwt.c:53-59)
ARX_DEVI uint64_t lf_step(const IndexView &ix, uint64_t k)
{
	if (k == ix.primary) return 0;
	const uint64_t x = k - (k > ix.primary);           // row of the $-removed string holding B[k]
	const uint32_t *blk = ix.bwt + ((x >> 7) << 4);
	int c;
	// occ(k, c) counts B[0..x'] with x' = k - (k >= primary); for k != primary that is the same row x (k > primary <=> k >= primary)
	const uint64_t occ = block_occ1_at(blk, load_head(blk), (int)(x & 127), &c);
	return (c == 0 ? ix.L2[0] : c == 1 ? ix.L2[1] : c == 2 ? ix.L2[2] : ix.L2[3]) + occ;
}

// bwt_sa (bwt.c:86-96): LF steps until a sampled row
ARX_DEVI uint64_t sa_lookup(const IndexView &ix, uint64_t k)
{
	uint64_t sa = 0, mask = (uint64_t)ix.sa_intv - 1;
	while (k & mask) { ++sa; k = lf_step(ix, k); }
	return sa + ix.sa[k / (uint64_t)ix.sa_intv];
}

// Scratch for the SMEM searches of one read: two interval lists of up to len+1 entries each plus the per-call result list.
struct SmemScratch { Biv *v0, *v1, *mem; };

// mem_collect_intv (bwamem.c:114-162), first two passes: SMEM pass (bwt_smem1a, bwt.c:289-351, max_intv = 0) and the
// re-seeding pass from the middle of long rare SMEMs.  (Third pass: StratLane below; the final sort: seed_merge.)
//
// The reference nests these loops around bwt_extend(); here they are flattened into a resumable lane program: advance()
// runs the bookkeeping of one read until it needs its next extension (the only expensive step: two random 64-byte Occ
// blocks) or is finished, consume() takes the extension's result.  A caller that drives many lanes (hip_fm_coop.h) lets
// the 64 reads of a wavefront -- each somewhere else in its own forward/backward search -- reconverge on extend1(), and
// gives a lane its next read as soon as the previous one is finished.  Entries with equal info describe the same query
// substring and hence the same bi-interval, so any sort reproduces ks_introsort's result.
struct QBytes { const uint8_t *p; ARX_DEVI int at(int i) const { return p[i]; } };       // base codes 0..4, one per byte
struct QNibbles { const uint8_t *p; ARX_DEVI int at(int i) const { return (p[i >> 1] >> ((i & 1) << 2)) & 15; } }; // two per byte (LDS staging)

template <class Q> struct SeedLane {
	enum { ST_P1_NEXT, ST_P2_NEXT, ST_FWD, ST_FWD_DONE, ST_BWD_ROW, ST_BWD_J, ST_SMEM_DONE, ST_DONE };
	Biv *prev, *curr, *mem, *out;
	Q q;
	int len, cap, overflow;
	int state, pass;
	int n, old_n, k2;               // output count; pass-2 bookkeeping
	int x, min_intv, ret;           // current smem1 call
	int i, j, c, n_prev, n_curr, nm, sx;
	uint64_t curr_last_s;           // curr[n_curr - 1].s and the start of mem[nm - 1], kept in registers: both are looked at after every
	int mem_last_start;             // backward extension and would otherwise be dependent loads from the lists in HBM
	Biv ik;

	ARX_DEVI void start(const SmemScratch &sc, int len_, const Q &q_, Biv *out_, int cap_)
	{
		prev = sc.v0; curr = sc.v1; mem = sc.mem; out = out_; q = q_; len = len_; cap = cap_; overflow = 0;
		state = ST_P1_NEXT; pass = 1; n = old_n = k2 = 0; x = 0; min_intv = 1; ret = 0;
		i = j = c = n_prev = n_curr = nm = sx = 0; ik = Biv(); curr_last_s = 0; mem_last_start = 0;
	}
	ARX_DEVI bool done() const { return state == ST_DONE; }

	// Bookkeeping until the read needs an extension (true: *req extended by symbol *rc, backward if *rb) or is finished
	// (false).  The three states that ask for extensions are cheap; everything between two searches (list reversal, SMEM
	// output, picking the next start) is rare per lane but long.  With slow_ok = false the lane stops in front of such a
	// state (false, !done()): a wavefront driver lets lanes queue up there and runs them together, instead of paying for
	// every rare path in every iteration because one of its 64 lanes is in it.
	ARX_DEVI bool advance(const IndexView &ix, Biv *req, int *rb, int *rc, bool slow_ok = true)
	{
		while (state != ST_DONE) {
			switch (state) {
			case ST_FWD: // forward extension at query position i; the interval is remembered each time its size changes
				if (i >= len || q.at(i) > 3) { curr[n_curr++] = ik; state = ST_FWD_DONE; break; }
				*req = ik; *rb = 0; *rc = 3 - q.at(i);
				return true;
			case ST_BWD_J:
				if (j >= n_prev) {
					if (n_curr == 0) { state = ST_SMEM_DONE; break; }
					Biv *sw = curr; curr = prev; prev = sw; n_prev = n_curr;
					--i; state = ST_BWD_ROW;
					break;
				}
				*req = prev[j]; *rb = 1; *rc = c;
				return true;
			default:
				if (!slow_ok) return false;
				slow_step(ix);
				break;
			}
		}
		return false;
	}
	ARX_DEV void slow_step(const IndexView &ix)
	{
		switch (state) {
		case ST_P1_NEXT:
			while (x < len && q.at(x) > 3) ++x;
			if (x >= len) { old_n = n; k2 = 0; pass = 2; state = ST_P2_NEXT; break; }
			min_intv = 1; ik = set_intv(ix, q.at(x)); ik.info = x + 1; i = x + 1; n_curr = 0; nm = 0; state = ST_FWD;
			break;
		case ST_P2_NEXT: {
			bool found = false;
			while (k2 < old_n) { // re-seed from the middle of SMEMs that are long and occur rarely
				const Biv p = out[k2];
				const int start = (int)(p.info >> 32), end = (int)(uint32_t)p.info;
				if (end - start < OPT_SPLIT_LEN || p.s > (uint64_t)OPT_SPLIT_WIDTH) { ++k2; continue; }
				x = (start + end) >> 1; min_intv = (int)p.s + 1;
				found = true;
				break;
			}
			if (!found) { state = ST_DONE; break; } // the third pass runs on its own (StratLane)
			if (q.at(x) > 3) { nm = 0; ret = x + 1; state = ST_SMEM_DONE; break; } // bwt_smem1a returns at once on an ambiguous base
			ik = set_intv(ix, q.at(x)); ik.info = x + 1; i = x + 1; n_curr = 0; nm = 0; state = ST_FWD;
			break;
		}
		case ST_FWD_DONE: {
			for (int t = 0; t < n_curr >> 1; ++t) { Biv tmp = curr[n_curr - 1 - t]; curr[n_curr - 1 - t] = curr[t]; curr[t] = tmp; } // longest first
			ret = (int)curr[0].info;
			Biv *sw = curr; curr = prev; prev = sw; n_prev = n_curr;
			i = x - 1; state = ST_BWD_ROW;
			break;
		}
		case ST_BWD_ROW: // backward extension by query position i (-1 = before the read)
			if (i < -1) { state = ST_SMEM_DONE; break; }
			c = i < 0 ? -1 : (q.at(i) < 4 ? q.at(i) : -1);
			n_curr = 0; j = 0;
			if (c < 0) { // nothing can be extended: the longest interval survives if it is not contained
				if (n_prev > 0 && (nm == 0 || i + 1 < mem_last_start)) { Biv t = prev[0]; t.info |= (uint64_t)(i + 1) << 32; mem[nm++] = t; mem_last_start = i + 1; }
				state = ST_SMEM_DONE;
				break;
			}
			state = ST_BWD_J;
			break;
		case ST_SMEM_DONE:
			for (int t = nm - 1; t >= 0; --t) { // mem holds the SMEMs by decreasing start; emit them by increasing start
				const int slen = (int)((uint32_t)mem[t].info - (uint32_t)(mem[t].info >> 32));
				if (slen >= OPT_MIN_SEED_LEN) { if (n < cap) out[n++] = mem[t]; else overflow = 1; }
			}
			if (pass == 1) { x = ret; state = ST_P1_NEXT; } else { ++k2; state = ST_P2_NEXT; }
			break;
		default: break;
		}
	}
	// result of the extension advance() asked for
	ARX_DEVI void consume(const Biv &req, const Biv &ok)
	{
		if (state == ST_FWD) {
			if (ok.s != ik.s) {
				curr[n_curr++] = ik;
				if (ok.s < (uint64_t)min_intv) { state = ST_FWD_DONE; return; }
			}
			ik = ok; ik.info = i + 1; ++i;
		} else { // ST_BWD_J
			if (ok.s < (uint64_t)min_intv) {
				if (n_curr == 0 && (nm == 0 || i + 1 < mem_last_start)) { Biv t = req; t.info |= (uint64_t)(i + 1) << 32; mem[nm++] = t; mem_last_start = i + 1; }
			} else if (n_curr == 0 || ok.s != curr_last_s) {
				Biv t = ok; t.info = req.info;
				curr[n_curr++] = t; curr_last_s = ok.s;
			}
			++j;
		}
	}
};

// Third pass of mem_collect_intv: bwt_seed_strategy1 (bwt.c:358-379) from every position a match can start at -- the
// shortest forward match longer than min_seed_len that occurs fewer than max_mem_intv times.  It does not look at the
// results of the first two passes, so it is its own lane program (forward extensions only, next to no bookkeeping) and
// runs as its own kernel; seed_merge() joins the two interval lists.
constexpr int CAP_STRAT = 16; // a hit consumes at least min_seed_len + 1 bases: <= MAX_READ_LEN / 20 hits per read
template <class Q> struct StratLane {
	Q q; Biv *out; int len, n, x, i, sx; bool fresh, finished; Biv ik;
	ARX_DEVI void start(int len_, const Q &q_, Biv *out_) { q = q_; out = out_; len = len_; n = 0; x = 0; i = 0; sx = 0; fresh = true; finished = false; ik = Biv(); }
	ARX_DEVI bool done() const { return finished; }
	ARX_DEVI bool advance(const IndexView &ix, Biv *req, int *rc)
	{
		for (;;) {
			if (fresh) {
				while (x < len && q.at(x) > 3) ++x;
				if (x >= len) { finished = true; return false; }
				ik = set_intv(ix, q.at(x)); sx = x; i = x + 1; fresh = false;
			}
			if (i >= len) { finished = true; return false; } // bwt_seed_strategy1 returns len: the pass ends
			if (q.at(i) > 3) { x = i + 1; fresh = true; continue; }
			*req = ik; *rc = 3 - q.at(i);
			return true;
		}
	}
	ARX_DEVI void consume(const Biv &ok)
	{
		if (ok.s < (uint64_t)OPT_MAX_MEM_INTV && i - sx >= OPT_MIN_SEED_LEN) {
			if (ok.s > 0 && n < CAP_STRAT) { Biv t = ok; t.info = (uint64_t)sx << 32 | (uint32_t)(i + 1); out[n++] = t; }
			x = i + 1; fresh = true;
		} else { ik = ok; ++i; }
	}
};

// Both interval lists of a read -> out (capacity cap), sorted by info (bwamem.c:160).  Entries with equal info describe the
// same query substring and hence the same bi-interval, so any sort reproduces ks_introsort's result.  Returns the length.
ARX_DEV int seed_merge(Biv *out, int n12, const Biv *strat, int n3, int cap, int *overflow)
{
	int n = n12;
	for (int a = 0; a < n3; ++a) { if (n < cap) out[n++] = strat[a]; else *overflow = 1; }
	for (int a = 1; a < n; ++a) { // insertion sort by info
		Biv t = out[a];
		int b = a;
		while (b > 0 && out[b - 1].info > t.info) { out[b] = out[b - 1]; --b; }
		out[b] = t;
	}
	return n;
}

} // namespace arx
