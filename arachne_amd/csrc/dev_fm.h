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

// counts of the four symbols among the first n (1..128) symbols of the block, added to the cumulative counts; w = the quarter word
// holding symbol n - 1 (load_quarter(blk, (n - 1) >> 5))
ARX_DEVI void block_occ4_w(const OccHead &h, uint64_t w, int n, uint64_t cnt[4])
{
	const int q = (n - 1) >> 5, nj = n - 32 * q; // nj in 1..32
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
ARX_DEVI void block_occ4(const uint32_t *blk, const OccHead &h, int n, uint64_t cnt[4]) { block_occ4_w(h, load_quarter(blk, (n - 1) >> 5), n, cnt); }
// The same counts modulo 2^32, and what the block itself adds to each (at most 128) -- all an extension needs but for ONE symbol: the sizes of
// the four children are differences of two such counts and lie below 2^32 (an interval never holds more rows than the text has symbols of one
// kind on both strands), so they are exact in 32 bits; only the new interval's own row wants the 40-bit count (ext_finish).
ARX_DEVI void block_occ4_lo(const OccHead &h, uint64_t w, int n, uint32_t cnt[4], uint32_t add[4])
{
	const int q = (n - 1) >> 5, nj = n - 32 * q; // nj in 1..32
	const uint64_t keep = ~0ull << (64 - 2 * nj);
	const uint64_t lo = w & 0x5555555555555555ull & keep, hi = (w >> 1) & 0x5555555555555555ull & keep;
	const uint32_t p3 = (uint32_t)__builtin_popcountll(hi & lo);
	const uint32_t c2 = (uint32_t)__builtin_popcountll(hi) - p3, c1 = (uint32_t)__builtin_popcountll(lo) - p3;
	const uint32_t c0 = (uint32_t)nj - c1 - c2 - p3;
	const uint32_t ck = head_ck(h, q);
	add[0] = (ck & 0xff) + c0; add[1] = ((ck >> 8) & 0xff) + c1; add[2] = ((ck >> 16) & 0xff) + c2; add[3] = (ck >> 24) + p3;
	cnt[0] = h.lo.x + add[0]; cnt[1] = h.lo.y + add[1]; cnt[2] = h.lo.z + add[2]; cnt[3] = h.lo.w + add[3];
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

// extend1() cut in two, for kernels that want every load of an iteration in flight before anything waits (hip_fm_coop.h
// k_seed_bwd2): ext_issue() computes the two rows and loads their blocks (head + the one quarter each row needs), ext_finish() is
// the arithmetic.  Same values as extend1().
struct ExtLoad { OccHead hk, hl; uint64_t wk, wl; int nk, nl; }; // nk / nl: symbols counted in the block (0: the row is -1, counts are zero)
// `on` = the lane has a request; lanes without one (and rows that are -1) load block 0 and count nothing: the loads are issued by every
// lane without a branch around them, so the compiler has no reason to wait for one before issuing the next
ARX_DEVI void ext_issue(const IndexView &ix, const Biv &ik, int is_back, bool on, ExtLoad &L)
{
	const uint64_t a = is_back ? ik.k : ik.l;
	uint64_t k = a - 1, l = a - 1 + ik.s;
	const bool vk = on && k != (uint64_t)-1, vl = on && l != (uint64_t)-1;
	k -= (k >= ix.primary); l -= (l >= ix.primary);
	const uint32_t *bk = ix.bwt + (vk ? (k >> 7) << 4 : 0), *bl = ix.bwt + (vl ? (l >> 7) << 4 : 0);
	L.nk = vk ? (int)(k & 127) + 1 : 0;
	L.nl = vl ? (int)(l & 127) + 1 : 0;
	L.hk = load_head(bk); L.wk = load_quarter(bk, L.nk ? (L.nk - 1) >> 5 : 0);
	L.hl = load_head(bl); L.wl = load_quarter(bl, L.nl ? (L.nl - 1) >> 5 : 0);
}
ARX_DEVI uint32_t sel4u(const uint32_t v[4], int c) { return c == 0 ? v[0] : c == 1 ? v[1] : c == 2 ? v[2] : v[3]; }
// every symbol occurs fewer than 2^32 times in the text (both strands): then no interval holds 2^32 rows and the sizes of an extension's children
// are exact as 32-bit differences (GRCh38: 1.8 G at most); a uniform test on the index header, the general form stays for larger texts
ARX_DEVI bool occ_counts_fit32(const IndexView &ix) { return (((ix.L2[1] - ix.L2[0]) | (ix.L2[2] - ix.L2[1]) | (ix.L2[3] - ix.L2[2]) | (ix.L2[4] - ix.L2[3])) >> 32) == 0; }
template <bool FIT32 = false> // FIT32: the caller has checked occ_counts_fit32() (a kernel compiled for it: both forms in one kernel cost it a wavefront per SIMD)
ARX_DEVI Biv ext_finish(const IndexView &ix, const Biv &ik, int is_back, int c, const ExtLoad &L)
{
	const uint64_t a = is_back ? ik.k : ik.l, b = is_back ? ik.l : ik.k;
	uint64_t x = b + (a <= ix.primary && a + ik.s - 1 >= ix.primary);
	const uint64_t l2c = c == 0 ? ix.L2[0] : c == 1 ? ix.L2[1] : c == 2 ? ix.L2[2] : ix.L2[3];
	Biv ok;
	uint64_t na;
	if (FIT32) {
		// counts modulo 2^32 for the four symbols at both ends: the children's sizes are their differences; the 40-bit count is put together
		// for symbol c at the lower end only
		uint32_t tk[4] = {0, 0, 0, 0}, tl[4] = {0, 0, 0, 0}, ak[4] = {0, 0, 0, 0}, al[4];
		if (L.nk) block_occ4_lo(L.hk, L.wk, L.nk, tk, ak);
		if (L.nl) block_occ4_lo(L.hl, L.wl, L.nl, tl, al);
		const uint32_t s3 = tl[3] - tk[3], s2 = tl[2] - tk[2], s1 = tl[1] - tk[1];
		x += (uint64_t)(c < 3 ? s3 : 0u) + (uint64_t)(c < 2 ? s2 : 0u) + (uint64_t)(c < 1 ? s1 : 0u); // (three sizes may pass 2^32 together)
		na = l2c + 1 + (L.nk ? head_cum(L.hk, c) + sel4u(ak, c) : 0);
		ok.s = (uint64_t)(uint32_t)(sel4u(tl, c) - sel4u(tk, c));
	} else {
		uint64_t tk[4] = {0, 0, 0, 0}, tl[4] = {0, 0, 0, 0};
		if (L.nk) block_occ4_w(L.hk, L.wk, L.nk, tk);
		if (L.nl) block_occ4_w(L.hl, L.wl, L.nl, tl);
		const uint64_t s3 = tl[3] - tk[3], s2 = tl[2] - tk[2], s1 = tl[1] - tk[1];
		if (c < 3) x += s3;
		if (c < 2) x += s2;
		if (c < 1) x += s1;
		const uint64_t tkc = sel4(tk, c), tlc = sel4(tl, c);
		na = l2c + 1 + tkc;
		ok.s = tlc - tkc;
	}
	if (is_back) { ok.k = na; ok.l = x; } else { ok.l = na; ok.k = x; }
	ok.info = 0;
	return ok;
}

ARX_DEVI Biv set_intv(const IndexView &ix, int c) // bwt_set_intv (bwt.h:78)
{
	// selects, not ix.L2[c]: a lane-dependent index into the kernel's argument block sends the whole block to scratch memory, and every
	// field of it the loop reads afterwards comes back through a scratch load instead of a scalar register (k_seed_fwd1: 304 bytes per lane)
	const uint64_t l0 = ix.L2[0], l1 = ix.L2[1], l2 = ix.L2[2], l3 = ix.L2[3], l4 = ix.L2[4];
	Biv ik;
	ik.k = (c == 0 ? l0 : c == 1 ? l1 : c == 2 ? l2 : l3) + 1;
	ik.s = c == 0 ? l1 - l0 : c == 1 ? l2 - l1 : c == 2 ? l3 - l2 : l4 - l3;
	ik.l = (c == 0 ? l3 : c == 1 ? l2 : c == 2 ? l1 : l0) + 1;
	ik.info = 0;
	return ik;
}

// ---- k-mer table.  The bi-interval a forward extension reaches after its first K bases is a pure function of those K bases: a table
// of 4^K entries, filled once per arx_open by the very extension it replaces (level d + 1 from level d, api_impl.h), turns K - 1
// DEPENDENT extensions (two scattered Occ blocks each) into one independent 16-byte load.  Entry of the K-mer q[0..K) with code
// sum q[i] << 2i: k (40 bits) | l (40 bits) | s (40 bits) in two 64-bit words -- 2^40 symbols is what the Occ blocks hold as well.
ARX_DEVI void ktab_pack(const Biv &b, uint64_t *w) { w[0] = (b.k & 0xffffffffffull) | (b.l & 0xffffffull) << 40; w[1] = ((b.l >> 24) & 0xffffull) | (b.s & 0xffffffffffull) << 16; }
ARX_DEVI Biv ktab_unpack(uint64_t w0, uint64_t w1)
{
	Biv b; b.k = w0 & 0xffffffffffull; b.l = (w0 >> 40) | (w1 & 0xffffull) << 24; b.s = w1 >> 16; b.info = 0;
	return b;
}
ARX_DEVI Biv ktab_load(const IndexView &ix, uint64_t code)
{
	struct alignas(16) W2 { uint64_t a, b; };
	const W2 w = *(const W2 *)(ix.ktab + 2 * code);
	return ktab_unpack(w.a, w.b);
}

ARX_DEVI Biv klv_load(const IndexView &ix, int d, uint64_t code) // level d (1 .. klv_k), code of the d-mer
{
	struct alignas(16) W2 { uint64_t a, b; };
	const uint64_t at = ((((uint64_t)1 << (2 * d)) - 4) / 3) + code;
	const W2 w = *(const W2 *)(ix.klv + 2 * at);
	return ktab_unpack(w.a, w.b);
}

// ---- 40-bit arrays (the whole suffix array and its inverse, IndexView::sa40 / isa40): entry i in bytes [5 i, 5 i + 5), little-endian.
// Read as the two 32-bit words at the 4-byte boundary below it (the entry ends at most 24 + 40 bits into them); the arrays are padded.
ARX_DEVI uint64_t p40_decode(uint32_t w0, uint32_t w1, uint64_t i) { return (((uint64_t)w1 << 32 | w0) >> (8 * ((5 * i) & 3))) & 0xffffffffffull; }
ARX_DEVI uint64_t p40_load(const uint8_t *a, uint64_t i) { const uint32_t *w = (const uint32_t *)(a + ((5 * i) & ~(uint64_t)3)); return p40_decode(w[0], w[1], i); }
ARX_DEVI void p40_store(uint8_t *a, uint64_t i, uint64_t v) { uint8_t *p = a + 5 * i; p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); p[4] = (uint8_t)(v >> 32); }

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
	if (ix.sa40) return p40_load(ix.sa40, k); // the whole array is resident: no walk
	uint64_t sa = 0, mask = (uint64_t)ix.sa_intv - 1;
	while (k & mask) { ++sa; k = lf_step(ix, k); }
	return sa + ix.sa[k / (uint64_t)ix.sa_intv];
}

// ---- mem_collect_intv (bwamem.c:114-162), first two passes: the SMEM pass (bwt_smem1a, bwt.c:289-351, max_intv = 0) from
// successive start positions, and the re-seeding pass from the middle of long rare SMEMs.  (Third pass: StratLane below; the
// final sort: seed_merge.)
//
// bwt_smem1a(x) is a forward extension from x that remembers the interval each time its size changes, followed by a backward
// sweep over those intervals.  The reference nests both around bwt_extend() and runs the calls of a read one after the
// other; here they are cut into small lane programs that do nothing but one kind of extension each, so that the loop a
// wavefront runs is little more than extend1() (two random 64-byte Occ blocks) -- the same shape as the third pass, which
// reaches ~85 % of the scattered-read ceiling of the memory system (profiles/):
//   FwdLane   the forward extension from one start.  The next start of the first pass is where the forward extension ended
//             (bwt_smem1a's return value), so a read's starts chain through FwdLanes only.
//   BwdLane   the backward sweep of one start: an independent task once its interval list exists.
//   seed_gather_pass1 / _pass2   collect the SMEMs of a read's tasks in call order; the first also picks the re-seeding
//             starts (bwamem.c:131-146), each of which is again one FwdLane + one BwdLane task.
// A task's lists live in a slice of a batch-wide interval pool: [0, n) the forward list (longest first), [n, 2n) the
// second list of the sweep, [2n, 3n) the SMEMs it finds.
struct QBytes { const uint8_t *p; ARX_DEVI int at(int i) const { return p[i]; } };       // base codes 0..4, one per byte
struct QNibbles { const uint8_t *p; ARX_DEVI int at(int i) const { return (p[i >> 1] >> ((i & 1) << 2)) & 15; } }; // two per byte (LDS staging)
// the same rows laid out word-major across the 64 lanes of a wavefront (word w of lane x at byte 256 w + 4 x: what a lane-linear
// LDS-direct load writes); p points at the lane's word 0
struct QNibblesT { const uint8_t *p; ARX_DEVI int at(int i) const { return (p[((i >> 3) << 8) + ((i >> 1) & 3)] >> ((i & 1) << 2)) & 15; } };

struct SeedTask { // off/n: pool slice; nm: SMEMs found; next: the read's next task (-1: last)
	int32_t read, x, min_intv, off, n, nm, next;
	int32_t flip;               // a sweep handed over at a row boundary (hip_fm_coop.h): 1 if the lists have changed places,
	int32_t row, n_prev, mls, pad; // the row to go on with, the entries of its list, the start of the last SMEM found
};

struct SeedPools { // batch-wide, filled through atomic cursors; an overflow raises ERR_POOL_OVERFLOW and the read yields no more tasks
	Biv *pool; int64_t pool_cap; SeedTask *tasks; int32_t task_cap; int32_t *cursors; // cursors[0]: pool entries handed out, [1]: tasks
	uint32_t *err;
	ARX_DEVI int new_task(int read, int x, int min_intv, int n) const // reserves the task and 3n pool entries; -1 on overflow
	{
		const int t = ARX_ATOMIC_ADD(cursors + 1, 1);
		const int off = ARX_ATOMIC_ADD(cursors, 3 * n);
		if (t >= task_cap || (int64_t)off + 3 * n > pool_cap) {
			ARX_ATOMIC_OR(err, ERR_POOL_OVERFLOW);
			if (t < task_cap) { SeedTask e = SeedTask(); e.read = read; e.next = -1; tasks[t] = e; } // the id is taken: leave an empty task (n = 0) the later kernels skip
			return -1;
		}
		SeedTask k = SeedTask(); k.read = read; k.x = x; k.min_intv = min_intv; k.off = off; k.n = n; k.nm = 0; k.next = -1;
		tasks[t] = k;
		return t;
	}
};

// forward half of bwt_smem1a (bwt.c:299-321) from position x (q[x] is a base): list[] receives the interval each time its
// size changes, shortest match first
// With the per-depth k-mer tables (ix.klv; round 3, the wavefront kernels of hip_fm_coop.h only): the intervals of the first K bases are a
// function of those bases, so an extension whose first K bases are A/C/G/T inside the read takes the interval of the K-mer from the table
// (start_jump / take_jump: ONE load instead of K - 1 dependent extensions) provided that interval still holds min_intv occurrences --
// sizes only shrink with depth, so the walk cannot have ended earlier -- and goes on base by base from there.  The list entries the walk
// would have pushed on the way (depth d whenever the size changes from d to d + 1, bwt.c:308-313) are NOT stored by the lane: n_def says
// that depths 1 .. n_def are still owed, and the wavefront adds them from the tables when the list is exported to its pool slice
// (hip_fm_coop.h: persistent_lanes, grant step).
// TEXT MODE (round 3; needs IndexView::sa40 / isa40).  Once the interval of a first-pass extension holds ONE occurrence (min_intv = 1), every
// further base either keeps that occurrence or ends the walk: bwt_extend leaves k where it is (the only child is the one that matches) and
// moves l one LF step along the reverse strand.  The pattern P = q[x0, i) then lies at text position p = SA[k] and nowhere else, so the walk
// goes on exactly as long as the read equals the text T = forward strand + its reverse complement (bntseq.c:398-419) behind it: compare
// q[i ...] with T[p + i - x0 ...] until a difference, an ambiguous base, the end of the read or the end of the text, m bases on.  The interval
// bwt_smem1a would have reached base by base is (k, l', 1) with l' the row of the reverse complement of q[x0, i + m), which lies at the
// mirror position: l' = ISA[seq_len - (p + i + m - x0)].  Three dependent loads (SA, text, ISA) instead of m round trips to two Occ blocks;
// the list gets the same single entry the walk would have pushed when it ended (bwt.c:308-316).
// A lane in text mode asks for what it needs through advance() like an extension: *rc < 0 names the kind, req->k the row / position, and the
// answer -- 16 bytes from a 4-byte aligned address, aux_addr() -- comes back through consume_aux().
enum { RC_TAB = -1, RC_SA = -2, RC_TEXT = -3, RC_ISA = -4 };
enum { FL_NORMAL = 0, FL_SA = 1, FL_TEXT = 2, FL_ISA = 3 };

// first 32-bit word of the 16-byte chunk of the packed forward strand that holds text position t and what follows it in T: on the forward
// strand the words from t's own on, on the reverse strand (read downwards) the word of the mirrored position and the three below it
ARX_DEVI int64_t text_chunk_word(const IndexView &ix, uint64_t t)
{
	if ((int64_t)t < ix.l_pac) return (int64_t)(t >> 4);
	return (((ix.l_pac << 1) - 1 - (int64_t)t) >> 4) - 3; // (down to -3: the packed strand is resident behind 16 spare bytes, api_impl.h)
}
ARX_DEVI const uint32_t *aux_addr(const IndexView &ix, int rc, uint64_t x)
{
	if (rc == RC_SA) return (const uint32_t *)(ix.sa40 + ((5 * x) & ~(uint64_t)3));
	if (rc == RC_ISA) return (const uint32_t *)(ix.isa40 + ((5 * x) & ~(uint64_t)3));
	if (rc == RC_TEXT) return (const uint32_t *)ix.pac + text_chunk_word(ix, x);
	return (const uint32_t *)(ix.klv + 2 * (((((uint64_t)1 << (2 * ix.klv_k)) - 4) / 3) + x)); // RC_TAB: the klv_k-mer with code x
}

// q[i ...] against T[t ...] within the 16-byte chunk w0..w3 (the words at text_chunk_word(t)): the number of equal bases; *more: the chunk ran
// out first, and both the read and the text go on.  Base by base (any read layout; the host test double and the one-thread kernels):
template <class Q> ARX_DEVI int text_match_chunk(const IndexView &ix, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint64_t t, const Q &q, int i, int len, bool *more)
{
	const uint32_t w[4] = {w0, w1, w2, w3};
	const uint8_t *wb = (const uint8_t *)w;
	const int64_t base0 = text_chunk_word(ix, t) * 16;
	int m = 0, avail;
	if ((int64_t)t < ix.l_pac) {
		const int64_t a0 = (int64_t)t - base0;
		avail = (int)(64 - a0 < ix.l_pac - (int64_t)t ? 64 - a0 : ix.l_pac - (int64_t)t);
		for (; m < avail; ++m) {
			const int rel = (int)a0 + m;
			if (i + m >= len || q.at(i + m) != ((wb[rel >> 2] >> ((~rel & 3) << 1)) & 3)) break;
		}
	} else {
		const int64_t u = (ix.l_pac << 1) - 1 - (int64_t)t;
		avail = (int)(u - base0 + 1 < u + 1 ? u - base0 + 1 : u + 1);
		for (; m < avail; ++m) {
			const int rel = (int)(u - base0) - m;
			if (i + m >= len || q.at(i + m) != 3 - ((wb[rel >> 2] >> ((~rel & 3) << 1)) & 3)) break;
		}
	}
	*more = m == avail && i + m < len && t + (uint64_t)m < ix.seq_len;
	return m;
}

// The same for a read staged as a row of 4-bit codes (QNibbles: the wavefront kernels), 64 bases at a time on words: both sides become streams
// of 2-bit codes, first base in the lowest bits -- the text by undoing the packed strand's byte order (forward strand: MSB-first within a
// byte, bytes ascending; the reverse strand is read downwards and complemented, which is the byte-swapped word inverted), the read by
// squeezing the low two bits out of every nibble, with bit 2 of a nibble (codes above 3) kept as a difference of its own.
ARX_DEVI uint32_t text_rev32(uint32_t x)
{
#if defined(__clang__)
	return __builtin_bitreverse32(x);
#else
	x = (x >> 16) | (x << 16); x = ((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8); x = ((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4);
	x = ((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2); return ((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1);
#endif
}
ARX_DEVI uint32_t text_fwd_word(uint32_t r) { const uint32_t x = text_rev32(__builtin_bswap32(r)); return ((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u); }
ARX_DEVI uint32_t text_rev_word(uint32_t r) { return ~__builtin_bswap32(r); }
ARX_DEVI uint32_t nib_squeeze(uint32_t n) { uint32_t y = n & 0x33333333u; y = (y | y >> 2) & 0x0f0f0f0fu; y = (y | y >> 4) & 0x00ff00ffu; return (y | y >> 8) & 0xffffu; }
ARX_DEVI uint32_t text_funnel(uint32_t hi, uint32_t lo, int sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh); } // sh in [0, 31]
ARX_DEVI int text_match_chunk(const IndexView &ix, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint64_t t, const QNibbles &q, int i, int len, bool *more)
{
	uint32_t s0, s1, s2, s3;
	int sh, avail;
	if ((int64_t)t < ix.l_pac) {
		const int a0 = (int)(t & 15);
		s0 = text_fwd_word(w0); s1 = text_fwd_word(w1); s2 = text_fwd_word(w2); s3 = text_fwd_word(w3);
		sh = 2 * a0;
		avail = (int)(64 - a0 < ix.l_pac - (int64_t)t ? 64 - a0 : ix.l_pac - (int64_t)t);
	} else {
		const int64_t u = (ix.l_pac << 1) - 1 - (int64_t)t;
		const int bu = (int)(u & 15);
		s0 = text_rev_word(w3); s1 = text_rev_word(w2); s2 = text_rev_word(w1); s3 = text_rev_word(w0);
		sh = 2 * (15 - bu);
		avail = (int)(bu + 49 < u + 1 ? bu + 49 : u + 1);
	}
	const uint32_t z0 = text_funnel(s1, s0, sh), z1 = text_funnel(s2, s1, sh), z2 = text_funnel(s3, s2, sh), z3 = s3 >> sh;
	const uint32_t *row = (const uint32_t *)q.p + (i >> 3);
	const int rs = 2 * (i & 7);
	uint32_t r[5], f[5];
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll
#endif
	for (int j = 0; j < 5; ++j) {
		const uint32_t a = row[2 * j], b = j < 4 ? row[2 * j + 1] : 0;
		r[j] = nib_squeeze(a) | nib_squeeze(b) << 16;
		f[j] = nib_squeeze(a >> 2) | nib_squeeze(b >> 2) << 16;
	}
	const uint32_t d0 = (z0 ^ text_funnel(r[1], r[0], rs)) | text_funnel(f[1], f[0], rs), d1 = (z1 ^ text_funnel(r[2], r[1], rs)) | text_funnel(f[2], f[1], rs);
	const uint32_t d2 = (z2 ^ text_funnel(r[3], r[2], rs)) | text_funnel(f[3], f[2], rs), d3 = (z3 ^ text_funnel(r[4], r[3], rs)) | text_funnel(f[4], f[3], rs);
	int m = d0 ? __builtin_ctz(d0) >> 1 : d1 ? 16 + (__builtin_ctz(d1) >> 1) : d2 ? 32 + (__builtin_ctz(d2) >> 1) : d3 ? 48 + (__builtin_ctz(d3) >> 1) : 64;
	if (m > avail) m = avail;
	if (m > len - i) m = len - i;
	*more = m == avail && i + m < len && t + (uint64_t)m < ix.seq_len;
	return m;
}

// The mirror image for the backward direction (the entry-parallel backward sweeps, hip_fm_coop.h k_seed_bwd_e): q[i], q[i - 1], ... against
// T[t], T[t - 1], ..., at most 24 bases per call.  The chunk is the 16 bytes at text_chunk_word_back(t): on the forward strand the word of t and
// the three below it (read downwards: the byte-swapped word is that order), on the reverse strand the words of the mirrored position upwards,
// complemented.  qw0 holds q[i] (the word i >> 3 of the read's row of 4-bit codes), qw1 .. qw3 the words below it (word 0 again below index 0).
ARX_DEVI int64_t text_chunk_word_back(const IndexView &ix, uint64_t t)
{
	if ((int64_t)t < ix.l_pac) return (int64_t)(t >> 4) - 3; // (down to -3: the spare bytes in front of the packed strand)
	return ((ix.l_pac << 1) - 1 - (int64_t)t) >> 4;
}
ARX_DEVI uint32_t nib_reverse(uint32_t x) { return __builtin_bswap32(((x & 0x0f0f0f0fu) << 4) | ((x >> 4) & 0x0f0f0f0fu)); } // the eight 4-bit codes of a word in reverse order
ARX_DEVI int text_match_back(const IndexView &ix, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint64_t t, uint32_t qw0, uint32_t qw1, uint32_t qw2, uint32_t qw3, int i, bool *more)
{
	uint32_t s0, s1, s2, s3;
	int sh, avail;
	if ((int64_t)t < ix.l_pac) {
		const int bt = (int)(t & 15);
		s0 = __builtin_bswap32(w3); s1 = __builtin_bswap32(w2); s2 = __builtin_bswap32(w1); s3 = __builtin_bswap32(w0);
		sh = 2 * (15 - bt);
		avail = (int)((uint64_t)(bt + 49) < t + 1 ? (uint64_t)(bt + 49) : t + 1);
	} else {
		const int64_t u = (ix.l_pac << 1) - 1 - (int64_t)t;
		const int a0 = (int)(u & 15);
		s0 = ~text_fwd_word(w0); s1 = ~text_fwd_word(w1); s2 = ~text_fwd_word(w2); s3 = ~text_fwd_word(w3);
		sh = 2 * a0;
		avail = (int)(64 - a0 < ix.l_pac - u ? 64 - a0 : ix.l_pac - u); // down to the strand boundary; the text goes on below it
	}
	const uint32_t z0 = text_funnel(s1, s0, sh), z1 = text_funnel(s2, s1, sh);
	(void)s3;
	// the read downwards from i: position 0 of the stream is q[i]
	const uint32_t r0 = nib_reverse(qw0), r1 = nib_reverse(qw1), r2 = nib_reverse(qw2), r3 = nib_reverse(qw3);
	const uint32_t a = nib_squeeze(r0) | nib_squeeze(r1) << 16, b = nib_squeeze(r2) | nib_squeeze(r3) << 16;
	const uint32_t fa = nib_squeeze(r0 >> 2) | nib_squeeze(r1 >> 2) << 16, fb = nib_squeeze(r2 >> 2) | nib_squeeze(r3 >> 2) << 16;
	const int rs = 2 * (7 - (i & 7));
	const uint32_t y0 = text_funnel(b, a, rs), y1 = b >> rs, f0 = text_funnel(fb, fa, rs), f1 = fb >> rs;
	const uint32_t d0 = (z0 ^ y0) | f0, d1 = ((z1 ^ y1) | f1) | 0xffff0000u; // 16 + 8 bases
	int m = d0 ? __builtin_ctz(d0) >> 1 : 16 + (__builtin_ctz(d1) >> 1);
	const int lim = avail < 24 ? avail : 24;
	if (m > lim) m = lim;
	if (m > i + 1) m = i + 1;
	*more = m == lim && m < i + 1 && (uint64_t)m < t + 1;
	return m;
}

template <class Q> struct FwdLane {
	Q q; Biv *list; int len, i, min_intv, n; bool finished; Biv ik;
	int n_def, x0; uint32_t code; // owed list prefix (0: none), the start, the code of its first klv_k bases
	int mode; bool text_ok; uint64_t tpos; // text mode: FL_*; what the next request is about: the row k (FL_SA), the text position beside q[i] (FL_TEXT), the position whose row is asked for (FL_ISA)
	ARX_DEVI void start(const IndexView &ix, int len_, const Q &q_, int x, int min_intv_, Biv *list_)
	{
		q = q_; list = list_; len = len_; min_intv = min_intv_; n = 0; finished = false; n_def = 0; x0 = x; code = 0;
		mode = FL_NORMAL; text_ok = ix.sa40 && ix.isa40 && min_intv_ == 1; tpos = 0;
		ik = set_intv(ix, q.at(x)); ik.info = x + 1; i = x + 1;
	}
	ARX_DEVI void enter_text() { if (text_ok && ik.s == 1 && i < len && q.at(i) <= 3) { mode = FL_SA; tpos = ik.k; } } // one occurrence left and a base to try
	ARX_DEVI void finish_text() { list[n++] = ik; finished = true; mode = FL_NORMAL; }
	ARX_DEVI void consume_aux(const IndexView &ix, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) // the 16 bytes at aux_addr() of the last request
	{
		if (mode == FL_SA) {
			const uint64_t t = p40_decode(w0, w1, ik.k) + (uint64_t)(i - x0);
			if (t >= ix.seq_len) { finish_text(); return; } // the pattern ends the text: no base extends it
			tpos = t; mode = FL_TEXT;
		} else if (mode == FL_TEXT) {
			bool more;
			const int m = text_match_chunk(ix, w0, w1, w2, w3, tpos, q, i, len, &more);
			i += m; tpos += (uint64_t)m; ik.info = (uint64_t)i;
			if (!more) { tpos = ix.seq_len - tpos; mode = FL_ISA; }
		} else {
			ik.l = p40_decode(w0, w1, tpos);
			finish_text();
		}
	}
	// start(), and true if the interval of the first klv_k bases may be asked for (*code_out: its table index); the caller then hands it to take_jump()
	ARX_DEVI bool start_jump(const IndexView &ix, int len_, const Q &q_, int x, int min_intv_, Biv *list_, uint64_t *code_out)
	{
		start(ix, len_, q_, x, min_intv_, list_);
		const int K = ix.klv_k;
		if (!ix.klv || x + K > len) return false;
		uint32_t c = (uint32_t)q.at(x);
		for (int p = 1; p < K; ++p) { const int b = q.at(x + p); if (b > 3) return false; c |= (uint32_t)b << (2 * p); }
		code = c; *code_out = c;
		return true;
	}
	ARX_DEVI void take_jump(const IndexView &ix, const Biv &t)
	{
		if (t.s < (uint64_t)min_intv) return; // the walk ends inside the first K bases: base by base from the start, as set up by start()
		const int K = ix.klv_k;
		ik = t; ik.info = x0 + K; i = x0 + K; n_def = K - 1;
		enter_text();
	}
	ARX_DEVI bool advance(Biv *req, int *rc)
	{
		if (finished) return false;
		if (mode != FL_NORMAL) { req->k = tpos; *rc = -1 - mode; return true; } // (tpos holds the operand of every kind: a select between two members here would send the lane's state to scratch memory)
		if (i >= len || q.at(i) > 3) { list[n++] = ik; finished = true; return false; }
		*req = ik; *rc = 3 - q.at(i);
		return true;
	}
	ARX_DEVI void consume(const Biv &ok)
	{
		if (ok.s != ik.s) {
			list[n++] = ik;
			if (ok.s < (uint64_t)min_intv) { finished = true; return; }
		}
		ik = ok; ik.info = i + 1; ++i;
		enter_text();
	}
	ARX_DEVI int ret() const { return (int)ik.info; } // once finished: where the longest match ends = the next start of the first pass (every way the walk ends pushes ik last, bwt.c:308-316)
};

// list[0..n) of a finished forward extension becomes a task: the pool slice gets it longest first (bwt.c:322)
ARX_DEVI int seed_export(const SeedPools &P, int read, int x, int min_intv, const Biv *list, int n)
{
	const int t = P.new_task(read, x, min_intv, n);
	if (t < 0) return -1;
	Biv *dst = P.pool + P.tasks[t].off;
	for (int k = 0; k < n; ++k) dst[k] = list[n - 1 - k];
	return t;
}

// same for a task that exists already (re-seeding): reserve its slice now that the length is known
ARX_DEVI void seed_export_into(const SeedPools &P, int t, const Biv *list, int n)
{
	const int off = ARX_ATOMIC_ADD(P.cursors, 3 * n);
	if ((int64_t)off + 3 * n > P.pool_cap) { ARX_ATOMIC_OR(P.err, ERR_POOL_OVERFLOW); return; } // n stays 0: the task is skipped
	Biv *dst = P.pool + off;
	for (int k = 0; k < n; ++k) dst[k] = list[n - 1 - k];
	P.tasks[t].off = off; P.tasks[t].n = n;
}

// backward half of bwt_smem1a (bwt.c:323-349) for one task.  Most rows of a sweep hold a single interval (a unique match
// narrows to one size quickly): entry 0 of both lists therefore lives in registers only (prev0 / curr0) and such rows touch
// no list memory at all; entries 1.. go through the task's pool slice.
// Text mode for the backward sweep (see FwdLane): a row that has shrunk to ONE interval holding ONE occurrence (min_intv = 1) goes on base by
// base until the read differs from the text before the pattern, an ambiguous base or either start is reached -- and then that interval is
// the SMEM, if it is not contained in the last one found (bwt.c:326-345).  Backward extension leaves l where it is and moves k: the pattern
// q[i + 1 ...) lies at p = SA[k], after m more bases it is q[i + 1 - m ...) at p - m, whose row is ISA[p - m].
// ent: the interval; i: the next read index to try (-1: before the read).  Returns the new number of SMEMs in mem[].
template <class Q> ARX_DEVI int bwd_text_tail(const IndexView &ix, const Q &q, const Biv &ent, int i, int nm, int mem_last_start, Biv *mem)
{
	const uint64_t p = p40_load(ix.sa40, ent.k);
	int m = 0;
	while (i - m >= 0 && (uint64_t)m < p) {
		const int b = q.at(i - m);
		if (b > 3 || b != ref_base(ix, (int64_t)(p - 1 - (uint64_t)m))) break;
		++m;
	}
	const int start = i - m + 1;
	if (nm == 0 || start < mem_last_start) {
		Biv x = ent;
		if (m) x.k = p40_load(ix.isa40, p - (uint64_t)m);
		x.info |= (uint64_t)start << 32;
		mem[nm++] = x;
	}
	return nm;
}

// what k_seed_bwd_g leaves of such a sweep (hip_fm_coop.h: the interval in the task's second list, row / nm / mls in the task, flag 2): one thread each
struct KSeedBwdTail {
	IndexView ix; const uint8_t *bases; const int32_t *base_off; Biv *pool; SeedTask *tasks; int t0; const uint8_t *flag;
	ARX_DEV void operator()(int item, int) const
	{
		if (flag[item] != 2) return;
		SeedTask &k = tasks[t0 + item];
		k.nm = bwd_text_tail(ix, QBytes{bases + base_off[k.read]}, pool[k.off + k.n], k.row, k.nm, k.mls, pool + k.off + 2 * k.n);
	}
};

template <class Q> struct BwdLane {
	Q q; Biv *prev, *curr, *mem; int min_intv, i, j, c, n_prev, n_curr, nm, mem_last_start; bool finished, in_row;
	uint64_t curr_last_s; // curr[n_curr - 1].s and the start of mem[nm - 1] are kept in registers: both are looked at after every extension
	Biv prev0, curr0;
	int n_done, handed; // extensions so far; 1: the sweep stopped at a row boundary for somebody else to go on with (state in *this)
	const IndexView *tix; // text mode (bwd_text_tail) for the one-thread form: null = off
	ARX_DEVI void use_text(const IndexView &ix) { tix = ix.sa40 && ix.isa40 && min_intv == 1 ? &ix : nullptr; } // after start()
	ARX_DEVI void start(const Q &q_, const SeedTask &t, Biv *pool)
	{
		n_done = 0; handed = 0; tix = nullptr;
		q = q_; prev = pool + t.off; curr = prev + t.n; mem = curr + t.n; min_intv = t.min_intv;
		n_prev = t.n; i = t.x - 1; j = 0; c = 0; n_curr = 0; nm = 0; mem_last_start = 0; curr_last_s = 0; finished = false; in_row = false;
		prev0 = prev[0]; curr0 = Biv();
	}
	// the list entry the lane will ask for after the one it is extending now, or `dummy` when there is none (end of the row: the next
	// row starts from curr0, a register)
	ARX_DEVI bool has_next_entry() const { return !finished && in_row && j + 1 < n_prev; }
	ARX_DEVI void start_with(const Q &q_, const SeedTask &t, Biv *pool, const Biv &first) // start() with prev[0] already in hand
	{
		n_done = 0; handed = 0; tix = nullptr;
		q = q_; prev = pool + t.off; curr = prev + t.n; mem = curr + t.n; min_intv = t.min_intv;
		n_prev = t.n; i = t.x - 1; j = 0; c = 0; n_curr = 0; nm = 0; mem_last_start = 0; curr_last_s = 0; finished = false; in_row = false;
		prev0 = first; curr0 = Biv();
	}
	// budget > 0: after that many extensions the sweep stops at the next row boundary (handed = 1): a sweep over a repeat
	// can be ten times longer than the typical one, and a lane that is alone with it keeps its whole wavefront waiting
	// nx (pipelined kernel): prev[j] for j >= 1, fetched by the caller while entry j - 1 was being extended (null: read it here)
	ARX_DEVI bool advance(Biv *req, int *rc, int budget = 0) { const Biv none = Biv(); return advance_nx(req, rc, budget, false, none); }
	ARX_DEVI bool advance_nx(Biv *req, int *rc, int budget, bool have_nx, const Biv &nx)
	{
		while (!finished) {
			if (!in_row) { // backward extension by query position i (-1 = before the read)
				if (i < -1) { finished = true; break; }
				if (tix && n_prev == 1 && prev0.s == 1) { nm = bwd_text_tail(*tix, q, prev0, i, nm, mem_last_start, mem); finished = true; break; }
				if (budget > 0 && n_done >= budget) { prev[0] = prev0; handed = 1; finished = true; break; }
				c = i < 0 ? -1 : (q.at(i) < 4 ? q.at(i) : -1);
				n_curr = 0; j = 0;
				if (c < 0) { // nothing can be extended: the longest interval survives if it is not contained
					if (n_prev > 0 && (nm == 0 || i + 1 < mem_last_start)) { Biv t = prev0; t.info |= (uint64_t)(i + 1) << 32; mem[nm++] = t; mem_last_start = i + 1; }
					finished = true;
					break;
				}
				in_row = true;
			}
			if (j >= n_prev) {
				if (n_curr == 0) { finished = true; break; }
				Biv *sw = curr; curr = prev; prev = sw; n_prev = n_curr; prev0 = curr0;
				--i; in_row = false;
				continue;
			}
			if (j == 0) *req = prev0; else if (have_nx) *req = nx; else *req = prev[j];
			*rc = c;
			return true;
		}
		return false;
	}
	ARX_DEVI void consume(const Biv &req, const Biv &ok)
	{
		if (ok.s < (uint64_t)min_intv) {
			if (n_curr == 0 && (nm == 0 || i + 1 < mem_last_start)) { Biv t = req; t.info |= (uint64_t)(i + 1) << 32; mem[nm++] = t; mem_last_start = i + 1; }
		} else if (n_curr == 0 || ok.s != curr_last_s) {
			Biv t = ok; t.info = req.info;
			if (n_curr == 0) curr0 = t; else curr[n_curr] = t;
			++n_curr; curr_last_s = ok.s;
		}
		++j; ++n_done;
	}
};

// SMEMs of task t -> out[n...] (bwt.c:350, bwamem.c:125-129): the sweep found them by decreasing start, they are emitted by
// increasing start, those shorter than min_seed_len dropped
ARX_DEVI int seed_emit(const SeedTask &t, const Biv *pool, Biv *out, int n, int cap, int *overflow)
{
	const Biv *mem = pool + t.off + 2 * t.n;
	for (int k = t.nm - 1; k >= 0; --k) {
		const int slen = (int)((uint32_t)mem[k].info - (uint32_t)(mem[k].info >> 32));
		if (slen >= OPT_MIN_SEED_LEN) { if (n < cap) out[n++] = mem[k]; else *overflow = 1; }
	}
	return n;
}

// After the first pass: the read's SMEMs in call order, then one re-seeding task per SMEM that is long and occurs rarely
// (bwamem.c:131-146): start in its middle, min_intv = its occurrences + 1.  The tasks are chained in that order through
// `next`; *first2 receives the head.  Their forward extensions have not run yet (n = 0 marks that).  Returns the SMEM count.
ARX_DEV int seed_gather_pass1(const SeedPools &P, int read, int first_task, const uint8_t *q, Biv *out, int cap, int *overflow, int32_t *first2)
{
	int n = 0;
	for (int t = first_task; t >= 0; t = P.tasks[t].next) n = seed_emit(P.tasks[t], P.pool, out, n, cap, overflow);
	int head = -1, last = -1;
	for (int k = 0; k < n; ++k) {
		const Biv p = out[k];
		const int start = (int)(p.info >> 32), end = (int)(uint32_t)p.info;
		if (end - start < OPT_SPLIT_LEN || p.s > (uint64_t)OPT_SPLIT_WIDTH) continue;
		const int x = (start + end) >> 1;
		if (q[x] > 3) continue; // bwt_smem1a returns at once on an ambiguous base
		const int t = ARX_ATOMIC_ADD(P.cursors + 1, 1);
		if (t >= P.task_cap) { ARX_ATOMIC_OR(P.err, ERR_POOL_OVERFLOW); break; }
		SeedTask kx = SeedTask(); kx.read = read; kx.x = x; kx.min_intv = (int)p.s + 1; kx.off = 0; kx.n = 0; kx.nm = 0; kx.next = -1;
		P.tasks[t] = kx;
		if (last >= 0) P.tasks[last].next = t; else head = t;
		last = t;
	}
	*first2 = head;
	return n;
}
ARX_DEV int seed_gather_pass2(const SeedPools &P, int first2, Biv *out, int n, int cap, int *overflow)
{
	for (int t = first2; t >= 0; t = P.tasks[t].next) if (P.tasks[t].n > 0) n = seed_emit(P.tasks[t], P.pool, out, n, cap, overflow);
	return n;
}

// Third pass of mem_collect_intv: bwt_seed_strategy1 (bwt.c:358-379) from every position a match can start at -- the
// shortest forward match longer than min_seed_len that occurs fewer than max_mem_intv times.  It does not look at the
// results of the first two passes, so it is its own lane program (forward extensions only, next to no bookkeeping) and
// runs as its own kernel; seed_merge() joins the two interval lists.
constexpr int CAP_STRAT = 16; // a hit consumes at least min_seed_len + 1 bases: <= MAX_READ_LEN / 20 hits per read
// With a k-mer table (ix.ktab, K = ix.ktab_k <= min_seed_len): nothing a start does before its match is min_seed_len + 1 long can be
// observed -- bwt_seed_strategy1 only looks at an interval once i - x >= min_len (bwt.c:370) -- so a start whose next min_seed_len bases
// are all A/C/G/T and inside the read jumps to the interval of its first K bases (advance() asks for it with *rc = -1, req->k = the
// K-mer's code; consume_tab() takes it); a start that meets an N or the end of the read first cannot yield anything and is passed over
// exactly as the base-by-base walk would leave it (restart behind the N; end of the pass).
template <class Q> struct StratLane {
	Q q; Biv *out; int len, n, x, i, sx; bool fresh, finished; Biv ik;
	ARX_DEVI void start(int len_, const Q &q_, Biv *out_) { q = q_; out = out_; len = len_; n = 0; x = 0; i = 0; sx = 0; fresh = true; finished = false; ik = Biv(); }
	ARX_DEVI bool done() const { return finished; }
	ARX_DEVI void consume_tab(const IndexView &ix, const Biv &t) { ik = t; i = sx + ix.ktab_k; }
	ARX_DEVI bool advance(const IndexView &ix, Biv *req, int *rc)
	{
		for (;;) {
			if (fresh) {
				while (x < len && q.at(x) > 3) ++x;
				if (x >= len) { finished = true; return false; }
				ik = set_intv(ix, q.at(x)); sx = x; i = x + 1; fresh = false;
				if (ix.ktab) {
					if (sx + OPT_MIN_SEED_LEN >= len) { finished = true; return false; } // the walk would reach the end of the read (or an N, and then the end from behind it) before any match is long enough
					uint64_t code = (uint64_t)q.at(sx);
					int bad = -1;
					for (int p = 1; p <= OPT_MIN_SEED_LEN; ++p) {
						const int b = q.at(sx + p);
						if (b > 3) { bad = sx + p; break; }
						if (p < ix.ktab_k) code |= (uint64_t)b << (2 * p);
					}
					if (bad >= 0) { x = bad + 1; fresh = true; continue; }
					req->k = code; *rc = -1;
					return true;
				}
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
