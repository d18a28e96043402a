// dev_fm.h -- FM-index seeding on the device: Occ, bidirectional extension, SMEM search,
// the three seeding passes of BWA-MEM and sampled-SA locate.
// Agrees with bwt.c:53-115,169-274,289-379 and bwamem.c:114-162 of the reference.
#pragma once
#include "arx_dev.h"

namespace arx {

// Packed per-word symbol counting.  A 32-bit BWT word holds 16 symbols, MSB first.
// marks(w, c): bit 2i set iff symbol i (from the LSB side) equals c.
ARX_DEVI uint32_t sym_marks(uint32_t w, int c)
{
	uint32_t x = w ^ (0x55555555u * (uint32_t)(3 - c)); // symbols equal to c become 0b11
	return x & (x >> 1) & 0x55555555u;
}

// counts of A,C,G,T among the first n (0..128) symbols of one 64-byte block (words at blk+8..blk+15)
ARX_DEVI void block_count(const uint32_t *blk, int n, uint32_t cnt[4])
{
	cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		int nj = n - 16 * j;
		if (nj <= 0) break;
		uint32_t w = blk[8 + j];
		uint32_t keep = nj >= 16 ? 0xffffffffu : ~((1u << ((16 - nj) << 1)) - 1);
#pragma unroll
		for (int c = 0; c < 4; ++c) cnt[c] += __builtin_popcount(sym_marks(w, c) & keep);
	}
}

// bwt_occ4 (bwt.c:169-187): counts in B[0..k] of the $-removed BWT.  Touches exactly one 64-byte block.
ARX_DEVI void occ4(const IndexView &ix, uint64_t k, uint64_t cnt[4])
{
	if (k == (uint64_t)-1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
	k -= (k >= ix.primary);
	const uint32_t *blk = ix.bwt + ((k >> 7) << 4);
	const uint64_t *cum = (const uint64_t *)blk;
	uint32_t c4[4];
	block_count(blk, (int)(k & 127) + 1, c4);
	cnt[0] = cum[0] + c4[0]; cnt[1] = cum[1] + c4[1]; cnt[2] = cum[2] + c4[2]; cnt[3] = cum[3] + c4[3];
}

// bwt_occ (bwt.c:107-130) for one symbol
ARX_DEVI uint64_t occ1(const IndexView &ix, uint64_t k, int c)
{
	if (k == ix.seq_len) return ix.L2[c + 1] - ix.L2[c];
	if (k == (uint64_t)-1) return 0;
	k -= (k >= ix.primary);
	const uint32_t *blk = ix.bwt + ((k >> 7) << 4);
	int n = (int)(k & 127) + 1;
	uint32_t cnt = 0;
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		int nj = n - 16 * j;
		if (nj <= 0) break;
		uint32_t keep = nj >= 16 ? 0xffffffffu : ~((1u << ((16 - nj) << 1)) - 1);
		cnt += __builtin_popcount(sym_marks(blk[8 + j], c) & keep);
	}
	return ((const uint64_t *)blk)[c] + cnt;
}

// bwt_extend (bwt.c:262-274), returning only the child for symbol c -- the SMEM search never looks at the other three.
// is_back = 1 extends x[0] (k) with the cumulative sizes fixing x[1] (l); is_back = 0 the other way round.
ARX_DEVI Biv extend1(const IndexView &ix, const Biv &ik, int is_back, int c)
{
	uint64_t a = is_back ? ik.k : ik.l, b = is_back ? ik.l : ik.k;
	uint64_t tk[4], tl[4];
	occ4(ix, a - 1, tk);
	occ4(ix, a - 1 + ik.s, tl);
	uint64_t s3 = tl[3] - tk[3], s2 = tl[2] - tk[2], s1 = tl[1] - tk[1];
	uint64_t x = b + (a <= ix.primary && a + ik.s - 1 >= ix.primary); // position of child 3 on the other strand
	if (c < 3) x += s3;
	if (c < 2) x += s2;
	if (c < 1) x += s1;
	Biv ok;
	uint64_t na = ix.L2[c] + 1 + tk[c];
	ok.s = tl[c] - tk[c];
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

// bwt_sa (bwt.c:86-96) via bwt_invPsi (bwt.c:53-59): LF steps until a sampled row
ARX_DEVI uint64_t sa_lookup(const IndexView &ix, uint64_t k)
{
	uint64_t sa = 0, mask = (uint64_t)ix.sa_intv - 1;
	while (k & mask) {
		uint64_t x = k - (k > ix.primary);
		int c = ix.bwt[((x >> 7) << 4) + 8 + ((x & 127) >> 4)] >> ((~x & 15) << 1) & 3;
		++sa;
		k = (k == ix.primary) ? 0 : ix.L2[c] + occ1(ix, k, c);
	}
	return sa + ix.sa[k / (uint64_t)ix.sa_intv];
}

// Scratch for one SMEM search: two interval lists of up to len+1 entries each plus the per-call result list.
struct SmemScratch { Biv *v0, *v1, *mem; };

// bwt_smem1a with max_intv = 0 (bwt.c:289-351).  Returns the end of the longest match from x; *n_mem SMEMs in sc.mem, sorted by start.
ARX_DEV int smem1(const IndexView &ix, int len, const uint8_t *q, int x, int min_intv, const SmemScratch &sc, int *n_mem)
{
	Biv *prev = sc.v0, *curr = sc.v1, *sw, *mem = sc.mem;
	int i, j, c, ret, n_curr = 0, n_prev, nm = 0;
	*n_mem = 0;
	if (q[x] > 3) return x + 1;
	if (min_intv < 1) min_intv = 1;
	Biv ik = set_intv(ix, q[x]), ok;
	ik.info = x + 1;
	for (i = x + 1; i < len; ++i) { // forward: remember the interval each time its size changes
		if (q[i] < 4) {
			ok = extend1(ix, ik, 0, 3 - q[i]);
			if (ok.s != ik.s) {
				curr[n_curr++] = ik;
				if (ok.s < (uint64_t)min_intv) break;
			}
			ik = ok; ik.info = i + 1;
		} else { curr[n_curr++] = ik; break; }
	}
	if (i == len) curr[n_curr++] = ik;
	for (j = 0; j < n_curr >> 1; ++j) { Biv t = curr[n_curr - 1 - j]; curr[n_curr - 1 - j] = curr[j]; curr[j] = t; } // longest first
	ret = (int)curr[0].info;
	sw = curr; curr = prev; prev = sw; n_prev = n_curr;
	for (i = x - 1; i >= -1; --i) { // backward: keep what cannot be extended and is not contained in a longer match
		c = i < 0 ? -1 : (q[i] < 4 ? q[i] : -1);
		for (j = 0, n_curr = 0; j < n_prev; ++j) {
			const Biv p = prev[j];
			if (c >= 0) ok = extend1(ix, p, 1, c);
			if (c < 0 || ok.s < (uint64_t)min_intv) {
				if (n_curr == 0) {
					if (nm == 0 || (uint64_t)(i + 1) < mem[nm - 1].info >> 32) {
						ik = p; ik.info |= (uint64_t)(i + 1) << 32;
						mem[nm++] = ik;
					}
				}
			} else if (n_curr == 0 || ok.s != curr[n_curr - 1].s) {
				ok.info = p.info;
				curr[n_curr++] = ok;
			}
		}
		if (n_curr == 0) break;
		sw = curr; curr = prev; prev = sw; n_prev = n_curr;
	}
	for (j = 0; j < nm >> 1; ++j) { Biv t = mem[nm - 1 - j]; mem[nm - 1 - j] = mem[j]; mem[j] = t; }
	*n_mem = nm;
	return ret;
}

// bwt_seed_strategy1 (bwt.c:358-379): shortest forward match of >= min_len+1 bases occurring < max_intv times
ARX_DEV int seed_strategy1(const IndexView &ix, int len, const uint8_t *q, int x, int min_len, int max_intv, Biv *mem)
{
	mem->k = mem->l = mem->s = mem->info = 0;
	if (q[x] > 3) return x + 1;
	Biv ik = set_intv(ix, q[x]);
	for (int i = x + 1; i < len; ++i) {
		if (q[i] < 4) {
			Biv ok = extend1(ix, ik, 0, 3 - q[i]);
			if (ok.s < (uint64_t)max_intv && i - x >= min_len) {
				*mem = ok;
				mem->info = (uint64_t)x << 32 | (uint32_t)(i + 1);
				return i + 1;
			}
			ik = ok;
		} else return i + 1;
	}
	return len;
}

// mem_collect_intv (bwamem.c:114-162): three seeding passes, then sort by info.  Entries with equal info describe the same
// query substring and hence the same bi-interval, so any sort reproduces ks_introsort's result.
// Returns the number of intervals written to out (capacity cap); sets *overflow when more were found.
ARX_DEV int collect_intv(const IndexView &ix, int len, const uint8_t *seq, const SmemScratch &sc, Biv *out, int cap, int *overflow)
{
	int n = 0, x = 0, nm;
	while (x < len) {
		if (seq[x] < 4) {
			x = smem1(ix, len, seq, x, 1, sc, &nm);
			for (int i = 0; i < nm; ++i) {
				int slen = (int)((uint32_t)sc.mem[i].info - (uint32_t)(sc.mem[i].info >> 32));
				if (slen >= OPT_MIN_SEED_LEN) { if (n < cap) out[n++] = sc.mem[i]; else *overflow = 1; }
			}
		} else ++x;
	}
	int old_n = n;
	for (int k = 0; k < old_n; ++k) { // re-seed from the middle of long SMEMs that occur rarely
		const Biv p = out[k];
		int start = (int)(p.info >> 32), end = (int)(uint32_t)p.info;
		if (end - start < OPT_SPLIT_LEN || p.s > (uint64_t)OPT_SPLIT_WIDTH) continue;
		smem1(ix, len, seq, (start + end) >> 1, (int)p.s + 1, sc, &nm);
		for (int i = 0; i < nm; ++i) {
			int slen = (int)((uint32_t)sc.mem[i].info - (uint32_t)(sc.mem[i].info >> 32));
			if (slen >= OPT_MIN_SEED_LEN) { if (n < cap) out[n++] = sc.mem[i]; else *overflow = 1; }
		}
	}
	x = 0;
	while (x < len) { // LAST-like pass
		if (seq[x] < 4) {
			Biv m;
			x = seed_strategy1(ix, len, seq, x, OPT_MIN_SEED_LEN, OPT_MAX_MEM_INTV, &m);
			if (m.s > 0) { if (n < cap) out[n++] = m; else *overflow = 1; }
		} else ++x;
	}
	for (int i = 1; i < n; ++i) { // insertion sort by info
		Biv t = out[i];
		int j = i;
		while (j > 0 && out[j - 1].info > t.info) { out[j] = out[j - 1]; --j; }
		out[j] = t;
	}
	return n;
}

} // namespace arx
