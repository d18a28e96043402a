// dev_sw.h -- the three Smith-Waterman flavours of the path, one task per thread, max-plus integer DP (no MFMA).
//   ext2_task    : banded affine-gap extension with z-drop           == ksw_extend2  (ksw.c:380-479)
//   u8_align     : 16-lane striped u8 local SW, forward + reverse    == ksw_align2/ksw_u8 (ksw.c:111-230,343-365)
//   global2_task : banded global alignment with traceback            == ksw_global2  (ksw.c:504-606)
// DP rows live in LDS as [column][lane] (one 32-bit word per column per lane): lane l of a wave always hits bank l,
// so the accesses stay conflict-free however far the lanes' column indices drift apart.
#pragma once
#include "arx_dev.h"

namespace arx {

// ------------------------------------------------------------------------------------------------
// Banded extension.  Row word = h (14 bits) | e (14 bits) << 14 | query base (3 bits) << 28.
// ------------------------------------------------------------------------------------------------
struct ExtTask {
	int64_t tpos;          // doubled coordinate of the first target base
	int32_t owner;         // read (or slot) the result belongs to
	int32_t qoff;          // index of the first query base in the batch's base array
	int32_t qlen, tlen;
	int32_t qdir, tdir;    // +1 / -1: left extensions walk both sequences backwards
	int32_t w, h0;
};
struct ExtRes { int32_t score, qle, tle, gtle, gscore, max_off; };

ARX_DEVI uint32_t eh_pack(int h, int e, uint32_t q) { return (uint32_t)h | (uint32_t)e << 14 | q << 28; }

// row: this thread's first word, consecutive columns are `stride` words apart; needs qlen+1 columns
ARX_DEV ExtRes ext2_task(const IndexView &ix, const uint8_t *bases, const ExtTask &t, uint32_t *row, int stride)
{
	const int o_del = OPT_O_DEL, e_del = OPT_E_DEL, o_ins = OPT_O_INS, e_ins = OPT_E_INS, oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	const int end_bonus = OPT_PEN_CLIP5, zdrop = OPT_ZDROP; // pen_clip5 == pen_clip3 == 5
	const int qlen = t.qlen, tlen = t.tlen, h0 = t.h0;
	int i, j, beg, end, max, max_i, max_j, max_ie, gscore, max_off, w = t.w;
	// first row: h0, then one gap open and extensions while positive (ksw.c:395-397)
	{
		int h = h0;
		for (j = 0; j <= qlen; ++j) {
			uint32_t qb = j < qlen ? bases[t.qoff + j * t.qdir] : 0;
			if (j == 1) h = h0 > oe_ins ? h0 - oe_ins : 0;
			else if (j >= 2) h = h > e_ins ? h - e_ins : 0;
			row[j * stride] = eh_pack(h, 0, qb);
		}
	}
	// the band cannot be wider than the longest gap the scores allow (ksw.c:402-407); with a=1,o=6,e=1 both bounds are qlen+end_bonus-5
	{
		int mg = qlen + end_bonus - 5;
		mg = mg > 1 ? mg : 1;
		w = w < mg ? w : mg;
	}
	max = h0; max_i = max_j = -1; max_ie = -1; gscore = -1; max_off = 0;
	beg = 0; end = qlen;
	for (i = 0; i < tlen; ++i) {
		int f = 0, h1, m = 0, mj = -1, tt;
		const int tb = ref_base(ix, t.tpos + (int64_t)i * t.tdir);
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; } else h1 = 0;
		for (j = beg; j < end; ++j) {
			uint32_t wd = row[j * stride];
			int M = wd & 0x3fff, e = (wd >> 14) & 0x3fff, h;
			uint32_t qb = wd >> 28;
			int hprev = h1;
			M = M ? M + sc_mat(tb, (int)qb) : 0;
			h = M > e ? M : e;
			h = h > f ? h : f;
			h1 = h;
			mj = m > h ? mj : j;
			m = m > h ? m : h;
			tt = M - oe_del; tt = tt > 0 ? tt : 0;
			e -= e_del; e = e > tt ? e : tt;
			row[j * stride] = eh_pack(hprev, e, qb);
			tt = M - oe_ins; tt = tt > 0 ? tt : 0;
			f -= e_ins; f = f > tt ? f : tt;
		}
		row[end * stride] = eh_pack(h1, 0, row[end * stride] >> 28);
		if (j == qlen) {
			max_ie = gscore > h1 ? max_ie : i;
			gscore = gscore > h1 ? gscore : h1;
		}
		if (m == 0) break;
		if (m > max) {
			max = m; max_i = i; max_j = mj;
			max_off = max_off > iabs(mj - i) ? max_off : iabs(mj - i);
		} else if (zdrop > 0) {
			if (i - max_i > mj - max_j) { if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break; }
			else { if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break; }
		}
		for (j = beg; j < end && (row[j * stride] & 0x0fffffffu) == 0; ++j) {}
		beg = j;
		for (j = end; j >= beg && (row[j * stride] & 0x0fffffffu) == 0; --j) {}
		end = j + 2 < qlen ? j + 2 : qlen;
	}
	ExtRes r;
	r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
	return r;
}

// ------------------------------------------------------------------------------------------------
// Striped u8 local SW.  The 16-way striping of the SSE2 original is observable in the results (E is fed the
// pre-lazy-F H, F restarts at every stripe boundary), so the cell order is kept: query position j + l*slen is
// "vector j, lane l".  Row word (index j*16+l) = Ha | Hb<<8 | E<<16 | Hmax<<24; Ha/Hb swap roles every row.
// Scores never reach the u8 ceiling for reads under 250 bp, but saturation is emulated all the same.
// ------------------------------------------------------------------------------------------------
struct U8Res { int32_t score, te, qe, score2, te2, tb, qb; };

struct PrefixRevView { // element i of a sequence whose first n_rev elements are read back to front (ksw.c:357 revseq)
	const uint8_t *p; int n_rev;
	ARX_DEVI int operator[](int i) const { return i < n_rev ? p[n_rev - 1 - i] : p[i]; }
};

ARX_DEVI int sat_add8(int a, int b) { int s = a + b; return s > 255 ? 255 : s; }
ARX_DEVI int sat_sub8(int a, int b) { return a > b ? a - b : 0; }

// one ksw_u8 pass; rowmax (>= tlen bytes) records the per-row maxima for the score2/te2 scan.
// i16 = true: the same pass as ksw_i16 runs it (ksw.c:232-334; ksw_align2 takes it when the caller did not set KSW_XBYTE, i.e. for mates
// of 250 bases and more, bwamem_pair.c:150): EIGHT stripes instead of sixteen, no bias, no 255 ceiling.  Everything else is the same
// arithmetic -- adds_epi16(h, s) followed by the maxima with e, f >= 0 is max(h + s, 0) like the biased byte form, the gap states use the
// same unsigned saturating subtractions -- and for reads of up to 255 bases every value still fits the byte fields of the row word.
// exact_f = true (tests): the row's F by the plain recurrence over the query positions in their order -- what hip_sw_coop.h's scan computes --
// instead of the striped main pass + lazy loop, and E from the final H: the H values, and with them every result, are the same (the host test
// double checks that against the restatement on gapped, low-complexity and random pairs: tests/test_sw_prefilter.py).
ARX_DEV U8Res u8_pass(const PrefixRevView &q, int qlen, const PrefixRevView &tg, int tlen, int xtra, uint32_t *row, int stride, uint8_t *rowmax, bool i16 = false, bool exact_f = false)
{
	const int shift = 4, qmax = 1; // ksw_qinit: shift = 256 - (uint8_t)min(mat) = 4, max = 1
	const int oe_del = OPT_O_DEL + OPT_E_DEL, e_del = OPT_E_DEL, oe_ins = OPT_O_INS + OPT_E_INS, e_ins = OPT_E_INS;
	const int NS = i16 ? 8 : 16;
	const int slen = (qlen + NS - 1) / NS, n16 = slen * NS;
	const int minsc = (xtra & KSW_XSUBO) ? (xtra & 0xffff) : 0x10000, endsc = (xtra & KSW_XSTOP) ? (xtra & 0xffff) : 0x10000;
	int i, j, l, te = -1, gmax = 0, cur = 0, rows = 0;
	for (i = 0; i < n16; ++i) row[i * stride] = 0;
	for (i = 0; i < tlen; ++i) {
		const int tb = tg[i];
		const int sh_prev = cur * 8, sh_new = (cur ^ 1) * 8;
		int imax = 0;
		int fl[16]; // F carried across the lazy loop, one per stripe
		if (exact_f) {
			int f = 0, h = 0; // H(i-1, k-1) for k = 0: nothing
			for (int k = 0; k < n16; ++k) {
				const int ll = k / slen, jj = k % slen;
				uint32_t wd = row[(jj * NS + ll) * stride];
				int e = (wd >> 16) & 0xff, tt;
				const int sc = (k >= qlen ? 0 : sc_mat(tb, q[k])) + shift;
				int hh = i16 ? h + sc - shift : sat_sub8(sat_add8(h, sc), shift);
				hh = hh > e ? hh : e;
				hh = hh > f ? hh : f;
				imax = imax > hh ? imax : hh;
				h = (wd >> sh_prev) & 0xff;
				e = sat_sub8(e, e_del); tt = sat_sub8(hh, oe_del); e = e > tt ? e : tt;
				f = sat_sub8(f, e_ins); tt = sat_sub8(hh, oe_ins); f = f > tt ? f : tt;
				row[(jj * NS + ll) * stride] = (wd & ~(0xffu << sh_new) & ~(0xffu << 16)) | (uint32_t)hh << sh_new | (uint32_t)e << 16;
			}
		} else {
		// main pass: each stripe is an independent chain over its slen consecutive query positions
		for (l = 0; l < NS; ++l) {
			int f = 0, mx = 0;
			int h = l == 0 ? 0 : (int)((row[((slen - 1) * NS + l - 1) * stride] >> sh_prev) & 0xff); // H(i-1) of the previous stripe's last cell
			for (j = 0; j < slen; ++j) {
				const int k = j + l * slen;
				uint32_t wd = row[(j * NS + l) * stride];
				int e = (wd >> 16) & 0xff, tt;
				int sc = (k >= qlen ? 0 : sc_mat(tb, q[k])) + shift;
				int hh = i16 ? h + sc - shift : sat_sub8(sat_add8(h, sc), shift); // (16-bit lanes do not saturate at 255: a 255-base mate can score 255)
				hh = hh > e ? hh : e;
				hh = hh > f ? hh : f;
				mx = mx > hh ? mx : hh;
				h = (wd >> sh_prev) & 0xff; // H(i-1, this cell) feeds the next cell's diagonal
				e = sat_sub8(e, e_del); tt = sat_sub8(hh, oe_del); e = e > tt ? e : tt;
				f = sat_sub8(f, e_ins); tt = sat_sub8(hh, oe_ins); f = f > tt ? f : tt;
				wd = (wd & ~(0xffu << sh_new) & ~(0xffu << 16)) | (uint32_t)hh << sh_new | (uint32_t)e << 16;
				row[(j * NS + l) * stride] = wd;
			}
			fl[l] = f;
			imax = imax > mx ? imax : mx;
		}
		// lazy-F (ksw.c:177-189): shift F one stripe up, sweep, stop as soon as no stripe can still raise an H
		{
			bool done = false;
			for (int k2 = 0; k2 < 16 && !done; ++k2) {
				for (l = NS - 1; l > 0; --l) fl[l] = fl[l - 1];
				fl[0] = 0;
				for (j = 0; j < slen; ++j) {
					bool all = true;
					for (l = 0; l < NS; ++l) {
						uint32_t wd = row[(j * NS + l) * stride];
						int hh = (wd >> sh_new) & 0xff;
						if (fl[l] > hh) { hh = fl[l]; row[(j * NS + l) * stride] = (wd & ~(0xffu << sh_new)) | (uint32_t)hh << sh_new; }
						hh = sat_sub8(hh, oe_ins);
						fl[l] = sat_sub8(fl[l], e_ins);
						if (fl[l] > hh) all = false;
					}
					if (all) { done = true; break; }
				}
			}
		}
		}
		rowmax[i] = (uint8_t)imax; ++rows;
		if (imax > gmax) {
			gmax = imax; te = i;
			for (j = 0; j < n16; ++j) { uint32_t wd = row[j * stride]; row[j * stride] = (wd & 0x00ffffffu) | ((wd >> sh_new) & 0xff) << 24; }
			if ((!i16 && gmax + shift >= 255) || gmax >= endsc) break;
		}
		cur ^= 1;
	}
	U8Res r;
	r.score = (i16 || gmax + shift < 255) ? gmax : 255; r.te = te; r.qe = -1; r.score2 = -1; r.te2 = -1; r.tb = -1; r.qb = -1;
	if (i16 || r.score != 255) {
		int mx = -1;
		for (i = 0; i < n16; ++i) { // smallest query index attaining the maximum of the saved row (ksw.c:212-216)
			int v = row[i * stride] >> 24, qpos = i / NS + i % NS * slen;
			if (v > mx) { mx = v; r.qe = qpos; }
			else if (v == mx && qpos < r.qe) r.qe = qpos;
		}
		if (minsc < 0x10000) { // replay of the b[] list (ksw.c:192-200,218-226): runs of consecutive rows >= minsc keep their best row
			const int d = (r.score + qmax - 1) / qmax, low = te - d, high = te + d;
			int bi = -1, bs = -1; // current last entry {row, score}
			for (i = 0; i < rows; ++i) {
				int im = rowmax[i];
				if (im < minsc) continue;
				if (bi < 0 || bi + 1 != i) { // append: the previous entry is final
					if (bi >= 0 && (bi < low || bi > high) && bs > r.score2) { r.score2 = bs; r.te2 = bi; }
					bi = i; bs = im;
				} else if (bs < im) { bi = i; bs = im; } // modify the last entry
			}
			if (bi >= 0 && (bi < low || bi > high) && bs > r.score2) { r.score2 = bs; r.te2 = bi; }
		}
	}
	return r;
}

// ---- An exact pre-filter for the rescue alignment.  mem_matesw only looks at the alignment when its score reaches
// min_seed_len = 19 (bwamem_pair.c:153); below that the call leaves no trace.  Most rescue windows sit next to a repeat copy
// of the other read and hold nothing, so it pays to prove "score < 19" without the DP.  With match +1, mismatch -4, gaps -(6+g)
// and no N (bwa.c:109-118, bwamem.c:48-84) take any local alignment of score S with G gaps: its G+1 gap-free segments have
// sum(M_j - 4 X_j) = S + sum(6 + g_i) >= S + 7G (M_j matching, X_j mismatching columns).  A segment's matches fall into at most
// X_j + 1 runs and a run of r columns holds r - 4 exact 5-mer matches on its diagonal, so the segment holds K_j >= M_j - 4 X_j - 4
// of them and the alignment sum K_j >= S + 7G - 4(G+1) = S - 4 + 3G, on at most G+1 distinct diagonals.  With c_d the number
// of 5-mer matches on the whole diagonal d that gives sum over those diagonals of (c_d - 3) >= S - 7, hence
//     S >= 19  implies  sum over all diagonals of max(0, c_d - 3) >= 12.
// ksw_u8's score is the score of some alignment (every H it forms comes from a path), so it is below 19 whenever that sum is
// below 12.  A read with an N is never filtered (an N costs 1 and breaks runs).  Returns true when the DP has to run.
constexpr int SWF_K = 5, SWF_NEED = 12, SWF_FREE = 3;
ARX_DEVI bool sw_prefilter_serial(const uint8_t *q, int qlen, const uint8_t *t, int tlen)
{
	uint8_t head[1024], nxt[256], cnt[1056];
	if (qlen > 255 || tlen > 800) return true;
	for (int i = 0; i < qlen; ++i) if (q[i] > 3) return true;
	for (int i = 0; i < 1024; ++i) head[i] = 0xff;
	for (int i = 0; i < qlen + tlen; ++i) cnt[i] = 0;
	for (int i = 0; i + SWF_K <= qlen; ++i) {
		int code = 0;
		for (int x = 0; x < SWF_K; ++x) code |= q[i + x] << (2 * x);
		nxt[i] = head[code]; head[code] = (uint8_t)i;
	}
	for (int j = 0; j + SWF_K <= tlen; ++j) {
		int code = 0;
		for (int x = 0; x < SWF_K; ++x) code |= t[j + x] << (2 * x);
		for (int i = head[code]; i != 0xff; i = nxt[i]) ++cnt[j - i + qlen - 1];
	}
	int s = 0;
	for (int d = 0; d < qlen + tlen; ++d) if (cnt[d] > SWF_FREE) s += cnt[d] - SWF_FREE;
	return s >= SWF_NEED;
}
#ifndef ARX_SW_FILTER_CHECK
#define ARX_SW_FILTER_CHECK(pass, score) ((void)0) // the host test double checks the filter against the DP here
#endif

// ksw_align2 (ksw.c:343-365) with XBYTE: forward pass, then a pass over the reversed prefixes to find the start
ARX_DEV U8Res u8_align(const uint8_t *query, int qlen, const uint8_t *target, int tlen, int xtra, uint32_t *row, int stride, uint8_t *rowmax, bool exact_f = false)
{
	PrefixRevView q{query, 0}, t{target, 0};
	const bool i16 = !(xtra & KSW_XBYTE); // ksw_align2's choice of element size (ksw.c:350-353); the second pass keeps it
	U8Res r = u8_pass(q, qlen, t, tlen, xtra, row, stride, rowmax, i16, exact_f);
	if ((xtra & KSW_XSTART) == 0 || ((xtra & KSW_XSUBO) && r.score < (xtra & 0xffff))) return r;
	PrefixRevView q2{query, r.qe + 1}, t2{target, r.te + 1};
	U8Res rr = u8_pass(q2, r.qe + 1, t2, tlen, KSW_XSTOP | r.score, row, stride, rowmax, i16, exact_f);
	if (r.score == rr.score) { r.tb = r.te - rr.te; r.qb = r.qe - rr.qe; }
	return r;
}

// ------------------------------------------------------------------------------------------------
// Banded global alignment with traceback, plus the CIGAR front end of bwa_gen_cigar2 (bwa.c:121-207).
// ------------------------------------------------------------------------------------------------
constexpr int NW_MINUS_INF = -0x40000000;

struct SegView { // query segment [qb,qe) / reference segment [rb,re), both read back to front on the reverse strand (bwa.c:135-140)
	const uint8_t *q; int qb, qe; int64_t rb, re; bool rev;
	ARX_DEVI int qat(int i) const { return rev ? q[qe - 1 - i] : q[qb + i]; }
	ARX_DEVI int tat(const IndexView &ix, int i) const { return ref_base(ix, rev ? re - 1 - i : rb + i); }
};

ARX_DEVI int push_cigar(uint32_t *cg, int n, int cap, int op, int len) // ksw.c:487-497; returns the new length (> cap on overflow)
{
	if (n == 0 || op != (int)(cg[n - 1] & 0xf)) { if (n < cap) cg[n] = (uint32_t)len << 4 | op; return n + 1; }
	cg[n - 1] += (uint32_t)len << 4;
	return n;
}

// eh: 2*(qlen+1) int32 (h then e); z: n_col*tlen bytes (may be null for score only); cigar built reversed then flipped.
ARX_DEV int global2_task(const IndexView &ix, const SegView &sv, int qlen, int tlen, int w, int32_t *eh, uint8_t *z, uint32_t *cg, int cap, int *n_cigar)
{
	const int o_del = OPT_O_DEL, e_del = OPT_E_DEL, o_ins = OPT_O_INS, e_ins = OPT_E_INS, oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	int32_t *H = eh, *E = eh + qlen + 1;
	const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
	int i, j;
	H[0] = 0; E[0] = NW_MINUS_INF;
	for (j = 1; j <= qlen && j <= w; ++j) { H[j] = -(o_ins + e_ins * j); E[j] = NW_MINUS_INF; }
	for (; j <= qlen; ++j) H[j] = E[j] = NW_MINUS_INF;
	for (i = 0; i < tlen; ++i) {
		int32_t f = NW_MINUS_INF, h1, beg, end, t;
		const int tb = sv.tat(ix, i);
		uint8_t *zi = z ? z + (size_t)i * n_col : 0;
		beg = i > w ? i - w : 0;
		end = i + w + 1 < qlen ? i + w + 1 : qlen;
		h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : NW_MINUS_INF;
		for (j = beg; j < end; ++j) {
			int32_t h, m = H[j], e = E[j];
			uint8_t d;
			H[j] = h1;
			m += sc_mat(tb, sv.qat(j));
			d = m >= e ? 0 : 1;
			h = m >= e ? m : e;
			d = h >= f ? d : 2;
			h = h >= f ? h : f;
			h1 = h;
			t = m - oe_del;
			e -= e_del;
			d |= e > t ? 1 << 2 : 0;
			e = e > t ? e : t;
			E[j] = e;
			t = m - oe_ins;
			f -= e_ins;
			d |= f > t ? 2 << 4 : 0;
			f = f > t ? f : t;
			if (zi) zi[j - beg] = d;
		}
		H[end] = h1; E[end] = NW_MINUS_INF;
	}
	int score = H[qlen];
	if (z) { // backtrack (ksw.c:588-603): ties prefer M over the gap states
		int n = 0, which = 0, k;
		i = tlen - 1; k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
		while (i >= 0 && k >= 0) {
			which = z[(size_t)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
			if (which == 0) { n = push_cigar(cg, n, cap, 0, 1); --i; --k; }
			else if (which == 1) { n = push_cigar(cg, n, cap, 2, 1); --i; }
			else { n = push_cigar(cg, n, cap, 1, 1); --k; }
		}
		if (i >= 0) n = push_cigar(cg, n, cap, 2, i + 1);
		if (k >= 0) n = push_cigar(cg, n, cap, 1, k + 1);
		if (n <= cap) for (i = 0; i < n >> 1; ++i) { uint32_t x = cg[i]; cg[i] = cg[n - 1 - i]; cg[n - 1 - i] = x; }
		*n_cigar = n;
	}
	return score;
}

// Score and mismatch count of a gap-free region: the l pairs (q[j], T[rb + j]), T = forward strand + its reverse complement, whichever way
// bwa_gen_cigar2 walks them (bwa.c:135-149, 169-199).  Four pairs per step: the read's four bytes as one (unaligned) word, squeezed to 2-bit
// codes with the ambiguity bit kept aside; the packed strand's byte slides along (one new byte per step: upwards on the forward strand,
// downwards and complemented on the reverse strand).  The target never holds an ambiguous base (the packed strand has two bits per base),
// so a pair scores OPT_A (equal), -1 (read base > 3) or -OPT_B.
ARX_DEVI void gapfree_counts(const IndexView &ix, const uint8_t *q, int64_t rb, int l, int *score, int *n_mm_out)
{
	int n_mm = 0, n_amb = 0, j = 0;
	const bool rev = rb >= ix.l_pac;
	if (l >= 4) {
		const int64_t p0 = rev ? (ix.l_pac << 1) - 1 - rb : rb; // position on the packed strand of pair 0; pair j: p0 + j (forward), p0 - j (reverse)
		const int o = (int)(p0 & 3);
		const int64_t b_lim = rev ? (p0 - (l - 1)) >> 2 : (p0 + (l - 1)) >> 2; // last byte the region touches
		int64_t b = p0 >> 2;
		uint32_t cur = ix.pac[b], nxt;
		if (!rev) { b = b + 1 < b_lim ? b + 1 : b_lim; nxt = ix.pac[b]; }
		else { b = b - 1 > b_lim ? b - 1 : b_lim; nxt = ix.pac[b]; }
#if defined(__HIP_DEVICE_COMPILE__) && defined(ARX_GAPFREE_UNROLL) // 4 / 8 measured: no gain (profiles/r03/README.md); the compiler's own choice by default
#pragma unroll ARX_GAPFREE_UNROLL
#endif
		for (; j + 4 <= l; j += 4) {
			uint32_t qw;
			__builtin_memcpy(&qw, q + j, 4);
			const uint32_t amb = ((qw >> 2) & 1u) | ((qw >> 8) & 4u) | ((qw >> 14) & 0x10u) | ((qw >> 20) & 0x40u); // bit 2m: read base m is > 3
			uint32_t t8, q8;
			if (!rev) { // bases p .. p + 3 from the top down in (cur, nxt): pair m in bits 7 - 2m, 6 - 2m
				t8 = (((cur << 8) | nxt) >> (8 - 2 * o)) & 0xffu;
				q8 = ((qw & 3u) << 6) | ((qw >> 4) & 0x30u) | ((qw >> 14) & 0xcu) | ((qw >> 24) & 3u);
			} else { // bases p, p - 1, .. p - 3 complemented: pair m in bits 2m + 1, 2m
				t8 = ((((nxt << 8) | cur) >> (6 - 2 * o)) & 0xffu) ^ 0xffu;
				q8 = (qw & 3u) | ((qw >> 6) & 0xcu) | ((qw >> 12) & 0x30u) | ((qw >> 18) & 0xc0u);
			}
			const uint32_t x = t8 ^ q8;
			const uint32_t d = ((x | (x >> 1)) & 0x55u) | (rev ? amb : ((amb & 1u) << 6) | ((amb & 4u) << 2) | ((amb >> 2) & 4u) | ((amb >> 6) & 1u));
			n_mm += __builtin_popcount(d);
			n_amb += __builtin_popcount(amb);
			cur = nxt;
			if (!rev) { b = b + 1 < b_lim ? b + 1 : b_lim; } else { b = b - 1 > b_lim ? b - 1 : b_lim; }
			nxt = ix.pac[b];
		}
	}
	for (; j < l; ++j) { const int t = ref_base(ix, rb + j), c = q[j]; n_mm += c != t; n_amb += c > 3; }
	*n_mm_out = n_mm;
	*score = (l - n_mm) * OPT_A - (n_mm - n_amb) * OPT_B - n_amb;
}

// bwa_gen_cigar2 (bwa.c:121-207) for an in-range region on one strand.  With want_cigar = false only the score is computed
// (mem_patch_reg's use).  Returns false when the region is rejected (score untouched).
ARX_DEV bool gen_cigar2(const IndexView &ix, int w_, const uint8_t *query, int qb, int qe, int64_t rb, int64_t re,
                        int32_t *eh, uint8_t *z, bool want_cigar, uint32_t *cg, int cap, int *score, int *n_cigar, int *NM)
{
	const int l_query = qe - qb;
	const int64_t L = ix.l_pac;
	if (want_cigar) { *n_cigar = 0; *NM = -1; }
	if (l_query <= 0 || rb >= re || (rb < L && re > L)) return false;
	if (re > L << 1 || rb < 0) return false; // bns_get_seq would clip: "re - rb != rlen" (bwa.c:134)
	const int rlen = (int)(re - rb);
	SegView sv{query, qb, qe, rb, re, rb >= L};
	if (l_query == rlen && w_ == 0) { // gap-free shortcut (bwa.c:141-149); NM of the single M run (bwa.c:169-199) in the same walk
		int s = 0, n_mm = 0;
gapfree_counts(ix, query + qb, rb, l_query, &s, &n_mm);
		*score = s;
		if (want_cigar) { cg[0] = (uint32_t)l_query << 4; *n_cigar = 1; if (cap >= 1) *NM = n_mm; }
		return true;
	} else {
		int max_gap = ((l_query + 1) >> 1) - 5; // max_ins == max_del with o=6, e=1, a=1
		max_gap = max_gap > 1 ? max_gap : 1;
		int w = (max_gap + iabs(rlen - l_query) + 1) >> 1;
		w = w < w_ ? w : w_;
		int min_w = iabs(rlen - l_query) + 3;
		w = w > min_w ? w : min_w;
		*score = global2_task(ix, sv, l_query, rlen, w, eh, want_cigar ? z : 0, cg, cap, n_cigar);
	}
	if (want_cigar && *n_cigar <= cap) { // NM = mismatches + gap bases; a leading/trailing D is not counted (bwa.c:169-199)
		int x = 0, y = 0, n_mm = 0, n_gap = 0, nc = *n_cigar;
		for (int k = 0; k < nc; ++k) {
			int op = cg[k] & 0xf, len = cg[k] >> 4;
			if (op == 0) {
				for (int i = 0; i < len; ++i) if (sv.qat(x + i) != sv.tat(ix, y + i)) ++n_mm;
				x += len; y += len;
			} else if (op == 2) { if (k > 0 && k < nc - 1) n_gap += len; y += len; }
			else if (op == 1) { x += len; n_gap += len; }
		}
		*NM = n_mm + n_gap;
	}
	return true;
}

} // namespace arx
