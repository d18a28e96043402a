// hip_sw_coop.h -- wavefront-cooperative Smith-Waterman kernels for gfx950 (HIP only).
//
// sw_u8_g16: the rescue SW (ksw_align2/ksw_u8, ksw.c:111-230,343-365).  The SSE2 original keeps 16 query stripes in the
// 16 byte lanes of an XMM register; here the 16 stripes live in 16 adjacent GPU lanes (one DPP row), four alignments
// per 64-wide wavefront.  H/E/Hmax stay in VGPRs (slen <= SL values per lane, loops fully unrolled), the byte shift of
// `_mm_slli_si128(x, 1)` is a lane shift inside the row, `_mm_movemask_epi8` is a ballot, the horizontal max a 4-step
// butterfly.  Saturation and tie rules are those of the reference and the results are bit-identical to u8_align() in dev_sw.h (the
// one-thread-per-alignment form the parity tests pin) -- but since the end of round 3 the row is no longer computed the reference's way:
// F comes from a prefix scan over the lanes instead of the lazy-F loop (ARX_SW_SCANF), and the 8-bit element size runs on packed 16-bit
// pairs, 32 segments of the query in the 16 lanes (ARX_SW_PACKED); the H values are the same function of the two sequences either way
// (tests/test_sw_prefilter.py on the host double, the GPU suite and the fuzz runs against the compiled reference).
#pragma once
#include "arx_dev.h"
#include "dev_sw.h"
#include "dev_regs.h"

namespace arx {

// Cross-lane traffic inside a 16-lane group uses DPP row operations (a DPP "row" is exactly 16 lanes): they are VALU
// operand modifiers with no LDS round trip, unlike __shfl (ds_bpermute), whose latency dominated the first version of
// these kernels (profiles/r01).  update_dpp(old, src, ctrl, 0xf, 0xf, false): lanes whose source lane falls outside the
// row keep `old`.
constexpr int DPP_ROW_SHR = 0x110, DPP_ROW_ROR = 0x120; // + shift amount 1..15
template <int CTRL> __device__ __forceinline__ int dpp_row(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false); }
// The same with zero fill (bound_ctrl:1, old = 0): in this form the compiler folds the lane movement into the instruction that uses the
// value (v_max_i32_dpp, v_sub_u32_dpp, ...: one instruction), where the general form costs a move of `old`, a v_mov_b32_dpp and the
// operation itself.  Rotations have no invalid source lane, so the fill never shows there.
template <int CTRL> __device__ __forceinline__ int dpp_rowz(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }

__device__ __forceinline__ int g16_shift_up(int v, int) // lane l of each 16-lane row receives lane l-1's value, lane 0 receives 0
{
	return dpp_rowz<DPP_ROW_SHR + 1>(v);
}
template <int N> __device__ __forceinline__ int g16_shift_up_n(int v, int fill) { return dpp_row<DPP_ROW_SHR + N>(fill, v); }
__device__ __forceinline__ int g16_max(int v) // butterfly by row rotations: every lane ends up with the group maximum
{
	int t;
	t = dpp_rowz<DPP_ROW_ROR + 8>(v); v = v > t ? v : t;
	t = dpp_rowz<DPP_ROW_ROR + 4>(v); v = v > t ? v : t;
	t = dpp_rowz<DPP_ROW_ROR + 2>(v); v = v > t ? v : t;
	t = dpp_rowz<DPP_ROW_ROR + 1>(v); v = v > t ? v : t;
	return v;
}
__device__ __forceinline__ int g16_min(int v)
{
	int t;
	t = dpp_rowz<DPP_ROW_ROR + 8>(v); v = v < t ? v : t;
	t = dpp_rowz<DPP_ROW_ROR + 4>(v); v = v < t ? v : t;
	t = dpp_rowz<DPP_ROW_ROR + 2>(v); v = v < t ? v : t;
	t = dpp_rowz<DPP_ROW_ROR + 1>(v); v = v < t ? v : t;
	return v;
}
__device__ __forceinline__ bool g16_all(bool p)
{
	unsigned long long m = __ballot(p);
	return ((m >> (__lane_id() & 48)) & 0xffffull) == 0xffffull;
}

// per-row score lookup: nibble q of the word is S(tb, q) + shift for q = 0..3 (bases), 4 (N) and 5 (stripe padding)
__device__ __forceinline__ uint32_t u8_score_word(int tb)
{
	// match 1+4 = 5, mismatch -4+4 = 0, N -1+4 = 3, padding 0+4 = 4
	if (tb > 3) return 0x433333u;
	return 0x430000u | (5u << (4 * tb));
}

#ifndef ARX_SW_SCANF
#define ARX_SW_SCANF 1 // 0: ksw_u8's lazy-F loop as the reference runs it (A/B)
#endif
struct SwSeqs { // how the two passes read their sequences without materialising them
	const uint8_t *mate; int l_ms;   // forward mate; the query is its reverse complement (bwamem_pair.c:134-137)
	const uint8_t *tl;               // the target window, staged in LDS once per alignment (both passes read it), two bases per byte
	int q_rev, t_rev;                // second pass: the first q_rev / t_rev elements are read back to front (ksw.c:357)
	__device__ __forceinline__ int q0(int k) const { int b = mate[l_ms - 1 - k]; return b < 4 ? 3 - b : 4; }
	__device__ __forceinline__ int q(int k) const { return q0(k < q_rev ? q_rev - 1 - k : k); }
	__device__ __forceinline__ int t(int i) const { const int k = i < t_rev ? t_rev - 1 - i : i; return (tl[k >> 1] >> ((k & 1) << 2)) & 15; }
};

// one ksw_u8 pass by a 16-lane group; every lane returns the same U8Res (score2/te2 only valid in lane 0 of the group)
// FULL: the query fills all SL stripes (slen == SL, e.g. 150 bases at SL = 10), which makes every stripe test a constant
// I16: the pass as ksw_i16 runs it (ksw.c:232-334, mates of 250 bases and more): EIGHT stripes -- lanes 8..15 of the group stay zero: what
// lane 8 would receive from lane 7 through the shifts is masked --, no 255 ceiling; the arithmetic is the byte form's (dev_sw.h: u8_pass).
template <int SL, bool FULL, bool I16 = false>
__device__ U8Res sw_u8_pass_g16_impl(const SwSeqs &sq, int qlen, int tlen, int xtra, uint8_t *rowmax)
{
	const int l = __lane_id() & 15;
	const bool lane_on = !I16 || l < 8;
	const int slen = FULL ? SL : (I16 ? (qlen + 7) >> 3 : (qlen + 15) >> 4);
	const int minsc = (xtra & KSW_XSUBO) ? (xtra & 0xffff) : 0x10000, endsc = (xtra & KSW_XSTOP) ? (xtra & 0xffff) : 0x10000;
	int H0[SL], H1[SL], E[SL], HM[SL], Q4[SL];
#pragma unroll
	for (int j = 0; j < SL; ++j) {
		const int k = j + l * slen;
		H0[j] = H1[j] = E[j] = HM[j] = 0;
		Q4[j] = 4 * ((j < slen && k < qlen && lane_on) ? sq.q(k) : 5);
	}
	int gmax = 0, te = -1, hlast = 0, rows = 0;
	// one row: reads the previous row's H from Hin, leaves this row's in Hout; the caller alternates the two arrays so that no
	// row ends with a register copy per cell.  Returns true when the pass stops after this row.
	auto row = [&](const int (&Hin)[SL], int (&Hout)[SL], int i) __attribute__((always_inline)) -> bool {
		const uint32_t W = u8_score_word(sq.t(i));
		int h = g16_shift_up(hlast, l), f = 0, mx = 0;
		if (I16) h = lane_on ? h : 0;
		if (ARX_SW_SCANF) {
			// F without the lazy loop (end of round 3).  f(j + 1) = max(f(j) - 1, H(j) - 7) with H = max(G, f), G = max(M', E): the f - 7 inside H - 7
			// never beats f - 1, so F is the max-plus prefix of G alone -- the lane's own cells (sweep 1: what they send on as F, starting from
			// nothing), the segments before it (a four-step scan over the 16 lanes, each segment passed costs its length), then the same chain
			// again from what really comes in (sweep 2: H, E, the row maximum).  ksw_u8's lazy-F loop (ksw.c:177-189) iterates to the same H; it
			// took a pass over the row per segment boundary that F crosses -- and behind a high-scoring cell F runs on for dozens of columns.
			// (E takes the final H here where the reference takes the H before its loop: H itself is the same either way -- a deletion
			// right after insertions scores what the insertions after the deletion score, which the next row's F delivers; 46 GPU tests and
			// the fuzz runs, also with the 32-segment layout below whose intermediate E differ again.)
#pragma unroll
			for (int j = 0; j < SL; ++j) {
				const bool valid = j < slen;
				const int s = (int)((W >> Q4[j]) & 15u);
				const int hd = h + s - 4;
				const int G = hd > E[j] ? hd : E[j];
				Hout[j] = G;
				const int g7 = G - 7, f1 = f - 1, fm = f1 > g7 ? f1 : g7;
				const int fn = fm > 0 ? fm : 0;
				f = valid ? fn : f;
				h = Hin[j];
			}
			int x = f, y;
			y = dpp_rowz<DPP_ROW_SHR + 1>(x) - slen; x = x > y ? x : y;
			y = dpp_rowz<DPP_ROW_SHR + 2>(x) - 2 * slen; x = x > y ? x : y;
			y = dpp_rowz<DPP_ROW_SHR + 4>(x) - 4 * slen; x = x > y ? x : y;
			y = dpp_rowz<DPP_ROW_SHR + 8>(x) - 8 * slen; x = x > y ? x : y;
			f = g16_shift_up(x, l); // what enters the lane's first cell (0 in lane 0; x >= 0)
#pragma unroll
			for (int j = 0; j < SL; ++j) {
				const bool valid = j < slen;
				const int G = Hout[j];
				const int hh = G > f ? G : f;
				Hout[j] = hh;
				const int h7 = hh - 7;
				const int e1 = E[j] - 1, em = e1 > h7 ? e1 : h7;
				E[j] = em > 0 ? em : 0;
				const int f1 = f - 1, fm = f1 > h7 ? f1 : h7;
				const int fn = fm > 0 ? fm : 0;
				f = valid ? fn : f;
				const int hm = (valid && lane_on) ? hh : 0; // (16-bit element size: the eight lanes beyond the reference's stripes hold nothing it has)
				mx = mx > hm ? mx : hm;
			}
			const int imax = g16_max(mx);
			if (minsc < 0x10000 && l == 0) rowmax[i] = (uint8_t)imax;
			++rows;
			bool brk = false;
			if (imax > gmax) {
				gmax = imax; te = i;
#pragma unroll
				for (int j = 0; j < SL; ++j) HM[j] = Hout[j];
				if ((!I16 && gmax + 4 >= 255) || gmax >= endsc) brk = true;
			}
#pragma unroll
			for (int j = 0; j < SL; ++j) if (j == slen - 1) hlast = Hout[j];
			return brk;
		}
		// straight-line select code over all SL stripes: stripes past slen (a shorter query in the second pass) come last in
		// the chain, so what they compute flows nowhere as long as they leave f and the row maximum alone.
		// _mm_adds_epu8(h, profile) cannot saturate (h <= 249 for reads below 250 bases, profile <= 5), and the floor of
		// _mm_subs_epu8(h, shift) comes for free from E, f >= 0 in the three-way maximum; both gap states take the floor of their
		// saturating subtractions as the third operand of a v_max3
#pragma unroll
		for (int j = 0; j < SL; ++j) {
			const bool valid = j < slen;
			const int s = (int)((W >> Q4[j]) & 15u);
			const int hd = h + s - 4;
			const int he = hd > E[j] ? hd : E[j];
			const int hh = he > f ? he : f;
			Hout[j] = hh;
			const int h7 = hh - 7;                         // subs(h, oe): o + e = 7 for both gap kinds
			const int e1 = E[j] - 1, em = e1 > h7 ? e1 : h7;
			E[j] = em > 0 ? em : 0;
			const int f1 = f - 1, fm = f1 > h7 ? f1 : h7;
			const int fn = fm > 0 ? fm : 0;
			f = valid ? fn : f;
			h = Hin[j];
		}
#pragma unroll
		for (int j = 0; j < SL; ++j) { const int hm = j < slen ? Hout[j] : 0; mx = mx > hm ? mx : hm; }
		// lazy-F (ksw.c:177-189).  Its first step nearly always ends it: that step is straight-line code, the general loop (which
		// carries the whole row of H through its iterations) sits behind a branch
		{
			bool stop;
			f = g16_shift_up(f, l);
			if (I16) f = lane_on ? f : 0;
			{
				const int hh = Hout[0] > f ? Hout[0] : f;
				Hout[0] = hh;
				int t7 = hh - 7; t7 = t7 > 0 ? t7 : 0;
				f = f - 1; f = f > 0 ? f : 0;
				stop = g16_all(!(f > t7));
			}
			if (!stop) {
				for (int k2 = 0; k2 < 16 && !stop; ++k2) {
					if (k2) { f = g16_shift_up(f, l); if (I16) f = lane_on ? f : 0; }
#pragma unroll
					for (int j = 0; j < SL; ++j) {
						if (j < slen && !stop && (k2 || j)) {
							int hh = Hout[j] > f ? Hout[j] : f;
							Hout[j] = hh;
							int t7 = hh - 7; t7 = t7 > 0 ? t7 : 0;
							f = f - 1; f = f > 0 ? f : 0;
							if (g16_all(!(f > t7))) stop = true;
						}
					}
				}
#pragma unroll
				for (int j = 0; j < SL; ++j) { const int hm = j < slen ? Hout[j] : 0; mx = mx > hm ? mx : hm; }
			}
		}
		const int imax = g16_max(mx);
		if (minsc < 0x10000 && l == 0) rowmax[i] = (uint8_t)imax;
		++rows;
		bool brk = false;
		if (imax > gmax) {
			gmax = imax; te = i;
#pragma unroll
			for (int j = 0; j < SL; ++j) HM[j] = Hout[j];
			if ((!I16 && gmax + 4 >= 255) || gmax >= endsc) brk = true;
		}
#pragma unroll
		for (int j = 0; j < SL; ++j) if (j == slen - 1) hlast = Hout[j];
		return brk;
	};
	for (int i = 0; i < tlen; i += 2) {
		if (row(H0, H1, i)) break;
		if (i + 1 < tlen && row(H1, H0, i + 1)) break;
	}
	U8Res r;
	r.score = (I16 || gmax + 4 < 255) ? gmax : 255; r.te = te; r.qe = -1; r.score2 = -1; r.te2 = -1; r.tb = -1; r.qb = -1;
	if (I16 || r.score != 255) {
		int bv = -1, bq = 0x7fffffff; // this lane's best saved value and the smallest query position holding it
#pragma unroll
		for (int j = 0; j < SL; ++j) {
			if (j < slen && lane_on) {
				int v = HM[j], qp = j + l * slen;
				if (v > bv) { bv = v; bq = qp; }
			}
		}
		const int vmax = g16_max(bv);
		r.qe = g16_min(bv == vmax ? bq : 0x7fffffff);
		if (minsc < 0x10000 && l == 0) { // replay of the b[] list (ksw.c:192-200,218-226), see u8_pass() in dev_sw.h
			const int d = r.score, low = te - d, high = te + d; // (score + max - 1) / max with max = 1
			int bi = -1, bs = -1;
			for (int i = 0; i < rows; ++i) {
				int im = rowmax[i];
				if (im < minsc) continue;
				if (bi < 0 || bi + 1 != i) {
					if (bi >= 0 && (bi < low || bi > high) && bs > r.score2) { r.score2 = bs; r.te2 = bi; }
					bi = i; bs = im;
				} else if (bs < im) { bi = i; bs = im; }
			}
			if (bi >= 0 && (bi < low || bi > high) && bs > r.score2) { r.score2 = bs; r.te2 = bi; }
		}
	}
	return r;
}

// The same pass on PACKED 16-bit pairs (end of round 3): a register holds two cells, the group is 32 virtual lanes -- lane l's low halves are
// segment l of the query, its high halves segment l + 16, each segment S2 = SL / 2 positions long -- and every add / subtract / maximum of
// the row is one v_pk_*_i16 for both.  The recurrence within a segment is the serial one of the unpacked form, the two halves of a register
// never meet except where a segment hands H or F to the next one: there the packed value moves one lane up and lane 0's high half takes lane
// 15's low half (one row rotation and a select).  The score lookup is one v_perm_b32 per pair: the row's five scores as bytes, the pair's two
// query codes as the selector.  Values stay below 255 + 5 (reads under 250 bases), far inside 16 bits; the results are the unpacked form's
// (the DP is the same function of the two sequences however its cells are laid out; the lazy-F loop runs to the same fixed point).
typedef short arx_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int pk_add(int a, int b) { return __builtin_bit_cast(int, (arx_s2)(__builtin_bit_cast(arx_s2, a) + __builtin_bit_cast(arx_s2, b))); }
__device__ __forceinline__ int pk_sub(int a, int b) { return __builtin_bit_cast(int, (arx_s2)(__builtin_bit_cast(arx_s2, a) - __builtin_bit_cast(arx_s2, b))); }
__device__ __forceinline__ int pk_max(int a, int b) { return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(arx_s2, a), __builtin_bit_cast(arx_s2, b))); }
__device__ __forceinline__ int pk_rot_up(int v, int l) // every half to the next segment: lane l <- lane l - 1; lane 0: low half 0 (no segment before the first), high half <- lane 15's low half
{
	const int r = dpp_rowz<DPP_ROW_ROR + 1>(v);
	return l == 0 ? (int)((uint32_t)r << 16) : r;
}
__device__ __forceinline__ int pk_g16_max(int v) // both halves reduced over the group, then against each other: the row maximum in every lane
{
	int t;
	t = dpp_rowz<DPP_ROW_ROR + 8>(v); v = pk_max(v, t);
	t = dpp_rowz<DPP_ROW_ROR + 4>(v); v = pk_max(v, t);
	t = dpp_rowz<DPP_ROW_ROR + 2>(v); v = pk_max(v, t);
	t = dpp_rowz<DPP_ROW_ROR + 1>(v); v = pk_max(v, t);
	const int lo = (int)(short)(v & 0xffff), hi = v >> 16;
	return lo > hi ? lo : hi;
}
template <int S2, bool FULL>
__device__ U8Res sw_u8_pass_p16_impl(const SwSeqs &sq, int qlen, int tlen, int xtra, uint8_t *rowmax)
{
	const int l = __lane_id() & 15;
	const int slen = FULL ? S2 : (qlen + 31) >> 5;
	const int minsc = (xtra & KSW_XSUBO) ? (xtra & 0xffff) : 0x10000, endsc = (xtra & KSW_XSTOP) ? (xtra & 0xffff) : 0x10000;
	const int C4 = 0x00040004, C7 = 0x00070007, C1 = 0x00010001;
	// The reference pads the query to 16 x ceil(qlen / 16) columns with score-0 columns, and those columns count in the row maxima (they hold what
	// the last base's column held a few rows earlier: ksw.c:129-131, 165-166); 32 segments may pad further -- columns from P on stay out of
	// every maximum (FULL: there are none)
	const int P = 16 * ((qlen + 15) >> 4);
	int H0[S2], H1[S2], E[S2], HM[S2];
	uint32_t QS[S2]; // v_perm selector of the pair: byte 0 <- score byte of the low half's query code, byte 2 <- the high half's, bytes 1 and 3 <- 0
	int KM[S2];      // which halves are columns the reference has
#pragma unroll
	for (int j = 0; j < S2; ++j) {
		const int ka = j + l * slen, kb = j + (l + 16) * slen;
		H0[j] = H1[j] = E[j] = HM[j] = 0;
		KM[j] = FULL ? -1 : (int)((ka < P ? 0xffffu : 0u) | (kb < P ? 0xffff0000u : 0u));
		const uint32_t qa = (j < slen && ka < qlen) ? (uint32_t)sq.q(ka) : 5u, qb = (j < slen && kb < qlen) ? (uint32_t)sq.q(kb) : 5u;
		QS[j] = qa | 0x0c00u | qb << 16 | 0x0c000000u;
	}
	int gmax = 0, te = -1, hlast = 0, rows = 0;
	auto row = [&](const int (&Hin)[S2], int (&Hout)[S2], int i) __attribute__((always_inline)) -> bool {
		const int tb = sq.t(i);
		// the row's scores + 4 as bytes: query codes 0..3 in the low word (match 5, mismatch 0; a target N: 3 everywhere), N -> 3 and padding -> 4 in the high word
		const uint32_t Wlo = tb > 3 ? 0x03030303u : 5u << (8 * tb), Whi = 0x0403u;
		int h = pk_rot_up(hlast, l), f = 0, mx = 0;
		if (ARX_SW_SCANF) { // F by a prefix scan over the 32 segments (see the unpacked pass): both halves scan side by side, then the high halves
			// (segments 16 .. 31) take what leaves segment 15, less the segments in between
#pragma unroll
			for (int j = 0; j < S2; ++j) {
				const bool valid = j < slen;
				const int s = (int)__builtin_amdgcn_perm(Whi, Wlo, QS[j]);
				const int G = pk_max(pk_sub(pk_add(h, s), C4), E[j]);
				Hout[j] = G;
				const int fn = pk_max(pk_max(pk_sub(f, C1), pk_sub(G, C7)), 0);
				f = valid ? fn : f;
				h = Hin[j];
			}
			const int sl2 = slen | slen << 16;
			int x = f, y;
			y = pk_sub(dpp_rowz<DPP_ROW_SHR + 1>(x), sl2); x = pk_max(x, y);
			y = pk_sub(dpp_rowz<DPP_ROW_SHR + 2>(x), 2 * sl2); x = pk_max(x, y);
			y = pk_sub(dpp_rowz<DPP_ROW_SHR + 4>(x), 4 * sl2); x = pk_max(x, y);
			y = pk_sub(dpp_rowz<DPP_ROW_SHR + 8>(x), 8 * sl2); x = pk_max(x, y);
			const int T = (int)(short)(__shfl(x, (__lane_id() & 48) | 15, 64) & 0xffff); // what leaves segment 15
			const int ex = dpp_rowz<DPP_ROW_SHR + 1>(x);                                   // what the segments of the own half send in (nothing in lane 0)
			int th = T - l * slen; th = th > 0 ? th : 0;
			f = pk_max(ex, (int)((uint32_t)th << 16));
#pragma unroll
			for (int j = 0; j < S2; ++j) {
				const bool valid = j < slen;
				const int hh = pk_max(Hout[j], f);
				Hout[j] = hh;
				const int h7 = pk_sub(hh, C7);
				E[j] = pk_max(pk_max(pk_sub(E[j], C1), h7), 0);
				const int fn = pk_max(pk_max(pk_sub(f, C1), h7), 0);
				f = valid ? fn : f;
				mx = pk_max(mx, valid ? (FULL ? hh : hh & KM[j]) : 0);
			}
			const int imax = pk_g16_max(mx);
			if (minsc < 0x10000 && l == 0) rowmax[i] = (uint8_t)imax;
			++rows;
			bool brk = false;
			if (imax > gmax) {
				gmax = imax; te = i;
#pragma unroll
				for (int j = 0; j < S2; ++j) HM[j] = Hout[j];
				if (gmax + 4 >= 255 || gmax >= endsc) brk = true;
			}
#pragma unroll
			for (int j = 0; j < S2; ++j) if (j == slen - 1) hlast = Hout[j];
			return brk;
		}
#pragma unroll
		for (int j = 0; j < S2; ++j) {
			const bool valid = j < slen;
			const int s = (int)__builtin_amdgcn_perm(Whi, Wlo, QS[j]);
			const int hd = pk_sub(pk_add(h, s), C4);
			const int hh = pk_max(pk_max(hd, E[j]), f);
			Hout[j] = hh;
			const int h7 = pk_sub(hh, C7);
			E[j] = pk_max(pk_max(pk_sub(E[j], C1), h7), 0);
			const int fn = pk_max(pk_max(pk_sub(f, C1), h7), 0);
			f = valid ? fn : f;
			mx = pk_max(mx, valid ? (FULL ? hh : hh & KM[j]) : 0);
			h = Hin[j];
		}
		{ // lazy-F (ksw.c:177-189): first step straight-line, the general loop behind a branch
			bool stop;
			f = pk_rot_up(f, l);
			{
				const int hh = pk_max(Hout[0], f);
				Hout[0] = hh;
				const int t7 = pk_max(pk_sub(hh, C7), 0);
				f = pk_max(pk_sub(f, C1), 0);
				stop = g16_all(pk_max(f, t7) == t7); // no half with f > t7
			}
			if (!stop) {
				for (int k2 = 0; k2 < 32 && !stop; ++k2) {
					if (k2) f = pk_rot_up(f, l);
#pragma unroll
					for (int j = 0; j < S2; ++j) {
						if (j < slen && !stop && (k2 || j)) {
							const int hh = pk_max(Hout[j], f);
							Hout[j] = hh;
							const int t7 = pk_max(pk_sub(hh, C7), 0);
							f = pk_max(pk_sub(f, C1), 0);
							if (g16_all(pk_max(f, t7) == t7)) stop = true;
						}
					}
				}
				mx = 0;
#pragma unroll
				for (int j = 0; j < S2; ++j) mx = pk_max(mx, j < slen ? (FULL ? Hout[j] : Hout[j] & KM[j]) : 0);
			}
		}
		const int imax = pk_g16_max(mx);
		if (minsc < 0x10000 && l == 0) rowmax[i] = (uint8_t)imax;
		++rows;
		bool brk = false;
		if (imax > gmax) {
			gmax = imax; te = i;
#pragma unroll
			for (int j = 0; j < S2; ++j) HM[j] = Hout[j];
			if (gmax + 4 >= 255 || gmax >= endsc) brk = true;
		}
#pragma unroll
		for (int j = 0; j < S2; ++j) if (j == slen - 1) hlast = Hout[j];
		return brk;
	};
	for (int i = 0; i < tlen; i += 2) {
		if (row(H0, H1, i)) break;
		if (i + 1 < tlen && row(H1, H0, i + 1)) break;
	}
	U8Res r;
	r.score = gmax + 4 < 255 ? gmax : 255; r.te = te; r.qe = -1; r.score2 = -1; r.te2 = -1; r.tb = -1; r.qb = -1;
	if (r.score != 255) {
		int bv = -1, bq = 0x7fffffff; // this lane's best saved value and the smallest query position holding it
#pragma unroll
		for (int j = 0; j < S2; ++j) {
			if (j < slen) {
				const int va = (int)(short)(HM[j] & 0xffff), vb = HM[j] >> 16, qa = j + l * slen, qb = j + (l + 16) * slen;
				if (qa < P && (va > bv || (va == bv && qa < bq))) { bv = va; bq = qa; }
				if (qb < P && (vb > bv || (vb == bv && qb < bq))) { bv = vb; bq = qb; }
			}
		}
		const int vmax = g16_max(bv);
		r.qe = g16_min(bv == vmax ? bq : 0x7fffffff);
		if (minsc < 0x10000 && l == 0) { // replay of the b[] list (ksw.c:192-200,218-226), see u8_pass() in dev_sw.h
			const int d = r.score, low = te - d, high = te + d;
			int bi = -1, bs = -1;
			for (int i = 0; i < rows; ++i) {
				int im = rowmax[i];
				if (im < minsc) continue;
				if (bi < 0 || bi + 1 != i) {
					if (bi >= 0 && (bi < low || bi > high) && bs > r.score2) { r.score2 = bs; r.te2 = bi; }
					bi = i; bs = im;
				} else if (bs < im) { bi = i; bs = im; }
			}
			if (bi >= 0 && (bi < low || bi > high) && bs > r.score2) { r.score2 = bs; r.te2 = bi; }
		}
	}
	return r;
}

#ifndef ARX_SW_PACKED
#define ARX_SW_PACKED 1 // 0: one cell per register (A/B).  (With the lazy-F loop the packed pass was SLOWER, 15.7 -> 17.1 ms: 32 segments double the boundaries that loop carries F across; with the scan it is the faster one)
#endif
template <int SL>
__device__ __forceinline__ U8Res sw_u8_pass_g16(const SwSeqs &sq, int qlen, int tlen, int xtra, uint8_t *rowmax)
{
	if (ARX_SW_PACKED && (SL & 1) == 0)
		return (((qlen + 31) >> 5) == SL / 2 && ((qlen + 15) >> 4) == SL) ? sw_u8_pass_p16_impl<SL / 2, true>(sq, qlen, tlen, xtra, rowmax) : sw_u8_pass_p16_impl<SL / 2, false>(sq, qlen, tlen, xtra, rowmax);
	return ((qlen + 15) >> 4) == SL ? sw_u8_pass_g16_impl<SL, true>(sq, qlen, tlen, xtra, rowmax) : sw_u8_pass_g16_impl<SL, false>(sq, qlen, tlen, xtra, rowmax);
}
template <int SL>
__device__ __forceinline__ U8Res sw_i16_pass_g16(const SwSeqs &sq, int qlen, int tlen, int xtra, uint8_t *rowmax)
{
	return sw_u8_pass_g16_impl<SL, false, true>(sq, qlen, tlen, xtra, rowmax);
}

// one rescue alignment per 16-lane group: forward pass, then the pass over the reversed prefixes (ksw.c:343-365)
// the target window into LDS, two bases per byte (a per-row load of the reference base would put HBM latency on every row)
__device__ __forceinline__ void sw_stage_target(const IndexView &ix, int64_t rb, int tlen, uint8_t *tl)
{
	for (int k = 2 * (__lane_id() & 15); k < tlen; k += 32) {
		const int lo = ref_base(ix, rb + k), hi = k + 1 < tlen ? ref_base(ix, rb + k + 1) : 0;
		tl[k >> 1] = (uint8_t)(lo | hi << 4);
	}
}

template <int SL>
__device__ void sw_u8_align_g16(const IndexView &ix, const uint8_t *mate, int l_ms, int64_t rb, int tlen, uint8_t *rowmax, uint8_t *tl, U8Res *out)
{
	const bool i16 = SL >= 32 && l_ms * OPT_A >= 250; // bwamem_pair.c:150: KSW_XBYTE only below 250; the SL = 32 instantiation serves the batches that hold such mates
	const int xtra = KSW_XSUBO | KSW_XSTART | (i16 ? 0 : KSW_XBYTE) | (OPT_MIN_SEED_LEN * OPT_A);
	// a per-row load of the reference base would put HBM latency on the critical path of every row: fetch the window once
	sw_stage_target(ix, rb, tlen, tl);
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	SwSeqs sq{mate, l_ms, tl, 0, 0};
	U8Res r = i16 ? sw_i16_pass_g16<SL>(sq, l_ms, tlen, xtra, rowmax) : sw_u8_pass_g16<SL>(sq, l_ms, tlen, xtra, rowmax);
	if (!(r.score < (xtra & 0xffff))) {
		SwSeqs s2{mate, l_ms, tl, r.qe + 1, r.te + 1};
		U8Res rr = i16 ? sw_i16_pass_g16<SL>(s2, r.qe + 1, tlen, KSW_XSTOP | r.score, rowmax) : sw_u8_pass_g16<SL>(s2, r.qe + 1, tlen, KSW_XSTOP | r.score, rowmax); // the second pass keeps the element size (ksw.c:358)
		if (r.score == rr.score) { r.tb = r.te - rr.te; r.qb = r.qe - rr.qe; }
	}
	if ((__lane_id() & 15) == 0) *out = r;
}

constexpr int SW_T_CAP = 800;  // rows of a rescue window: PES_HIGH - PES_LOW + MAX_READ_LEN = 784 at most

// The pre-filter of dev_sw.h (sw_prefilter_serial has the derivation) by a 16-lane group: the query's 5-mers go into chained
// lists in LDS (lane l owns the codes with code % 16 == l, so there is nothing to synchronise), the lanes walk the window's
// 5-mers and count the hits per diagonal in byte counters (LDS atomics on the containing word: a diagonal holds at most 245).
// Tasks that need the DP are appended to `order`; the others get a result that says "below min_seed_len" right away.
struct SwFilterLds { uint8_t tl[SW_T_CAP / 2]; uint8_t q[256]; uint8_t head[1024]; uint8_t nxt[256]; uint32_t cnt[(SW_T_CAP + 256) / 4]; };

static __global__ void __launch_bounds__(64) k_sw_filter_g16(IndexView ix, const uint8_t *bases, const int32_t *base_off, const int32_t *lens,
                                                             const SwTask *tasks, U8Res *res, int n, int32_t *order, int32_t *n_order)
{
	__shared__ SwFilterLds L[4];
	const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
	SwFilterLds &S = L[g];
	for (int i = blockIdx.x * 4 + g; i < n; i += gridDim.x * 4) {
		const SwTask t = tasks[i];
		const int r = 2 * t.pair + t.o, l_ms = lens[r], tlen = (int)(t.re - t.rb);
		const uint8_t *mate = bases + base_off[r];
		bool has_n = l_ms > 255 || tlen > SW_T_CAP;
		if (!has_n) {
			sw_stage_target(ix, t.rb, tlen, S.tl);
			for (int k = l; k < l_ms; k += 16) { const int b = mate[l_ms - 1 - k]; S.q[k] = (uint8_t)(b < 4 ? 3 - b : 4); has_n |= b > 3; } // the reverse complement, as the DP reads it
			for (int w = l; w < 256; w += 16) ((uint32_t *)S.head)[w] = 0xffffffffu;
			for (int w = l; w < (SW_T_CAP + 256) / 4; w += 16) S.cnt[w] = 0;
		}
		has_n = (__ballot(has_n) >> (g * 16) & 0xffff) != 0;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		int s = 0;
		if (!has_n) {
			int code = 0;
			for (int k = 0; k < l_ms; ++k) { // every lane rolls over the whole query and files the positions whose code it owns
				code = code >> 2 | S.q[k] << 8;
				if (k >= SWF_K - 1 && (code & 15) == l) { const int p = k - (SWF_K - 1); S.nxt[p] = S.head[code]; S.head[code] = (uint8_t)p; }
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			for (int j = l; j + SWF_K <= tlen; j += 16) {
				int c = 0;
#pragma unroll
				for (int x = 0; x < SWF_K; ++x) { const int k = j + x; c |= ((S.tl[k >> 1] >> ((k & 1) << 2)) & 3) << (2 * x); }
				for (int p = S.head[c]; p != 0xff; p = S.nxt[p]) {
					const int d = j - p + l_ms - 1;
					atomicAdd(&S.cnt[d >> 2], 1u << ((d & 3) << 3));
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			for (int w = l; w < (l_ms + tlen + 3) / 4; w += 16) {
				const uint32_t x = S.cnt[w];
#pragma unroll
				for (int b = 0; b < 4; ++b) { const int c = (int)(x >> (8 * b) & 255) - SWF_FREE; s += c > 0 ? c : 0; }
			}
			for (int m = 8; m; m >>= 1) s += __shfl_xor(s, m, 16);
		}
		if (l == 0) {
			if (has_n || s >= SWF_NEED) order[atomicAdd(n_order, 1)] = i;
			else { U8Res none; none.score = 0; none.te = none.qe = none.score2 = none.te2 = none.tb = none.qb = -1; res[t.slot] = none; }
		}
		__builtin_amdgcn_wave_barrier(); // the LDS block is reused by the group's next task
	}
}

template <int SL>
__global__ void __launch_bounds__(64) k_sw_u8_g16(IndexView ix, const uint8_t *bases, const int32_t *base_off, const int32_t *lens,
                                                  const SwTask *tasks, U8Res *res, int n, const int32_t *order, const int32_t *n_order)
{
	__shared__ uint8_t rowmax_lds[4][SW_T_CAP];
	__shared__ uint8_t target_lds[4][SW_T_CAP / 2];
	const int g = threadIdx.x >> 4;
	if (order) n = *n_order; // the tasks the pre-filter left
	for (int x = blockIdx.x * 4 + g; x < n; x += gridDim.x * 4) {
		const int i = order ? order[x] : x;
		const SwTask t = tasks[i];
		const int r = 2 * t.pair + t.o;
		sw_u8_align_g16<SL>(ix, bases + base_off[r], lens[r], t.rb, (int)(t.re - t.rb), rowmax_lds[g], target_lds[g], &res[t.slot]);
		__builtin_amdgcn_wave_barrier(); // the LDS rows are reused by the group's next alignment
	}
}

// ------------------------------------------------------------------------------------------------------------------
// ext2_g16: banded extension (ksw_extend2, ksw.c:380-479) by a 16-lane group, four extensions per wavefront.
// Column j of the reference's eh[] array lives in lane j / C, register j % C (C columns per lane, 16*C > qlen: the kernel is
// instantiated for a few C and extensions are binned by query length).  Within a row, M(i,j) only depends on the previous
// row, and both gap states are fed from M (not H), so the row is evaluated in sweeps instead of a serial chain: (1) M and the
// insertion seeds t(j) = max(M(j) - 7, 0); (2) F(j) = max_{beg<=k<j} (t(k) + k) - (j - 1), a prefix maximum done per lane
// and then across the 16 lanes with a 4-step scan (the max-plus form of the serial f = max(f - 1, t) chain, exact because
// every t is >= 0); (3) H, E, the row maximum with the reference's tie rule (largest column wins) and the adaptive band,
// the latter as group reductions.  Columns outside [beg, end] keep their stale values exactly as the reference's array
// does, which is what makes the shrinking/growing band bit-exact.  The row body is straight-line select code: every lane
// runs the same instructions, so predication costs nothing and branches would.
// ------------------------------------------------------------------------------------------------------------------
#ifndef ARX_EXT_CHECK_MASK
#define ARX_EXT_CHECK_MASK 3 // the dead-row bound of ext2_g16 is evaluated on rows with all of these bits set
#endif
constexpr int EXT_T_CAP = 512;
template <int C>
__device__ ExtRes ext2_g16(const IndexView &ix, const uint8_t *bases, const ExtTask &t, uint8_t *tl)
{
	const int l = __lane_id() & 15;
	const int qlen = t.qlen, tlen = t.tlen, h0 = t.h0;
	const int c0 = l * C;
	const int NEG = -0x40000000, BIG = 0x7fff;
	int H[C], E[C], Q[C], MIS[C], Mv[C], hv[C], pref[C];
#pragma unroll
	for (int u = 0; u < C; ++u) {
		const int j = c0 + u;
		int v = j == 0 ? h0 : h0 - 6 - j; // first row (ksw.c:395-397): h0, h0-7, then minus one per column while positive
		H[u] = v > 0 ? v : 0;
		E[u] = 0;
		Q[u] = j < qlen ? bases[t.qoff + j * t.qdir] : 4;
		MIS[u] = Q[u] > 3 ? -1 : -OPT_B; // the target never holds an N (2-bit pac)
	}
	int w = t.w;
	{
		int mg = qlen + OPT_PEN_CLIP5 - 5; // max_ins == max_del (ksw.c:402-407)
		mg = mg > 1 ? mg : 1;
		w = w < mg ? w : mg;
	}
	// the target once into LDS (a per-row load would put HBM latency on every row's critical path); rows beyond the staging
	// capacity, which only very long reads reach, load directly
	const int n_stage = tlen < EXT_T_CAP ? tlen : EXT_T_CAP;
	for (int k = l; k < n_stage; k += 16) tl[k] = (uint8_t)ref_base(ix, t.tpos + (int64_t)k * t.tdir);
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	int max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0, beg = 0, end = qlen;
	for (int i = 0; i < tlen; ++i) {
		const int tb = i < EXT_T_CAP ? tl[i] : ref_base(ix, t.tpos + (int64_t)i * t.tdir);
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		int h1init = 0;
		if (beg == 0) { h1init = h0 - (OPT_O_DEL + OPT_E_DEL * (i + 1)); if (h1init < 0) h1init = 0; }
		const unsigned nb = end > beg ? (unsigned)(end - beg) : 0u; // beg can pass end once the rows run beyond the query: empty row, m == 0 ends it
		const int jb0 = c0 - beg; // column - beg for register 0
		// sweep 1: M, and the running maximum of t(k) + k inside the lane
		int pm = NEG;
#pragma unroll
		for (int u = 0; u < C; ++u) {
			const bool act = (unsigned)(jb0 + u) < nb;
			const int sc = Q[u] == tb ? OPT_A : MIS[u];
			const int M = (act && H[u] != 0) ? H[u] + sc : 0;
			Mv[u] = M;
			int tv = M - 7; tv = tv > 0 ? tv : 0;
			pref[u] = pm;
			const int key = act ? tv + (c0 + u) : NEG;
			pm = pm > key ? pm : key;
		}
		// exclusive prefix maximum of the lane maxima across the group
		int x = pm, y;
		y = g16_shift_up_n<1>(x, NEG); x = x > y ? x : y;
		y = g16_shift_up_n<2>(x, NEG); x = x > y ? x : y;
		y = g16_shift_up_n<4>(x, NEG); x = x > y ? x : y;
		y = g16_shift_up_n<8>(x, NEG); x = x > y ? x : y;
		const int ex = g16_shift_up_n<1>(x, NEG);
		// sweep 2: F, H, E and the lane's row maximum as (h << 8 | column): the largest column wins ties (ksw.c:437)
		int best = -1;
#pragma unroll
		for (int u = 0; u < C; ++u) {
			const int j = c0 + u;
			const bool act = (unsigned)(jb0 + u) < nb;
			const int pmx = ex > pref[u] ? ex : pref[u];
			const int F = pmx - (j - 1); // at j == beg this is hugely negative, like the serial chain's f = 0 it never beats M, E >= 0
			int h = Mv[u] > E[u] ? Mv[u] : E[u];
			h = h > F ? h : F;
			hv[u] = act ? h : 0;
			int tv = Mv[u] - 7; tv = tv > 0 ? tv : 0;
			int e = E[u] - 1; e = e > tv ? e : tv;
			E[u] = act ? e : E[u];
			const int pk = act ? (h << 8 | j) : -1;
			best = best > pk ? best : pk;
		}
		// sweep 3: eh[j].h <- H(i, j-1): shift by one column, across the lane boundary by one lane; eh[beg].h <- h1, eh[end].e <- 0;
		// then the adaptive band (ksw.c:466-469) on the freshly written row: first / last column of [beg, end] that is not all zero
		const int from_prev = g16_shift_up_n<1>(hv[C - 1], 0);
		int first = BIG, last = -1;
#pragma unroll
		for (int u = 0; u < C; ++u) {
			const int j = c0 + u;
			const int prev = u == 0 ? from_prev : hv[u > 0 ? u - 1 : 0];
			const bool shifted = (unsigned)(jb0 + u - 1) < nb; // beg < j <= end
			int hn = shifted ? prev : H[u];
			hn = j == beg ? h1init : hn;
			H[u] = hn;
			const int en = j == end ? 0 : E[u];
			E[u] = en;
			const bool nz = (hn | en) != 0;
			const bool in_row = (unsigned)(jb0 + u) <= nb; // beg <= j <= end
			first = (nz && in_row && j != end && j < first) ? j : first;
			last = (nz && in_row) ? j : last; // u ascends: the last hit is the largest column of this lane
		}
		const int packed = g16_max(best);
		const int m = packed < 0 ? 0 : packed >> 8, mj = packed < 0 ? -1 : packed & 0xff;
		if (end == qlen) {
			int own = -1;
#pragma unroll
			for (int u = 0; u < C; ++u) own = (c0 + u == end - 1 && end - 1 >= beg) ? hv[u] : own;
			int h1 = g16_max(own);
			if (h1 < 0) h1 = h1init; // empty row: h1 keeps its initial value
			max_ie = gscore > h1 ? max_ie : i;
			gscore = gscore > h1 ? gscore : h1;
		}
		if (m == 0) break;
		if (m > max) {
			max = m; max_i = i; max_j = mj;
			int off = mj - i; off = off < 0 ? -off : off;
			max_off = max_off > off ? max_off : off;
		} else {
			if (i - max_i > mj - max_j) { if (max - m - ((i - max_i) - (mj - max_j)) * OPT_E_DEL > OPT_ZDROP) break; }
			else { if (max - m - ((mj - max_j) - (i - max_i)) * OPT_E_INS > OPT_ZDROP) break; }
		}
		first = g16_min(first);
		last = g16_max(last);
		const int nbeg = first < end ? first : end;
		if (last < nbeg) last = nbeg - 1;
		beg = nbeg;
		end = last + 2 < qlen ? last + 2 : qlen;
		// Rows that can no longer matter.  Nothing after this row is reported unless a later row beats `max` or reaches
		// `gscore` in the last column.  A score only grows by matches, one per remaining query column: whatever follows from
		// eh[j].h (the diagonal predecessor of column j in the next row, stale columns included) is at most eh[j].h + (qlen - j),
		// from eh[j].e at most eh[j].e + (qlen - 1 - j), from the row boundary at most h1 + qlen; insertions and deletions only
		// lose.  If that bound is below both, the reference's loop would run on (up to z-drop) without changing its result:
		// about half of the rows of a typical extension.  Checked every fourth row.
		if ((i & ARX_EXT_CHECK_MASK) == ARX_EXT_CHECK_MASK && m < max && gscore >= 0) {
			int bound = beg == 0 ? h0 - (OPT_O_DEL + OPT_E_DEL * (i + 2)) + qlen : -1;
#pragma unroll
			for (int u = 0; u < C; ++u) {
				const int j = c0 + u;
				const int a = H[u] + (qlen - j), b = E[u] + (qlen - 1 - j);
				const int x = j <= qlen ? (a > b ? a : b) : -1;
				bound = bound > x ? bound : x;
			}
			bound = g16_max(bound);
			if (bound < max && bound < gscore) break;
		}
	}
	ExtRes r;
	r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
	return r;
}

// ------------------------------------------------------------------------------------------------------------------
// ext2_b16 (round 3): the same recurrence and the same column layout as ext2_g16 (column j in lane j / C, register j % C), with the
// band predicates taken out of the arithmetic -- 27 vector instructions per column instead of ~50:
//  * left of the band.  beg only grows, so a column that has left the band on the left is never read again (ksw.c:411-412, 466-467).
//    Such columns hold H = E = 0 here: the zero scan that moves beg only passes columns that are zero already, and the one column a row
//    can lose to the band limit (beg = i - w) is zeroed when that happens.  A zero column computes zeros (M = 0, e = 0, its key t + k = k
//    never lifts the F of a column inside the band above 0), hands H = 0 = h1 to column beg, and ranks below every live column in the
//    row maximum: no "j >= beg" test anywhere; the row's first column gets h1 as the fill of the lane shift (only column 0 can need a
//    non-zero one).
//  * right of the band.  Columns beyond `end` keep their stale values as the reference's array does (they are read again when the
//    band re-grows, ksw.c:468-469); they are computed like the others and only the write-back is masked: one compare for "j < end",
//    one for "j == end" (eh[end] = {h1, 0}).  Their keys only reach prefixes of columns further right, i.e. columns that are masked too.
//  * M = H ? H + s : 0 is min(H + s, H << 15): for H = 0 that is min(s, 0) <= 0, and a non-positive M acts exactly like 0 in
//    everything that follows (h = max(M, e, f) with e, f >= 0; t = max(M - 7, 0)); the score s comes out of one nibble table per column.
//  * positions inside a lane are compile-time constants (u), the lane's base column is added once per row where a reduction needs it;
//    the zero scan is one bit per column, first / last set bit per lane, one packed 16-bit reduction for both.
//  * dead-row bound as in ext2_g16, but a column with H == 0 (E == 0) contributes 0, not qlen - j: nothing follows from a zero.
// Results are those of ext2_g16 / ext2_task bit for bit (ARX_EXT_OLD=1 runs the old kernel for comparison).
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int g16_pkmax_u16(int v) // both 16-bit halves reduced at once
{
	typedef unsigned short us2 __attribute__((ext_vector_type(2)));
	auto pk = [](int a, int b) { us2 x = __builtin_bit_cast(us2, a), y = __builtin_bit_cast(us2, b); us2 z = __builtin_elementwise_max(x, y); return __builtin_bit_cast(int, z); };
	int t;
	t = dpp_rowz<DPP_ROW_ROR + 8>(v); v = pk(v, t);
	t = dpp_rowz<DPP_ROW_ROR + 4>(v); v = pk(v, t);
	t = dpp_rowz<DPP_ROW_ROR + 2>(v); v = pk(v, t);
	t = dpp_rowz<DPP_ROW_ROR + 1>(v); v = pk(v, t);
	return v;
}

template <int C>
__device__ ExtRes ext2_b16(const IndexView &ix, const uint8_t *bases, const ExtTask &t, uint8_t *tl)
{
	const int l = __lane_id() & 15;
	const int qlen = t.qlen, tlen = t.tlen, h0 = t.h0;
	const int c0 = l * C;
	const int NEG = -0x40000000;
	int H[C], E[C], Mv[C], pref[C];
	uint32_t SCW[C]; // nibble tb of SCW[u]: score of column c0 + u against target base tb, plus 4
#pragma unroll
	for (int u = 0; u < C; ++u) {
		const int j = c0 + u;
		int v = j == 0 ? h0 : h0 - 6 - j; // first row (ksw.c:395-397)
		H[u] = v > 0 ? v : 0;
		E[u] = 0;
		const int q = j < qlen ? bases[t.qoff + j * t.qdir] : 4;
		SCW[u] = q > 3 ? 0x3333u : (5u << (4 * q)); // match 1 + 4, mismatch -4 + 4, N -1 + 4 (the target never holds an N)
	}
	int w = t.w;
	{
		int mg = qlen + OPT_PEN_CLIP5 - 5; // max_ins == max_del (ksw.c:402-407)
		mg = mg > 1 ? mg : 1;
		w = w < mg ? w : mg;
	}
	const int n_stage = tlen < EXT_T_CAP ? tlen : EXT_T_CAP;
	for (int k = l; k < n_stage; k += 16) tl[k] = (uint8_t)ref_base(ix, t.tpos + (int64_t)k * t.tdir);
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	const int rq_lane = l == qlen / C ? qlen % C : -1; // the register that holds column qlen, in the lane that holds it
	int max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0, beg = 0, end = qlen;
	for (int i = 0; i < tlen; ++i) {
		const int tb4 = 4 * (i < EXT_T_CAP ? tl[i] : ref_base(ix, t.tpos + (int64_t)i * t.tdir));
		if (beg < i - w) { // the band limit drops column i - w - 1 (at most one per row): columns left of the band hold zeros
			const int dj = i - w - 1 - c0;
#pragma unroll
			for (int u = 0; u < C; ++u) { const bool z = dj == u; H[u] = z ? 0 : H[u]; E[u] = z ? 0 : E[u]; }
			beg = i - w;
		}
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		int h1init = h0 - (OPT_O_DEL + OPT_E_DEL * (i + 1));
		h1init = (beg == 0 && h1init > 0) ? h1init : 0;
		const int nact = end - c0; // registers u < nact are inside the band, u == nact is column `end`
		// sweep 1: M and the lane's running maximum of t(k) + k + 1 (k counted from the lane's first column; the + 1 makes every key
		// positive, so that 0 is "no column yet" and the lane shifts of the scan can fill with zeros -- the form the compiler folds into
		// the maximum, one instruction per step)
		int pm = 0;
#pragma unroll
		for (int u = 0; u < C; ++u) {
			const int sc4 = (int)((SCW[u] >> tb4) & 15u);
			const int hs = H[u] + sc4 - 4, hz = H[u] << 15;
			const int M = hs < hz ? hs : hz;
			Mv[u] = M;
			pref[u] = pm;
			int key = M + (u + 1 - 7); key = key > u + 1 ? key : u + 1;
			pm = pm > key ? pm : key;
		}
		int x = pm + c0, y;
		y = dpp_rowz<DPP_ROW_SHR + 1>(x); x = x > y ? x : y;
		y = dpp_rowz<DPP_ROW_SHR + 2>(x); x = x > y ? x : y;
		y = dpp_rowz<DPP_ROW_SHR + 4>(x); x = x > y ? x : y;
		y = dpp_rowz<DPP_ROW_SHR + 8>(x); x = x > y ? x : y;
		const int ex = dpp_rowz<DPP_ROW_SHR + 1>(x) - c0; // exclusive prefix maximum relative to this lane's first column: >= 0 (the lane before ends on a key >= its base + C), 0 in lane 0
		// sweep 2: F, H, E; the lane's row maximum as (h << 8 | u)
		int best = 0, hlast = 0;
#pragma unroll
		for (int u = 0; u < C; ++u) {
			const int pmx = ex > pref[u] ? ex : pref[u];
			const int F = pmx - u; // (max of t(k) + k + 1) - 1 - (j - 1); 0 - u <= 0 where no column precedes: like the serial chain's f = 0 it never beats M, E >= 0
			int h = Mv[u] > E[u] ? Mv[u] : E[u];
			h = h > F ? h : F;
			const int m7 = Mv[u] - 7;
			int e = E[u] - 1; e = e > m7 ? e : m7; e = e > 0 ? e : 0;
			const bool act = nact > u;
			E[u] = act ? e : E[u];
			const int pk = (act ? h : 0) << 8 | u;
			best = best > pk ? best : pk;
			// eh[j + 1].h <- H(i, j) for j + 1 <= end (H[u + 1] was consumed by sweep 1: written in place, no copy of the row kept)
			if (u + 1 < C) H[u + 1 < C ? u + 1 : 0] = act ? h : H[u + 1 < C ? u + 1 : 0];
			else hlast = h;
		}
		// sweep 3: the lane's first column takes the last one of the lane before; eh[end].e <- 0; one bit per column for the zero scan
		const int h1lane = l == 0 ? h1init : 0;
		int from_prev = dpp_rowz<DPP_ROW_SHR + 1>(hlast);
		from_prev = from_prev > h1lane ? from_prev : h1lane; // lane 0 (column 0): h1; scores are >= 0
		H[0] = nact >= 0 ? from_prev : H[0];
		uint32_t zm = 0;
		int own = -1;
#pragma unroll
		for (int u = 0; u < C; ++u) {
			const int hn = H[u];
			const int en = nact == u ? 0 : E[u];
			E[u] = en;
			const uint32_t nz = (uint32_t)(hn | en);
			zm |= (nz < 1u ? nz : 1u) << u;
			own = rq_lane == u ? hn : own;
		}
		const int packed = g16_max(best + c0);
		const int m = packed >> 8, mj = packed & 0xff;
		if (end == qlen) { // column qlen now holds H(i, qlen - 1), or h1 after an empty row
			const int h1 = g16_max(own);
			max_ie = gscore > h1 ? max_ie : i;
			gscore = gscore > h1 ? gscore : h1;
		}
		if (m == 0) break;
		if (m > max) {
			max = m; max_i = i; max_j = mj;
			int off = mj - i; off = off < 0 ? -off : off;
			max_off = max_off > off ? max_off : off;
		} else {
			if (i - max_i > mj - max_j) { if (max - m - ((i - max_i) - (mj - max_j)) * OPT_E_DEL > OPT_ZDROP) break; }
			else { if (max - m - ((mj - max_j) - (i - max_i)) * OPT_E_INS > OPT_ZDROP) break; }
		}
		// adaptive band (ksw.c:466-469): first non-zero column of [beg, end), last non-zero column of [beg', end]
		{
			int nk = nact + 1; nk = nk < 0 ? 0 : (nk > C ? C : nk);
			int ne = nact < 0 ? 0 : (nact > C ? C : nact);
			const uint32_t zk = zm & ((1u << nk) - 1u), ze = zm & ((1u << ne) - 1u);
			const int lo = ze ? c0 + __builtin_ctz(ze) : 1023, hi = zk ? c0 + 32 - __builtin_clz(zk) : 0; // hi = last column + 1
			const int red = g16_pkmax_u16((1023 - lo) << 16 | hi);
			const int first = 1023 - (int)((uint32_t)red >> 16);
			int last = (red & 0xffff) - 1;
			const int nbeg = first < end ? first : end;
			if (last < nbeg) last = nbeg - 1;
			beg = nbeg;
			end = last + 2 < qlen ? last + 2 : qlen;
		}
		// rows that can no longer matter (see ext2_g16); a zero H or E starts nothing
		if ((i & ARX_EXT_CHECK_MASK) == ARX_EXT_CHECK_MASK && m < max && gscore >= 0) {
			int bound = beg == 0 ? h0 - (OPT_O_DEL + OPT_E_DEL * (i + 2)) + qlen : -1;
			const int rem = qlen - c0;
#pragma unroll
			for (int u = 0; u < C; ++u) {
				const int a = H[u] ? H[u] + (rem - u) : 0, b = E[u] ? E[u] + (rem - 1 - u) : 0;
				const int xx = rem - u >= 0 ? (a > b ? a : b) : -1;
				bound = bound > xx ? bound : xx;
			}
			bound = g16_max(bound);
			if (bound < max && bound < gscore) break;
		}
	}
	ExtRes r;
	r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
	return r;
}

// (Handing a workgroup 16 or 64 extensions at a time, sorted by query length so that the four groups of a wavefront run rows of similar
// count in lockstep, was measured in round 3: no gain at 16, twice the time at 64 -- the kernel is not bound by the idle lanes of a
// finished group.  Extensions are taken four at a time in queue order.)
template <int C, bool OLD>
__device__ __forceinline__ void extend_class_b16(const IndexView &ix, const uint8_t *bases, const ExtTask *tasks, ExtRes *res, int n, int blk, int n_blk, uint8_t *tl)
{
	const int g = threadIdx.x >> 4;
	for (int i = blk * 4 + g; i < n; i += n_blk * 4) {
		const ExtTask t = tasks[i];
		ExtRes r = OLD ? ext2_g16<C>(ix, bases, t, tl) : ext2_b16<C>(ix, bases, t, tl);
		if ((threadIdx.x & 15) == 0) res[t.owner] = r;
		__builtin_amdgcn_wave_barrier(); // the LDS row is reused by the group's next extension
	}
}

#ifndef ARX_EXT_WAVES_ATTR
#define ARX_EXT_WAVES_ATTR
#endif
template <int C, bool OLD>
__global__ void __launch_bounds__(64) ARX_EXT_WAVES_ATTR k_extend_b16(IndexView ix, const uint8_t *bases, const ExtTask *tasks, ExtRes *res, int n)
{
	__shared__ uint8_t target_lds[4][EXT_T_CAP];
	extend_class_b16<C, OLD>(ix, bases, tasks, res, n, blockIdx.x, gridDim.x, target_lds[threadIdx.x >> 4]);
}

// All query-length classes of a round in one launch: blocks [0, nb[0]) take class 0, the next nb[1] class 1, ...  Every wavefront
// runs one tiling, and the classes run side by side (late rounds hold few extensions: launched one class after the other, each launch
// would cost the latency of a whole DP).
struct ExtClassShape { int n[EXT_CLASSES], nb[EXT_CLASSES]; };
template <bool OLD>
__global__ void __launch_bounds__(64) k_extend_classes_b(IndexView ix, const uint8_t *bases, const ExtTask *tasks, int stride, ExtRes *res, ExtClassShape sh)
{
	__shared__ uint8_t target_lds[4][EXT_T_CAP];
	uint8_t *tl = target_lds[threadIdx.x >> 4];
	int b = blockIdx.x;
#define ARX_EXT_CASE(c, CN, CO) if (b < sh.nb[c]) { extend_class_b16<OLD ? CO : CN, OLD>(ix, bases, tasks + (size_t)(c) * stride, res, sh.n[c], b, sh.nb[c], tl); return; } b -= sh.nb[c];
	ARX_EXT_CASE(0, 2, 4) ARX_EXT_CASE(1, 3, 4) ARX_EXT_CASE(2, 4, 4) ARX_EXT_CASE(3, 6, 7) ARX_EXT_CASE(4, 8, 10) ARX_EXT_CASE(5, 10, 10)
	extend_class_b16<16, OLD>(ix, bases, tasks + (size_t)6 * stride, res, sh.n[6], b, sh.nb[6], tl);
#undef ARX_EXT_CASE
}

} // namespace arx
