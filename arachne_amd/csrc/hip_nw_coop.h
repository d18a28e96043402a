// hip_nw_coop.h -- CIGAR generation for the gapped regions by 16-lane groups (HIP only): mem_reg2aln (bwamem.c:1086-1156)
// -> bwa_gen_cigar2 (bwa.c:121-207) -> ksw_global2 (ksw.c:499-606), four regions per wavefront.
//
// The band of ksw_global2 is narrow (n_col = min(qlen, 2w + 1), typically ~20 columns), so the group keeps the band, not the
// whole row: lane l owns band offsets [l*CP, (l+1)*CP), offset o being column beg_i + o of row i.  The reference's eh[]
// array holds H(i-1, j-1) in eh[j].h and the pending E(i, j) in eh[j].e; when the band start moves right by one column
// between rows (i >= w) H therefore stays in place and E moves down one offset, while the band start is pinned at column 0
// (i < w) it is the other way round -- one DPP row shift per row either way.  Both gap states are fed from M, so F is a
// max-plus prefix scan over the row (lane-local, then a 4-step scan over the 16 lanes), with the reference's -2^30 "minus
// infinity" carried through the same integer arithmetic so that every direction bit of the traceback matrix is identical.
// The matrix goes to this region's own slice of HBM (sized from its band); lane 0 walks it back and builds the CIGAR.
#pragma once
#include "hip_sw_coop.h"
#include "dev_fm.h" // Q16

namespace arx {

constexpr int DPP_ROW_SHL = 0x100;
template <int N> __device__ __forceinline__ int g16_shift_down_n(int v, int fill) { return dpp_row<DPP_ROW_SHL + N>(fill, v); }


constexpr int NW_ZL_BYTES = 2048; // LDS per group for the rows of the traceback matrix being walked
__device__ __forceinline__ int g16_bcast0(int v) { return __shfl(v, __lane_id() & 48, 64); } // lane 0 of the group

struct NwSeg { const uint8_t *q, *t; int qlen, tlen; }; // oriented as bwa_gen_cigar2 reads them (reverse strand: both back to front)

// ksw_global2 with traceback.  Returns the score in every lane; *n_cigar (lane 0 only meaningful) may exceed cap on overflow.
template <int CP>
__device__ int nw_g16(const NwSeg &sg, int w, uint8_t *z, uint8_t *zl, uint32_t *cg, int cap, int *n_cigar)
{
	const int l = __lane_id() & 15, o0 = l * CP;
	const int qlen = sg.qlen, tlen = sg.tlen;
	constexpr int stride = 16 * CP; // bytes of one row of the traceback matrix: the band (n_col <= stride columns) padded so that a lane's CP direction bytes go out as one aligned store
	const int NEG = NW_MINUS_INF, LOW = -0x7f000000;
	int H[CP], E[CP], M[CP], hv[CP], en[CP], pref[CP];
#pragma unroll
	for (int u = 0; u < CP; ++u) {
		const int j = o0 + u;
		H[u] = j == 0 ? 0 : ((j <= qlen && j <= w) ? -(OPT_O_INS + OPT_E_INS * j) : NEG);
		E[u] = NEG;
	}
	int last_h = NEG;
	for (int i = 0; i < tlen; ++i) {
		const int beg = i > w ? i - w : 0, end = i + w + 1 < qlen ? i + w + 1 : qlen, nb = end - beg;
		const int tb = sg.t[i];
		const int h1init = beg == 0 ? -(OPT_O_DEL + OPT_E_DEL * (i + 1)) : NEG;
		int pm = LOW;
#pragma unroll
		for (int u = 0; u < CP; ++u) {
			const int o = o0 + u;
			const bool act = o < nb;
			const int m = H[u] + (act ? sc_mat(tb, sg.q[beg + o]) : 0);
			M[u] = m;
			pref[u] = pm;
			if (act) { const int key = m - (OPT_O_INS + OPT_E_INS) + o * OPT_E_INS; pm = pm > key ? pm : key; }
		}
		int x = pm, y;
		y = g16_shift_up_n<1>(x, LOW); x = x > y ? x : y;
		y = g16_shift_up_n<2>(x, LOW); x = x > y ? x : y;
		y = g16_shift_up_n<4>(x, LOW); x = x > y ? x : y;
		y = g16_shift_up_n<8>(x, LOW); x = x > y ? x : y;
		const int ex = g16_shift_up_n<1>(x, LOW);
		uint8_t *zi = z + (size_t)i * stride;
		uint32_t dpk[(CP + 3) / 4];
#pragma unroll
		for (int u = 0; u < (CP + 3) / 4; ++u) dpk[u] = 0;
#pragma unroll
		for (int u = 0; u < CP; ++u) {
			const int o = o0 + u;
			if (o < nb) {
				const int pmx = ex > pref[u] ? ex : pref[u];
				int f = pmx - (o - 1) * OPT_E_INS;          // max over k < o of t(k) - (o - 1 - k) e_ins ...
				const int f0 = NEG - o * OPT_E_INS;          // ... and the row's initial -inf, decremented like the serial chain does
				f = f > f0 ? f : f0;
				const int m = M[u];
				int e = E[u], h, t, d;
				d = m >= e ? 0 : 1;
				h = m >= e ? m : e;
				d = h >= f ? d : 2;
				h = h >= f ? h : f;
				hv[u] = h;
				t = m - (OPT_O_DEL + OPT_E_DEL);
				e -= OPT_E_DEL;
				d |= e > t ? 1 << 2 : 0;
				e = e > t ? e : t;
				en[u] = e;
				t = m - (OPT_O_INS + OPT_E_INS);
				f -= OPT_E_INS;
				d |= f > t ? 2 << 4 : 0;
				dpk[u >> 2] |= (uint32_t)d << (8 * (u & 3));
			} else { hv[u] = NEG; en[u] = NEG; }
		}
		if (o0 < nb) { // this lane's direction bytes of the row, one store
			if (CP == 1) zi[o0] = (uint8_t)dpk[0];
			else if (CP == 2) *(uint16_t *)(zi + o0) = (uint16_t)dpk[0];
			else {
#pragma unroll
				for (int u = 0; u < (CP + 3) / 4; ++u) ((uint32_t *)(zi + o0))[u] = dpk[u];
			}
		}
		if (i == tlen - 1) { // score = eh[qlen].h = H(tlen-1, qlen-1): the last active offset (w >= |tlen - qlen| + 3 puts column qlen-1 in the band)
			int own = LOW;
#pragma unroll
			for (int u = 0; u < CP; ++u) if (o0 + u == nb - 1) own = hv[u];
			last_h = g16_max(own);
		}
		if (i >= w) { // next row starts one column further right: eh[j].h = H(i, j-1) is already in place, E(i+1, j) moves down one offset
			const int from_next = g16_shift_down_n<1>(en[0], NEG);
#pragma unroll
			for (int u = 0; u < CP; ++u) {
				H[u] = hv[u];
				E[u] = u + 1 < CP ? en[u + 1] : from_next;
			}
		} else { // band start pinned at column 0: H moves up one offset, offset 0 takes the row's boundary value
			const int from_prev = g16_shift_up_n<1>(hv[CP - 1], h1init);
#pragma unroll
			for (int u = 0; u < CP; ++u) {
				H[u] = u > 0 ? hv[u - 1] : from_prev;
				E[u] = en[u];
			}
		}
	}
	// backtrack (ksw.c:588-603): lane 0 walks the matrix; the group copies it into LDS a block of rows at a time (a walk
	// through HBM costs a dependent access per step), and the CIGAR run being built stays in registers until its op changes
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	{
		constexpr int ROWS = NW_ZL_BYTES / stride;
		int n = 0, which = 0, run_op = -1, run_len = 0;
		int i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
		while (i >= 0 && k >= 0) {
			const int lo = i - ROWS + 1 > 0 ? i - ROWS + 1 : 0;
			const int bytes = (i - lo + 1) * stride;
			__builtin_amdgcn_wave_barrier();
			for (int b16 = l * 16; b16 < bytes; b16 += 256) *(Q16 *)(zl + b16) = *(const Q16 *)(z + (size_t)lo * stride + b16);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			// every lane of the group walks the same path (the reads are broadcasts); while the walk is on a diagonal the 16 lanes look at the next
			// 16 cells of it at once and the run of matches up to the first cell that leaves the diagonal is taken in one step -- lane 0 alone,
			// one cell per step, was the longest serial stretch of this kernel (the DP rows are done by the group, the walk was not)
			while (i >= lo && k >= 0) {
				if (which == 0) {
					const int ii = i - l, kk = k - l;
					const bool valid = ii >= lo && kk >= 0;
					const int b = valid ? zl[(ii - lo) * stride + (kk - (ii > w ? ii - w : 0))] : 3;
					const unsigned m16 = (unsigned)(__ballot(valid && (b & 3) == 0) >> (__lane_id() & 48)) & 0xffffu;
					const int J = __builtin_ctz(~m16); // cells ahead that stay on the diagonal (0..16)
					if (J > 0) {
						if (run_op == 0) run_len += J;
						else { if (run_op >= 0) { if (l == 0 && n < cap) cg[n] = (uint32_t)run_len << 4 | run_op; ++n; } run_op = 0; run_len = J; }
						i -= J; k -= J;
						continue;
					}
				}
				which = zl[(i - lo) * stride + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
				const int op = which == 0 ? 0 : (which == 1 ? 2 : 1);
				if (op == run_op) ++run_len;
				else { if (run_op >= 0) { if (l == 0 && n < cap) cg[n] = (uint32_t)run_len << 4 | run_op; ++n; } run_op = op; run_len = 1; }
				if (which == 0) { --i; --k; } else if (which == 1) --i; else --k;
			}
		}
		if (l == 0) {
			if (i >= 0) { if (run_op == 2) run_len += i + 1; else { if (run_op >= 0) { if (n < cap) cg[n] = (uint32_t)run_len << 4 | run_op; ++n; } run_op = 2; run_len = i + 1; } }
			if (k >= 0) { if (run_op == 1) run_len += k + 1; else { if (run_op >= 0) { if (n < cap) cg[n] = (uint32_t)run_len << 4 | run_op; ++n; } run_op = 1; run_len = k + 1; } }
			if (run_op >= 0) { if (n < cap) cg[n] = (uint32_t)run_len << 4 | run_op; ++n; }
			if (n <= cap) for (int a2 = 0; a2 < n >> 1; ++a2) { uint32_t xx = cg[a2]; cg[a2] = cg[n - 1 - a2]; cg[n - 1 - a2] = xx; }
			*n_cigar = n;
		}
	}
	return last_h;
}

// bwa_gen_cigar2 for a region already staged in LDS (score, CIGAR, NM); every lane returns the same values.
// LO / HI: the lane tilings (band columns per lane) this instantiation holds.  The kernel is compiled once per band class with only the
// tilings that class can need -- its own and the one a doubled band takes -- because the registers of the widest tiling are what set
// the occupancy of the whole kernel (180 VGPRs, two wavefronts per SIMD, with all five in one kernel; the dependent chain of a DP row
// needs many more to hide).  Returns false when the band needs a wider tiling than HI: the caller hands the region to the launch that
// holds them all.
template <int LO, int HI>
__device__ bool gen_cigar2_g16(const NwSeg &sg, int w_, uint8_t *z, uint8_t *zl, uint32_t *cg, int cap, int *score, int *n_cigar, int *NM)
{
	const int l = __lane_id() & 15;
	const int l_query = sg.qlen, rlen = sg.tlen;
	int nc = 0, sc = 0;
	if (l_query == rlen && w_ == 0) { // gap-free shortcut (bwa.c:141-149)
		int s = 0;
		for (int i = l; i < l_query; i += 16) s += sc_mat(sg.t[i], sg.q[i]);
		s += dpp_row<DPP_ROW_ROR + 8>(s, s); s += dpp_row<DPP_ROW_ROR + 4>(s, s); s += dpp_row<DPP_ROW_ROR + 2>(s, s); s += dpp_row<DPP_ROW_ROR + 1>(s, s);
		sc = s; nc = 1;
		if (l == 0) cg[0] = (uint32_t)l_query << 4;
	} else {
		int max_gap = ((l_query + 1) >> 1) - 5; // max_ins == max_del with o=6, e=1, a=1
		max_gap = max_gap > 1 ? max_gap : 1;
		int w = (max_gap + iabs(rlen - l_query) + 1) >> 1;
		w = w < w_ ? w : w_;
		const int min_w = iabs(rlen - l_query) + 3;
		w = w > min_w ? w : min_w;
		const int n_col = l_query < 2 * w + 1 ? l_query : 2 * w + 1;
		int n = 0;
		if (n_col > 16 * HI) return false;
		if (LO <= 1 && n_col <= 16) sc = nw_g16<1>(sg, w, z, zl, cg, cap, &n);
		else if (LO <= 2 && HI >= 2 && n_col <= 32) sc = nw_g16<(HI >= 2 ? 2 : HI)>(sg, w, z, zl, cg, cap, &n);
		else if (LO <= 4 && HI >= 4 && n_col <= 64) sc = nw_g16<(HI >= 4 ? 4 : HI)>(sg, w, z, zl, cg, cap, &n);
		else if (LO <= 8 && HI >= 8 && n_col <= 128) sc = nw_g16<(HI >= 8 ? 8 : HI)>(sg, w, z, zl, cg, cap, &n);
		else sc = nw_g16<HI>(sg, w, z, zl, cg, cap, &n);
		nc = g16_bcast0(n);
	}
	int nm = -1;
	if (nc <= cap) { // NM = mismatches + gap bases; a leading/trailing D is not counted (bwa.c:169-199)
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		// every lane reads the CIGAR; the bases of a match run are compared 16 at a time
		int x = 0, y = 0, n_mm = 0, n_gap = 0;
		for (int k = 0; k < nc; ++k) {
			const int op = cg[k] & 0xf, len = cg[k] >> 4;
			if (op == 0) { for (int i = l; i < len; i += 16) n_mm += sg.q[x + i] != sg.t[y + i] ? 1 : 0; x += len; y += len; }
			else if (op == 2) { if (k > 0 && k < nc - 1) n_gap += len; y += len; }
			else if (op == 1) { x += len; n_gap += len; }
		}
		n_mm += dpp_rowz<DPP_ROW_ROR + 8>(n_mm); n_mm += dpp_rowz<DPP_ROW_ROR + 4>(n_mm); n_mm += dpp_rowz<DPP_ROW_ROR + 2>(n_mm); n_mm += dpp_rowz<DPP_ROW_ROR + 1>(n_mm);
		nm = n_mm + n_gap;
	}
	*score = sc; *n_cigar = nc; *NM = nm;
	return true;
}

struct NwArgs {
	IndexView ix; const uint8_t *bases; const int32_t *base_off, *lens, *preg_off, *n_regs; int n_reads; const Reg *pregs;
	Aln *alns; uint32_t *cig; int cig_w; uint8_t *z; const int32_t *z_off; const int32_t *list; uint32_t *err;
	const int32_t *order; // queue positions in the order they are taken (one band class per launch)
	int32_t *punt, *n_punt; // queue positions whose retry wants a wider tiling than their class kernel holds; taken by the launch with all tilings
};

// mem_reg2aln for queued region list[i]; z_off in 64-byte units.  n_dev != null: the number of queued positions is read from the device
// (the launch behind the class launches that takes what they could not hold)
template <int LO, int HI>
__global__ void __launch_bounds__(64) k_reg2aln_nw_g16(NwArgs A, int n, const int32_t *n_dev)
{
	if (n_dev) n = *n_dev;
	__shared__ uint8_t lds_q[4][NW_Q_CAP];
	__shared__ uint8_t lds_t[4][NW_T_CAP];
	__shared__ __attribute__((aligned(16))) uint8_t lds_z[4][NW_ZL_BYTES];
	const int grp = threadIdx.x >> 4, l = threadIdx.x & 15;
	for (int it0 = blockIdx.x * 4 + grp; it0 < n; it0 += gridDim.x * 4) {
		const int it = A.order[it0];
		const int g = A.list[it];
		int lo = 0, hi = A.n_reads;
		while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (A.preg_off[mid] <= g) lo = mid; else hi = mid; }
		const int r = lo;
		const Reg ar = A.pregs[g];
		const int l_query = A.lens[r];
		const uint8_t *query = A.bases + A.base_off[r];
		const int qb = ar.qb, qe = ar.qe;
		const int64_t rb = ar.rb, re = ar.re, L = A.ix.l_pac;
		uint32_t *cg = A.cig + (size_t)g * A.cig_w;
		const int cap = A.cig_w;
		Aln a = Aln();
		a.cigar_off = g * A.cig_w;
		a.flag = ar.secondary >= 0 ? 0x100 : 0;
		int w2 = reg2aln_w0(ar);
		// stage the two segments once: every band retry reads the same sequences
		const bool rev = rb >= L;
		const bool ok = qe - qb > 0 && rb < re && !(rb < L && re > L) && re <= L << 1 && rb >= 0; // bwa_gen_cigar2's rejects (bwa.c:126-134)
		NwSeg sg; sg.q = lds_q[grp]; sg.t = lds_t[grp]; sg.qlen = qe - qb; sg.tlen = (int)(re - rb);
		if (ok) {
			for (int k = l; k < sg.qlen; k += 16) lds_q[grp][k] = rev ? query[qe - 1 - k] : query[qb + k];
			for (int k = l; k < sg.tlen; k += 16) lds_t[grp][k] = (uint8_t)ref_base(A.ix, rev ? re - 1 - k : rb + k);
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		int NM = -1, score = 0, n_cigar = 0, last_sc = -(1 << 30), i = 0;
		bool overflow = false, punted = false;
		uint8_t *z = A.z + (size_t)A.z_off[it] * 64;
		do {
			w2 = w2 < OPT_W << 2 ? w2 : OPT_W << 2;
			n_cigar = 0; NM = -1;
			if (ok && !gen_cigar2_g16<LO, HI>(sg, w2, z, lds_z[grp], cg + 1, cap - 2, &score, &n_cigar, &NM)) { punted = true; break; } // (cg + 1: room for both clips)
			if (n_cigar > cap - 2) { overflow = true; break; }
			if (score == last_sc || w2 == OPT_W << 2) break;
			last_sc = score;
			w2 <<= 1;
		} while (++i < 3 && score < ar.truesc - OPT_A);
		if (punted) { if (l == 0) A.punt[atomicAdd(A.n_punt, 1)] = it; __builtin_amdgcn_wave_barrier(); continue; }
		if (overflow) { if (l == 0) atomicOr(A.err, ERR_CIGAR_OVERFLOW); continue; }
		if (l == 0) {
			a.NM = NM;
			int is_rev;
			int64_t pos = depos(A.ix, rb < L ? rb : re - 1, &is_rev);
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
			if (c0 != cg) for (int k = 0; k < n_cigar; ++k) cg[k] = c0[k];
			a.n_cigar = n_cigar;
			a.rid = pos2rid(A.ix, pos);
			a.pos = pos - A.ix.ann_off[a.rid];
			a.score = ar.score; a.sub = ar.sub > ar.csub ? ar.sub : ar.csub;
			a.is_alt = ar.is_alt; a.alt_sc = ar.alt_sc;
			A.alns[g] = a;
		}
		__builtin_amdgcn_wave_barrier(); // the LDS rows are reused by the group's next region
	}
}

} // namespace arx
