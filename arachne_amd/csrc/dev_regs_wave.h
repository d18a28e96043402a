// dev_regs_wave.h -- the rescue replay of ONE read pair by a whole 64-lane wavefront (gfx950 only; arx_cold.hip: k_rescue_heavy).
//
// dev_regs.h runs GoBwaMemMateSW's two rescue loops (gobwa.go:285-324 around mem_matesw, bwamem_pair.c:111-180) with one thread per
// pair.  A pair from a high-copy repeat carries 100-190 regions per read: its thread walks 50 anchors over the mate list, and every
// rescued region costs a scan and a shift of that list -- a million or more dependent instructions in ONE lane while the rest of the
// chip waits for the launch to end (10-14 ms per round at GRCh38 size; staging the lists in LDS alone took 13 % off: the chain is
// instructions, not memory).  Here the same state machine runs with every lane executing the same control flow on the lists in LDS:
//   * the scans (is a mate region already in the window?  what does the new region meet in the list?) take one list entry per lane, the
//     answers are ballots and wave reductions;
//   * the order-dependent part of mem_sort_dedup_patch for one inserted region is settled from those masks exactly as dedup_insert()
//     of dev_regs.h does it (who is met first on either side, where the reference's loop stops);
//   * the shift that makes room for the region moves 64 records per step;
//   * a tie with a listed region takes the general pass, also by the wavefront (w_sort_dedup: klib's introsort reproduced by w_introsort, the
//     redundancy loop on bit masks); what is rarer still (regions that go on an insertion, lists beyond 256 regions) is run by lane 0 with
//     the serial code of dev_regs.h.
// Results are those of rescue_step() bit for bit (GPU tests on repeat-rich workloads, tests/test_config_shapes.py).
#pragma once
#include "dev_regs.h"

namespace arx {

#ifdef ARX_WAVE_STATS
__device__ unsigned long long g_wstat[24];
#define ARX_WSTAT(k) do { if (threadIdx.x == 0) atomicAdd(&g_wstat[k], 1ull); } while (0)
#define ARX_WT0() const unsigned long long wt0_ = wall_clock64()
#define ARX_WT(acc) (acc) += wall_clock64() - wt0_
#else
#define ARX_WSTAT(k) do {} while (0)
#define ARX_WT0() do {} while (0)
#define ARX_WT(acc) do {} while (0)
#endif
struct WTimes { unsigned long long fast, general, skip, enumerate; };
__device__ __shared__ WTimes w_times;

__device__ __forceinline__ int64_t w_max_i64(int64_t v)
{
	for (int d = 32; d > 0; d >>= 1) { const int64_t o = (int64_t)((uint64_t)__shfl_xor((int)((uint64_t)v >> 32), d, 64) << 32 | (uint32_t)__shfl_xor((int)v, d, 64)); v = o > v ? o : v; }
	return v;
}
__device__ __forceinline__ int64_t w_min_i64(int64_t v) { return -w_max_i64(-v); }
__device__ __forceinline__ int w_min_i32(int v) { for (int d = 32; d > 0; d >>= 1) { const int o = __shfl_xor(v, d, 64); v = o < v ? o : v; } return v; }

__device__ bool w_rescue_skipped(const IndexView &ix, const Reg &a, const Reg *ma, int n_ma) // bwamem_pair.c:118-124
{
	bool hit = false;
	for (int j = threadIdx.x; j < n_ma; j += 64) {
		int64_t dist;
		const int r = infer_dir(ix.l_pac, a.rb, ma[j].rb, &dist);
		hit = hit || (r == 1 && dist >= PES_LOW && dist <= PES_HIGH);
	}
	return __ballot(hit) != 0;
}

// rescue_enumerate() of dev_regs.h: out == nullptr counts and computes *mask, otherwise *mask says which anchors get a task
__device__ int w_rescue_enumerate(const IndexView &ix, int pair, const int *lens2, Reg *const regs[2], const int n_regs[2], const ResState &st, uint64_t *mask,
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
		else if (w_rescue_skipped(ix, a, regs[o], n_regs[o])) continue;
		int64_t rb, re;
		if (!rescue_window(ix, a, l_ms, &rb, &re)) continue;
		if (!out) *mask |= (uint64_t)1 << k;
		if (out && threadIdx.x == 0) { SwTask t; t.rb = rb; t.re = re; t.pair = pair; t.o = o; t.slot = slot0 + cnt; t.pad = 0; out[cnt] = t; }
		++cnt;
	}
	return cnt;
}

#define W_SORT_MAX 256
// klib's ks_introsort (arx_dev.h) on an index array of up to W_SORT_MAX entries in LDS, by the whole wavefront, with klib's result: the
// permutation depends on how the algorithm orders equal keys, so every step is reproduced, only faster.
//   * One partition step (median of three to the end of the range, then `do ++i while (a[i] < rp)` / `do --j while (i <= j && rp < a[j])`
//     / swap) visits every position of the range once: i stops at the positions whose key is not below the pivot ("left stops", ascending),
//     j at those whose key is not above it ("right stops", descending); the k-th left stop is swapped with the k-th right stop as long as
//     it lies left of it, and swapped positions are never looked at again.  So the lanes classify the range's positions side by side,
//     ballots rank the stops, m = the number of pairs in order, the m swaps are independent, and the scan ends at
//     i = min(left stop m+1, right stop m) (the value swapped into right stop m stops i as well); the pivot goes to i.  The first
//     position of the range is never examined (i starts at s + 1), exactly as in klib.
//   * Ranges of 16 or fewer are left to the final insertion sort, which is a stable sort of the whole array: every lane counts, for its
//     elements, the elements with a smaller key plus the equal ones before it.
//   * The range stack and the depth budget (comb sort when it runs out: lane 0, as rare as in klib) are wave-uniform scalars.
// lpos / rpos: scratch, W_SORT_MAX ints of LDS each.
template <int MAXN, class LT> __device__ void w_introsort(int n, int *a, LT lt, int *lpos, int *rpos) // n <= MAXN (a multiple of 64)
{
	const int lane = threadIdx.x;
	if (n < 1) return;
	if (n == 2) { if (lane == 0 && lt(a[1], a[0])) { const int x = a[0]; a[0] = a[1]; a[1] = x; } __syncthreads(); return; }
	int st_l[24], st_r[24], st_d[24], top = 0; // the longer side of a partition is pushed (if longer than 16), the shorter one continued with: depth <= log2(n / 16) + 1
	int d, s = 0, t = n - 1;
	for (d = 2; (1 << d) < n; ++d) {}
	d <<= 1;
	for (;;) {
		if (s < t) {
			--d;
			if (d == 0) {
				if (lane == 0) ks_combsort(t - s + 1, a + s, lt);
				__syncthreads();
				t = s;
				continue;
			}
			int k = s + ((t - s) >> 1) + 1;
			{
				const int vk = a[k], vi = a[s], vj = a[t];
				if (lt(vk, vi)) { if (lt(vk, vj)) k = t; }
				else k = lt(vj, vi) ? s : t;
			}
			const int rp = a[k];
			__syncthreads(); // every lane has read a[k], a[t]
			if (k != t && lane == 0) { a[k] = a[t]; a[t] = rp; }
			__syncthreads();
			int n_l = 0, n_r = 0;
			for (int x0 = s + 1; x0 <= t; x0 += 64) {
				const int x = x0 + lane;
				bool lf = false, rf = false;
				if (x <= t) { const int v = a[x]; lf = !lt(v, rp); rf = x < t && !lt(rp, v); }
				const uint64_t bl = __ballot(lf), br = __ballot(rf), below = (1ull << lane) - 1;
				if (lf) lpos[n_l + __builtin_popcountll(bl & below)] = x;
				if (rf) rpos[n_r + __builtin_popcountll(br & below)] = x;
				n_l += __builtin_popcountll(bl); n_r += __builtin_popcountll(br);
			}
			__syncthreads();
			int m = 0; // pairs (k-th left stop, k-th right stop from the top) that are in order: a prefix
			const int n_p = n_l < n_r ? n_l : n_r;
			for (int k0 = 0; k0 < n_p; k0 += 64) {
				const int kk = k0 + lane;
				const bool ok = kk < n_p && lpos[kk] < rpos[n_r - 1 - kk];
				const uint64_t b = __ballot(ok);
				m += __builtin_popcountll(b);
				if (b != ~0ull) break;
			}
			int i = lpos[m];
			if (m > 0) { const int rm = rpos[n_r - m]; i = rm < i ? rm : i; }
			__syncthreads();
			for (int kk = lane; kk < m; kk += 64) { const int x = lpos[kk], y = rpos[n_r - 1 - kk], v = a[x]; a[x] = a[y]; a[y] = v; }
			__syncthreads();
			if (lane == 0) { const int v = a[i]; a[i] = a[t]; a[t] = v; }
			__syncthreads();
			if (i - s > t - i) {
				if (i - s > 16) { st_l[top] = s; st_r[top] = i - 1; st_d[top] = d; ++top; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { st_l[top] = i + 1; st_r[top] = t; st_d[top] = d; ++top; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else if (top == 0) break;
		else { --top; s = st_l[top]; t = st_r[top]; d = st_d[top]; }
	}
	// ks_insertsort over the whole array = its stable sort
	int mine[MAXN / 64], dest[MAXN / 64];
#pragma unroll
	for (int u = 0; u < MAXN / 64; ++u) {
		const int x = u * 64 + lane;
		mine[u] = -1; dest[u] = 0;
		if (x < n) {
			const int v = a[x];
			int cnt = 0;
			for (int y = 0; y < n; ++y) { const int w = a[y]; cnt += (lt(w, v) || (y < x && !lt(v, w))) ? 1 : 0; }
			mine[u] = v; dest[u] = cnt;
		}
	}
	__syncthreads();
#pragma unroll
	for (int u = 0; u < MAXN / 64; ++u) if (mine[u] >= 0) a[dest[u]] = mine[u];
	__syncthreads();
}

// LDS scratch of the wave's general pass (lists of up to W_SORT_MAX regions)
struct WaveScratch { int idx[W_SORT_MAX]; uint64_t R[W_SORT_MAX][4], S[W_SORT_MAX][4]; }; // R / S double as w_introsort's scratch (not live during a sort)

// ma[at..n) one place up, 64 records per step from the top, then ma[at] = b
__device__ void w_insert_at(Reg *ma, int n, int at, const Reg &b)
{
	const int lane = threadIdx.x;
	for (int hi = n; hi > at; hi -= 64) {
		const int lo = hi - 64 > at ? hi - 64 : at, j = lo + lane;
		Reg r = Reg();
		if (j < hi) r = ma[j];
		__syncthreads();
		if (j < hi) ma[j + 1] = r;
		__syncthreads();
	}
	if (lane == 0) ma[at] = b;
	__syncthreads();
}

// a[i] = a[idx[i]] for i < n through tmp (HBM), the lanes side by side
__device__ void w_permute(int n, Reg *a, Reg *tmp, const int *idx)
{
	for (int i = threadIdx.x; i < n; i += 64) tmp[i] = a[idx[i]];
	__syncthreads();
	for (int i = threadIdx.x; i < n; i += 64) a[i] = tmp[i];
	__syncthreads();
}

// keep[rd] bit l: region rd * 64 + l stays; the others are squeezed out, order kept.  Returns the new length.
__device__ int w_squeeze(int n, Reg *a, Reg *tmp, const uint64_t keep[4])
{
	const int lane = threadIdx.x;
	int m = 0;
	for (int rd = 0; rd * 64 < n; ++rd) {
		const uint64_t k = keep[rd];
		if (k >> lane & 1) tmp[m + __builtin_popcountll(k & ((1ull << lane) - 1))] = a[rd * 64 + lane];
		m += __builtin_popcountll(k);
	}
	__syncthreads();
	for (int i = lane; i < m; i += 64) a[i] = tmp[i];
	__syncthreads();
	return m;
}

// the sort of the general pass: w_introsort; ARX_WSORT_SERIAL (A/B builds): lane 0 running ks_introsort; ARX_WSORT_CHECK (test builds): both,
// and a count of the sorts whose results differ (arx_cold.hip prints it after every launch) -- neither is part of the product build
#if defined(ARX_WSORT_SERIAL)
#define W_SORT(n, idx, lt, ws) do { if (threadIdx.x == 0) ks_introsort((n), (idx), (lt)); __syncthreads(); } while (0)
#elif defined(ARX_WSORT_CHECK)
__device__ unsigned long long g_wsort_bad[2];
#define W_SORT(n, idx, lt, ws) do { int *chk_ = (int *)(ws).S; for (int q_ = threadIdx.x; q_ < (n); q_ += 64) chk_[q_] = (idx)[q_]; __syncthreads(); \
	if (threadIdx.x == 0) ks_introsort((n), chk_, (lt)); __syncthreads(); \
	w_introsort<W_SORT_MAX>((n), (idx), (lt), (int *)(ws).R, (int *)(ws).R + W_SORT_MAX); \
	bool bad_ = false; for (int q_ = threadIdx.x; q_ < (n); q_ += 64) bad_ = bad_ || chk_[q_] != (idx)[q_]; \
	if (threadIdx.x == 0) atomicAdd(&g_wsort_bad[1], 1ull); if (__ballot(bad_) && threadIdx.x == 0) atomicAdd(&g_wsort_bad[0], 1ull); __syncthreads(); } while (0)
#else
#define W_SORT(n, idx, lt, ws) w_introsort<W_SORT_MAX>((n), (idx), (lt), (int *)(ws).R, (int *)(ws).R + W_SORT_MAX)
#endif

// sort_dedup_patch() of dev_regs.h for n <= W_SORT_MAX, without patching: either because there is none (query == nullptr, the mate-rescue
// call site: patch_l_pac < 0) or because no pair of regions gets past mem_patch_reg's geometric tests (patch_l_pac = l_pac; -2 is returned
// if one does, before anything but the first sort has happened to the list in LDS).  The two
// introsorts are w_introsort on an index array in LDS (their order of equal keys is part of the result); the rest is spread over the
// lanes as well: for the redundancy pass every lane walks the earlier neighbours of its own regions and notes which are
// redundant with it (R) and which of those score higher (S) -- geometry only, independent of who has been dropped -- and the
// reference's loop (bwamem.c:443-473) is then replayed on the masks: region i drops its alive redundant neighbours from the nearest
// down, until one that scores higher drops i instead.
__device__ int w_sort_dedup(int n, Reg *a, Reg *tmp, WaveScratch &ws, int64_t patch_l_pac = -1)
{
	const int lane = threadIdx.x;
	if (n <= 1) return n;
	for (int i = lane; i < n; i += 64) ws.idx[i] = i;
	__syncthreads();
	{ RegReLt lt; lt.r = a; W_SORT(n, ws.idx, lt, ws); }
	w_permute(n, a, tmp, ws.idx);
	uint64_t A[4] = {0, 0, 0, 0}; // regions that can still be met as the earlier one of a pair
	bool may_patch = false;
	for (int rd = 0; rd < 4; ++rd) {
		const int i = rd * 64 + lane;
		bool alive = false;
		if (i < n) {
			a[i].n_comp = 1;
			const RegHead p = *(const RegHead *)(a + i);
			alive = p.qe != p.qb;
			ws.R[i][0] = ws.R[i][1] = ws.R[i][2] = ws.R[i][3] = 0;
			ws.S[i][0] = ws.S[i][1] = ws.S[i][2] = ws.S[i][3] = 0;
			for (int j = i - 1; j >= 0; --j) {
				const RegHead q = *(const RegHead *)(a + j);
				if (!(p.rid == q.rid && p.rb < q.re + OPT_MAX_CHAIN_GAP)) break;
				const int64_t orr = q.re - p.rb, oq = q.qb < p.qb ? q.qe - p.qb : p.qe - q.qb;
				const int64_t mr = q.re - q.rb < p.re - p.rb ? q.re - q.rb : p.re - p.rb, mq = q.qe - q.qb < p.qe - p.qb ? q.qe - q.qb : p.qe - p.qb;
				if ((float)orr > OPT_MASK_LEVEL_REDUN * (float)mr && (float)oq > OPT_MASK_LEVEL_REDUN * (float)mq) {
					ws.R[i][j >> 6] |= 1ull << (j & 63);
					if (p.score < q.score) ws.S[i][j >> 6] |= 1ull << (j & 63);
				} else if (patch_l_pac >= 0 && q.rb < p.rb) { // could mem_patch_reg(q, p) get as far as its alignment?  (bwamem.c:406-421)
					if (!(q.rb < patch_l_pac && p.rb >= patch_l_pac) && !(q.qb >= p.qb || q.qe >= p.qe || q.re >= p.re)) {
						int w = (int)((q.re - p.rb) - (q.qe - p.qb));
						w = w > 0 ? w : -w;
						double r = (double)(q.re - p.rb) / (double)(p.re - q.rb) - (double)(q.qe - p.qb) / (double)(p.qe - q.qb);
						r = r > 0. ? r : -r;
						if (q.re < p.rb || q.qe < p.qb) { if (!(w > OPT_W << 1 || r >= (double)0.05f)) may_patch = true; }
						else if (!(w > OPT_W << 2 || r >= (double)(0.05f * 2))) may_patch = true;
					}
				}
			}
		}
		A[rd] = __ballot(alive);
	}
	if (__ballot(may_patch)) return -2; // the caller takes the one-thread pass on the untouched list
	__syncthreads();
	for (int i = 1; i < n; ++i) { // every lane replays the same loop on the masks
		uint64_t c[4], st = 0;
		int js = -1;
#pragma unroll
		for (int wd = 3; wd >= 0; --wd) {
			c[wd] = ws.R[i][wd] & A[wd];
			st = c[wd] & ws.S[i][wd];
			if (js < 0 && st) js = wd * 64 + 63 - __builtin_clzll(st);
		}
		if (js < 0) {
#pragma unroll
			for (int wd = 0; wd < 4; ++wd) A[wd] &= ~c[wd];
		} else {
#pragma unroll
			for (int wd = 0; wd < 4; ++wd) {
				const int lo = js + 1 - wd * 64; // bits >= lo of this word lie above js
				const uint64_t above = lo <= 0 ? ~0ull : lo >= 64 ? 0ull : ~0ull << lo;
				A[wd] &= ~(c[wd] & above);
				if ((i >> 6) == wd) A[wd] &= ~(1ull << (i & 63));
			}
		}
	}
	uint64_t keep[4];
	for (int rd = 0; rd < 4; ++rd) {
		const int i = rd * 64 + lane;
		keep[rd] = __ballot(i < n && (A[rd] >> lane & 1) && a[i].qe > a[i].qb);
	}
	n = w_squeeze(n, a, tmp, keep);
	for (int i = lane; i < n; i += 64) ws.idx[i] = i;
	__syncthreads();
	{ RegScoreLt lt; lt.r = a; W_SORT(n, ws.idx, lt, ws); }
	w_permute(n, a, tmp, ws.idx);
	for (int rd = 0; rd < 4; ++rd) { // identical (score, rb, qb) as the region before: dropped; region 0 always stays (bwamem.c:480-487)
		const int i = rd * 64 + lane;
		bool k = false;
		if (i < n) {
			k = a[i].qe > a[i].qb;
			if (i > 0) k = k && !(a[i].score == a[i - 1].score && a[i].rb == a[i - 1].rb && a[i].qb == a[i - 1].qb);
			else k = true;
		}
		keep[rd] = __ballot(k);
	}
	return w_squeeze(n, a, tmp, keep);
}

// dedup_insert() of dev_regs.h with the scan spread over the lanes.  ma: LDS, n <= 256 regions, room for one more.  Returns the new
// length, or -1 when the outcome would depend on introsort's order of equal keys (the caller takes the general pass).
__device__ int w_dedup_insert(const Reg &b_in, Reg *ma, int n, Reg *tmp)
{
	const int lane = threadIdx.x;
	Reg b = b_in;
	b.n_comp = 1;
	uint64_t E[4] = {0, 0, 0, 0}, L[4] = {0, 0, 0, 0};
	int s1 = -1, s2 = -1, at = n;
	int64_t s1_re = 0, s2_re = 0;
	for (int rd = 0; rd * 64 < n; ++rd) {
		const int j = rd * 64 + lane;
		const bool valid = j < n;
		RegHead q = RegHead();
		if (valid) q = *(const RegHead *)(ma + j);
		const bool tie = valid && (q.re == b.re || (q.score == b.score && q.rb == b.rb && q.qb == b.qb));
		if (__ballot(tie)) { ARX_WSTAT(3); return -1; }
		const bool before = q.score > b.score || (q.score == b.score && (q.rb < b.rb || (q.rb == b.rb && q.qb < b.qb)));
		const int at_r = w_min_i32(valid && !before ? j : 0x7fffffff);
		if (at_r < at) at = at_r;
		bool ef = false, lf = false, st1 = false, st2 = false;
		if (valid && q.rid == b.rid) {
			if (q.re < b.re) {
				if (b.rb < q.re + OPT_MAX_CHAIN_GAP) {
					const int64_t orr = q.re - b.rb, oq = q.qb < b.qb ? q.qe - b.qb : b.qe - q.qb;
					const int64_t mr = q.re - q.rb < b.re - b.rb ? q.re - q.rb : b.re - b.rb, mq = q.qe - q.qb < b.qe - b.qb ? q.qe - q.qb : b.qe - b.qb;
					ef = (float)orr > OPT_MASK_LEVEL_REDUN * (float)mr && (float)oq > OPT_MASK_LEVEL_REDUN * (float)mq;
					st1 = ef && b.score < q.score;
				}
			} else if (q.rb < b.re + OPT_MAX_CHAIN_GAP) {
				const int64_t orr = b.re - q.rb, oq = b.qb < q.qb ? b.qe - q.qb : q.qe - b.qb;
				const int64_t mr = b.re - b.rb < q.re - q.rb ? b.re - b.rb : q.re - q.rb, mq = b.qe - b.qb < q.qe - q.qb ? b.qe - b.qb : q.qe - q.qb;
				lf = (float)orr > OPT_MASK_LEVEL_REDUN * (float)mr && (float)oq > OPT_MASK_LEVEL_REDUN * (float)mq;
				st2 = lf && !(q.score < b.score);
			}
		}
		E[rd] = __ballot(ef); L[rd] = __ballot(lf);
		if (__ballot(st1)) { // the stopper of the earlier side: largest re, the smaller index among equals
			const int64_t mre = w_max_i64(st1 ? q.re : (int64_t)0x8000000000000000ll);
			const int mj = w_min_i32(st1 && q.re == mre ? j : 0x7fffffff);
			if (s1 < 0 || mre > s1_re) { s1 = mj; s1_re = mre; }
		}
		if (__ballot(st2)) { // later side: smallest re, the smaller index among equals
			const int64_t mre = w_min_i64(st2 ? q.re : (int64_t)0x7fffffffffffffffll);
			const int mj = w_min_i32(st2 && q.re == mre ? j : 0x7fffffff);
			if (s2 < 0 || mre < s2_re) { s2 = mj; s2_re = mre; }
		}
	}
	// the regions that go (uniform: every lane walks the same masks; the lists of such regions are short)
	bool b_gone = s1 >= 0;
	uint64_t G[4] = {0, 0, 0, 0};
	int n_gone = 0;
	for (int wd = 0; wd < 4; ++wd)
		for (uint64_t m = E[wd]; m; m &= m - 1) {
			const int j = wd * 64 + __builtin_ctzll(m);
			if (s1 >= 0) { const int64_t re = ma[j].re; if (!(re > s1_re || (re == s1_re && j < s1))) continue; }
			G[wd] |= 1ull << (j & 63); ++n_gone;
		}
	if (!b_gone) {
		b_gone = s2 >= 0;
		for (int wd = 0; wd < 4; ++wd)
			for (uint64_t m = L[wd]; m; m &= m - 1) {
				const int j = wd * 64 + __builtin_ctzll(m);
				if (s2 >= 0) { const int64_t re = ma[j].re; if (!(re < s2_re || (re == s2_re && j < s2))) continue; }
				G[wd] |= 1ull << (j & 63); ++n_gone;
			}
	}
	if (n_gone == 0) {
		for (int j = lane; j < n; j += 64) ma[j].n_comp = 1;
		__syncthreads();
		ARX_WSTAT(1);
		if (b_gone) return n;
		w_insert_at(ma, n, at, b);
		return n + 1;
	}
	ARX_WSTAT(2);
	// regions go: lane 0 finishes as dedup_insert() does (the lists of such cases are short on the benchmark workloads)
	int m = 0;
	if (lane == 0) {
		for (int wd = 0; wd < 4; ++wd)
			for (uint64_t g = G[wd]; g; g &= g - 1) { Reg &x = ma[wd * 64 + __builtin_ctzll(g)]; x.qe = x.qb; }
		bool placed = b_gone;
		for (int i = 0; i < n; ++i) {
			const Reg &x = ma[i];
			if (!(x.qe > x.qb)) continue;
			if (!placed && !(x.score > b.score || (x.score == b.score && (x.rb < b.rb || (x.rb == b.rb && x.qb < b.qb))))) { tmp[m++] = b; placed = true; }
			tmp[m] = x; tmp[m].n_comp = 1; ++m;
		}
		if (!placed) tmp[m++] = b;
		for (int i = 0; i < m; ++i) ma[i] = tmp[i];
	}
	m = __shfl(m, 0);
	__syncthreads();
	return m;
}

// matesw_apply() of dev_regs.h, wave-uniform
__device__ int w_matesw_apply(const IndexView &ix, const Reg &a, int l_ms, const U8Res &aln, int64_t rb, Reg *ma, int n_ma, Reg *tmp, int *idx, int32_t *clean, int32_t *fresh, WaveScratch &ws)
{
	const int64_t l_pac = ix.l_pac;
	const bool inserts = aln.score >= OPT_MIN_SEED_LEN && aln.qb >= 0;
	if (!inserts && *clean) return n_ma;
	*fresh = 0;
	ARX_WSTAT(inserts ? 0 : 5);
	if (inserts) {
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
		if (*clean == 2 && n_ma >= 1 && n_ma <= 256) {
			ARX_WT0();
			const int m = w_dedup_insert(b, ma, n_ma, tmp);
			ARX_WT(w_times.fast);
			if (m >= 0) return m;
		} else ARX_WSTAT(*clean == 2 ? 6 : 4);
		{ // the general pass starts from the list with b put in by score (bwamem_pair.c:168-173)
			bool lower = false;
			int at = n_ma;
			for (int rd = 0; rd * 64 < n_ma && at == n_ma; ++rd) {
				const int j = rd * 64 + threadIdx.x;
				lower = j < n_ma && ma[j].score < b.score;
				const uint64_t bal = __ballot(lower);
				if (bal) at = rd * 64 + __builtin_ctzll(bal);
			}
			w_insert_at(ma, n_ma, at, b);
		}
		++n_ma;
	}
	*clean = n_ma >= 2 ? 2 : 1;
	int m = 0;
	ARX_WT0();
	if (n_ma <= W_SORT_MAX) m = w_sort_dedup(n_ma, ma, tmp, ws);
	else {
		if (threadIdx.x == 0) m = sort_dedup_patch(ix, 0, n_ma, ma, tmp, idx, 0);
		m = __shfl(m, 0);
		__syncthreads();
	}
	ARX_WT(w_times.general);
	return m;
}

// rescue_step() of dev_regs.h, wave-uniform.  regs: LDS; n_loc[e]: the lists' lengths (kept by the caller, written back at the end)
__device__ bool w_rescue_step(const IndexView &ix, int pair, const int *lens2, Reg *const regs[2], int n_loc[2], Reg *const tmp[2], int *const idx[2],
                              ResState &st, const U8Res *sres, const SwEmit &emit, WaveScratch &ws)
{
	const int lane = threadIdx.x;
	for (;;) {
		if (st.phase == 2) return false;
		if (st.phase == 3) {
			uint64_t mask = 0;
			int cnt = emit.no_ahead ? 0 : w_rescue_enumerate(ix, pair, lens2, regs, n_loc, st, &mask, nullptr, 0);
			if (emit.no_ahead) mask = 0;
			st.spec_mask = mask; st.spec_off = 0; st.i = 0; st.num = 0; st.phase = 0;
			st.mask_fresh = emit.no_ahead ? 0 : 1;
			if (cnt > 0) {
				int so = 0, to = 0;
				if (lane == 0) { so = ARX_ATOMIC_ADD(emit.n_slots, cnt); to = ARX_ATOMIC_ADD(emit.n_tasks, cnt); }
				so = __shfl(so, 0); to = __shfl(to, 0);
				st.spec_off = so;
				w_rescue_enumerate(ix, pair, lens2, regs, n_loc, st, &mask, emit.tasks + to, so);
				return true;
			}
		}
		const int e = st.e, o = 1 - e;
		if (st.phase == 1) {
			const Reg a1 = regs[e][st.i];
			n_loc[o] = w_matesw_apply(ix, a1, lens2[o], sres[emit.single_slot], st.rb, regs[o], n_loc[o], tmp[o], idx[o], &st.clean[o], &st.mask_fresh, ws);
			st.phase = 0; ++st.i;
		}
		if (st.i >= st.n_snap || st.num >= MAX_RESCUE || lens2[o] <= 0) {
			if (e == 1) { st.e = 0; st.n_snap = n_loc[0]; st.phase = 3; continue; }
			st.phase = 2;
			return false;
		}
		const Reg a = regs[e][st.i];
		if (a.score < st.best[e] - 25) { ++st.i; continue; }
		const int k = st.num++;
		if (st.mask_fresh && !(st.spec_mask >> k & 1)) { ++st.i; continue; }
		{ ARX_WT0(); const bool sk = !st.mask_fresh && w_rescue_skipped(ix, a, regs[o], n_loc[o]); ARX_WT(w_times.skip); if (sk) { ++st.i; continue; } }
		int64_t rb, re;
		if (!rescue_window(ix, a, lens2[o], &rb, &re)) { ++st.i; continue; }
		if (st.spec_mask >> k & 1) {
			const int slot = st.spec_off + __builtin_popcountll(st.spec_mask & (((uint64_t)1 << k) - 1));
			n_loc[o] = w_matesw_apply(ix, a, lens2[o], sres[slot], rb, regs[o], n_loc[o], tmp[o], idx[o], &st.clean[o], &st.mask_fresh, ws);
			++st.i;
			continue;
		}
		st.rb = rb; st.re = re; st.phase = 1;
		if (lane == 0) {
			SwTask t; t.rb = rb; t.re = re; t.pair = pair; t.o = o; t.slot = emit.single_slot; t.pad = 0;
			emit.tasks[ARX_ATOMIC_ADD(emit.n_tasks, 1)] = t;
			ARX_ATOMIC_INC(emit.n_slots + 1);
		}
		return true;
	}
}

} // namespace arx
