// arx_cold.hip -- the two list-bookkeeping kernels (region de-duplication, rescue state machine) as their own translation unit
// so that the library's units compile side by side.  Until round 2 this unit was built at -O1: hipcc 7.2 at -O2/-O3 emitted
// a dedup kernel that never terminated on gfx950.  The cause was bisected to the optimised body of ks_introsort (arx_dev.h has the
// record and the source form that terminates); the unit is -O3 like the rest.
#ifdef ARX_CHAIN_STATS // diagnostics build (-DARX_CHAIN_STATS): per-phase clock of the heavy-read chaining kernel, printed per launch
#include <hip/hip_runtime.h>
__shared__ unsigned long long lds_cstat[8];
__device__ unsigned long long g_cstat[24];
#define ARX_CHAIN_T(k) do { lds_cstat[k] = wall_clock64(); } while (0)
#endif
#include "../../include/arachne_amd.h"
#include "hip_rt.h"
#include "pipeline.h"
#include "dev_regs_wave.h"
#include "dev_chain_wave.h"
#ifdef ARX_WSORT_CHECK
static void wsort_report(hipStream_t st, const char *who) { unsigned long long h[2]; hipStreamSynchronize(st); hipMemcpyFromSymbol(h, HIP_SYMBOL(arx::g_wsort_bad), sizeof h); fprintf(stderr, "wsort check after %s: %llu sorts so far, %llu differ from ks_introsort\n", who, h[1], h[0]); }
#else
static void wsort_report(hipStream_t, const char *) {}
#endif

namespace arx {
// One heavy pair per 64-lane workgroup: the lanes copy both region lists into LDS, the wavefront replays the rescue state machine on
// them (w_rescue_step of dev_regs_wave.h; wave = 0: lane 0 alone runs rescue_step(), kept for A/B runs), the lanes copy the lists back.  The scratch lists of the general dedup pass (ptmp, pidx) stay in HBM.
// Launched once per LDS footprint (cap_lo < capacity of both lists <= cap_hi records of dynamic LDS): a pair of 60 + 60 regions takes 15 KB
// where the longest ones take 58 KB, and the workgroups a CU holds -- two at the largest footprint -- are what this latency-bound kernel
// scales with.  A launch walks the whole list through its own cursor and skips the other launches' pairs.
static __global__ void __launch_bounds__(64) k_rescue_heavy(KRescueStep f, const int32_t *list, int n, int wave, int cap_lo, int cap_hi, int32_t *cursor)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_rescue[];
	Reg *lds_regs = (Reg *)lds_rescue;
	__shared__ int new_n[2];
	__shared__ WaveScratch ws;
	__shared__ int next_h;
	const int lane = threadIdx.x;
	for (;;) {
		if (lane == 0) next_h = atomicAdd(cursor, 1);
		__syncthreads();
		const int h = next_h;
		__syncthreads();
		if (h >= n) break; // every workgroup reaches this: the cursor only grows
		const int p = list[h];
		ResState st = f.state[p];
		if (st.phase == 2) continue; // uniform
		const int o0 = f.preg_off[2 * p], o1 = f.preg_off[2 * p + 1], o2 = f.preg_off[2 * p + 2];
		const int c0 = o1 - o0, c1 = o2 - o1; // capacities (KPairCap); c0 + c1 <= RESCUE_LDS_REGS by construction of the list
		if (c0 + c1 <= cap_lo || c0 + c1 > cap_hi) continue; // another launch's pair
		const int n0 = f.n_regs[2 * p], n1 = f.n_regs[2 * p + 1];
		{ // in: the live entries of both lists, word by word
			const uint32_t *s0 = (const uint32_t *)(f.pregs + o0), *s1 = (const uint32_t *)(f.pregs + o1);
			uint32_t *d0 = (uint32_t *)lds_regs, *d1 = (uint32_t *)(lds_regs + c0);
			for (int k = lane; k < n0 * (int)(sizeof(Reg) / 4); k += 64) d0[k] = s0[k];
			for (int k = lane; k < n1 * (int)(sizeof(Reg) / 4); k += 64) d1[k] = s1[k];
		}
		__syncthreads();
#ifdef ARX_WAVE_STATS
		const unsigned long long tb0 = wall_clock64();
		if (lane == 0) { w_times.fast = w_times.general = w_times.skip = w_times.enumerate = 0; }
		__syncthreads();
#endif
		if (wave) { // every lane runs the same control flow (dev_regs_wave.h)
			Reg *rg[2] = { lds_regs, lds_regs + c0 };
			Reg *tm[2] = { f.ptmp + o0, f.ptmp + o1 };
			int *ix2[2] = { f.pidx + o0, f.pidx + o1 };
			int nl[2] = { n0, n1 };
			SwEmit em; em.tasks = f.tasks; em.n_tasks = f.n_tasks; em.n_slots = f.n_slots; em.single_slot = f.single_base + p; em.no_ahead = f.no_ahead;
			w_rescue_step(f.ix, p, f.lens + 2 * p, rg, nl, tm, ix2, st, f.res, em, ws);
			if (lane == 0) { f.state[p] = st; f.n_regs[2 * p] = nl[0]; f.n_regs[2 * p + 1] = nl[1]; new_n[0] = nl[0]; new_n[1] = nl[1]; }
		} else if (lane == 0) {
			Reg *rg[2] = { lds_regs, lds_regs + c0 };
			Reg *tm[2] = { f.ptmp + o0, f.ptmp + o1 };
			int *ix2[2] = { f.pidx + o0, f.pidx + o1 };
			int *nr[2] = { f.n_regs + 2 * p, f.n_regs + 2 * p + 1 };
			SwEmit em; em.tasks = f.tasks; em.n_tasks = f.n_tasks; em.n_slots = f.n_slots; em.single_slot = f.single_base + p; em.no_ahead = f.no_ahead;
			rescue_step(f.ix, p, f.lens + 2 * p, rg, nr, tm, ix2, st, f.res, em);
			f.state[p] = st;
			new_n[0] = *nr[0]; new_n[1] = *nr[1];
		}
#ifdef ARX_WAVE_STATS
		if (lane == 0) { const unsigned long long tt = wall_clock64() - tb0; const unsigned long long old = atomicMax(&g_wstat[8], tt); if (tt > old) { g_wstat[9] = w_times.fast; g_wstat[10] = w_times.general; g_wstat[11] = w_times.skip; g_wstat[12] = (unsigned long long)n0 << 32 | (unsigned)n1; }
			atomicAdd(&g_wstat[13], tt); atomicAdd(&g_wstat[14], w_times.fast); atomicAdd(&g_wstat[15], w_times.general); atomicAdd(&g_wstat[16], w_times.skip); }
#endif
		__syncthreads();
		{ // out: the lists as they are now
			const int m0 = new_n[0], m1 = new_n[1];
			uint32_t *t0 = (uint32_t *)(f.pregs + o0), *t1 = (uint32_t *)(f.pregs + o1);
			const uint32_t *u0 = (const uint32_t *)lds_regs, *u1 = (const uint32_t *)(lds_regs + c0);
			for (int k = lane; k < m0 * (int)(sizeof(Reg) / 4); k += 64) t0[k] = u0[k];
			for (int k = lane; k < m1 * (int)(sizeof(Reg) / 4); k += 64) t1[k] = u1[k];
		}
		__syncthreads();
	}
}
template <> void HipRT::run_rescue_heavy<KRescueStep>(const char *nm, int n, const int32_t *list, const KRescueStep &f)
{
	if (n <= 0) return;
	static const int wave = getenv("ARX_RESCUE_WAVE") ? atoi(getenv("ARX_RESCUE_WAVE")) : 1;
	// 1: three launches by LDS footprint (170 / 340 / 680 records: 5 / 3 / 2 workgroups per CU).  Measured in round 3: slower (36 -> 43 ms per step
	// alone on the repeat-rich workload, 14 -> 18 on the default one) -- the launches of one stream run one after the other and each ends on
	// its own longest pair; what bounds this kernel is the serial depth of its longest pairs, not the workgroups a CU holds.  Default: one launch.
	static const int split = getenv("ARX_RESCUE_LDS_CLASSES") ? atoi(getenv("ARX_RESCUE_LDS_CLASSES")) : 0;
	if (!rescue_heavy_attr_set) { ARX_HIP_CHECK(hipFuncSetAttribute((const void *)k_rescue_heavy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(RESCUE_LDS_REGS * sizeof(Reg)))); rescue_heavy_attr_set = true; }
	int32_t *cur = alloc<int32_t>(4);
	memset0(cur, 16);
	on_aux([&]() { // beside the thread-per-pair launch of the same round (the caller joins before it reads the round's task count)
		Scope sc(*this, nm, n);
		// footprints: 170 / 340 / 680 records = 15 / 30 / 60 KB of dynamic LDS beside the 17 KB of sort scratch: 5 / 3 / 2 workgroups per CU
		const int caps[4] = {0, split ? 170 : 0, split ? 340 : 0, RESCUE_LDS_REGS};
		for (int c = 2; c >= 0; --c) { // the longest first: their tail is what the launch ends on
			if (caps[c + 1] <= caps[c]) continue;
			const int per_cu = c == 0 ? 5 : c == 1 ? 3 : 2, blocks = n < n_cu * per_cu ? n : n_cu * per_cu;
			hipLaunchKernelGGL(k_rescue_heavy, dim3(blocks), dim3(64), (size_t)caps[c + 1] * sizeof(Reg), stream, f, list, n, wave, caps[c], caps[c + 1], cur + c);
		}
		ARX_HIP_CHECK(hipGetLastError());
	});
	wsort_report(stream, nm);
#ifdef ARX_WAVE_STATS
	{ unsigned long long h[24], z[24] = {0}; hipStreamSynchronize(stream); hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wstat), sizeof h); hipMemcpyToSymbol(HIP_SYMBOL(g_wstat), z, sizeof z);
	  fprintf(stderr, "wstat inserts %llu fast %llu gone %llu tie %llu unclean %llu noinsert-general %llu long %llu | worst block: total %llu fast %llu general %llu skip %llu (100 MHz ticks) n0 %llu n1 %llu | sums: total %llu fast %llu general %llu skip %llu\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[8], h[9], h[10], h[11], h[12] >> 32, h[12] & 0xffffffff, h[13], h[14], h[15], h[16]); }
#endif
}

// One read with many seed occurrences per 64-lane workgroup: the lanes copy the read's occurrences into LDS, the chaining state (lists,
// chains, B-tree nodes, the filter's rank arrays) lives there too, the wavefront chains and filters together (dev_chain_wave.h; wave = 0:
// lane 0 alone runs chain_and_filter(), kept for A/B runs); the kept chains and their seeds go to HBM as from KChain.
struct ChainLds { // carved out of the dynamic LDS block for a read with n occurrences
	Seed *occ; int32_t *rid, *next; Chain *ctmp; BtNode *nodes; int32_t *iscr, *xch; int cap_nodes;
	static __host__ __device__ size_t bytes(int n) { return (size_t)n * (sizeof(Seed) + 4 + 4 + sizeof(Chain) + 28) + ((size_t)n / 3 + 4) * sizeof(BtNode) + 64; }
	__device__ void carve(unsigned char *p, int n)
	{
		cap_nodes = n / 3 + 4;
		occ = (Seed *)p; p += (size_t)n * sizeof(Seed);
		ctmp = (Chain *)p; p += (size_t)n * sizeof(Chain);
		nodes = (BtNode *)p; p += (size_t)cap_nodes * sizeof(BtNode);
		xch = (int32_t *)p; p += 16;
		rid = (int32_t *)p; p += (size_t)n * 4;
		next = (int32_t *)p; p += (size_t)n * 4;
		iscr = (int32_t *)p;
	}
};
constexpr int CHAIN_LDS_SMALL = 256; // two launches: reads with up to 256 occurrences (40 KB of LDS, four workgroups per CU), and the rest (one per CU)
static __global__ void __launch_bounds__(64) k_chain_heavy(KChain f, int n_lo, int n_hi, int32_t *cursor, int wave)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_chain[];
	__shared__ int next_h;
	const int lane = threadIdx.x, n_heavy = *f.n_heavy;
	for (;;) {
		if (lane == 0) next_h = atomicAdd(cursor, 1);
		__syncthreads();
		const int h = next_h;
		__syncthreads();
		if (h >= n_heavy) break; // every workgroup reaches this: the cursor only grows
		const int r = f.heavy_list[h];
		const int g0 = f.occ_off[r], n = f.occ_off[r + 1] - g0;
		if (n < n_lo || n > n_hi) continue; // the other launch's read
		ChainLds L; L.carve(lds_chain, n);
		{ // in: the occurrences (16 bytes each) and their contigs
			const uint32_t *src = (const uint32_t *)(f.occ_seed + g0);
			uint32_t *dst = (uint32_t *)L.occ;
			for (int k = lane; k < n * (int)(sizeof(Seed) / 4); k += 64) dst[k] = src[k];
			for (int k = lane; k < n; k += 64) L.rid[k] = f.occ_rid[g0 + k];
		}
		__syncthreads();
		int m = 0;
		if (wave) m = w_chain_and_filter(f.ix, f.lens[r], f.intv + (size_t)r * CAP_INTV, f.n_intv[r], L.occ, L.rid, n, L.next, L.ctmp, L.nodes, L.cap_nodes, L.iscr,
		                                 f.cout + g0, f.sout + g0, g0, L.xch);
		else if (lane == 0) m = chain_and_filter(f.ix, f.lens[r], f.intv + (size_t)r * CAP_INTV, f.n_intv[r], L.occ, L.rid, n, L.next, L.ctmp, L.nodes, L.cap_nodes, L.iscr,
		                                         f.cout + g0, f.sout + g0, g0);
		if (lane == 0) {
			if (m < 0) { raise_err(f.err, ERR_POOL_OVERFLOW); m = 0; }
			f.n_chain[r] = m;
#ifdef ARX_CHAIN_STATS
			if (m > 0) {
				const unsigned long long tt = lds_cstat[5] - lds_cstat[0];
				for (int k = 0; k < 5; ++k) atomicAdd(&g_cstat[k], lds_cstat[k + 1] - lds_cstat[k]);
				atomicAdd(&g_cstat[5], 1ull);
				if (atomicMax(&g_cstat[8], tt) < tt) { for (int k = 0; k < 5; ++k) g_cstat[9 + k] = lds_cstat[k + 1] - lds_cstat[k]; g_cstat[14] = n; g_cstat[15] = m; }
			}
#endif
		}
		__syncthreads();
	}
}
template <> void HipRT::run_chain_heavy<KChain>(const char *nm, int n_reads, const KChain &f)
{
	Scope sc(*this, nm, n_reads);
	static const int wave = getenv("ARX_CHAIN_WAVE") ? atoi(getenv("ARX_CHAIN_WAVE")) : 1;
	static const size_t lds_s = ChainLds::bytes(CHAIN_LDS_SMALL), lds_l = ChainLds::bytes(CHAIN_LDS_OCC);
	// the opt-in applies to the device that is current when it is made: once per runtime (= per device context), not once per process
	if (!chain_heavy_attr_set) { ARX_HIP_CHECK(hipFuncSetAttribute((const void *)k_chain_heavy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_l)); chain_heavy_attr_set = true; }
	// f.n_heavy[0]: the list's length (stays on the device), [1] and [2]: the two launches' cursors into it
	static const int l_div = getenv("ARX_CHAIN_L_DIV") ? atoi(getenv("ARX_CHAIN_L_DIV")) : 1; // the long ones' launch holds 128 KB of LDS per workgroup: on n_cu / l_div CUs
	on_aux([&]() { hipLaunchKernelGGL(k_chain_heavy, dim3(n_cu / (l_div > 0 ? l_div : 1)), dim3(64), lds_l, stream, f, CHAIN_LDS_SMALL + 1, CHAIN_LDS_OCC, f.n_heavy + 2, wave); }); // the few long ones (beside the rest with ARX_AUX_STREAM=1)
	hipLaunchKernelGGL(k_chain_heavy, dim3(n_cu * 4), dim3(64), lds_s, stream, f, 0, CHAIN_LDS_SMALL, f.n_heavy + 1, wave); // (a third launch for reads of up to 128 occurrences at half the LDS changed nothing: round 3)
	ARX_HIP_CHECK(hipGetLastError());
	aux_join();
#ifdef ARX_CHAIN_STATS
	{ unsigned long long h[24], z[24] = {0}; hipStreamSynchronize(stream); hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cstat), sizeof h); hipMemcpyToSymbol(HIP_SYMBOL(g_cstat), z, sizeof z);
	  fprintf(stderr, "cstat reads %llu | sums (100 MHz ticks): chaining %llu traverse+weights %llu sort %llu filter %llu output %llu | worst read: total %llu = %llu %llu %llu %llu %llu, n_occ %llu kept %llu\n",
	          h[5], h[0], h[1], h[2], h[3], h[4], h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15]); }
#endif
}

// One read with a long region list per 64-lane workgroup: the list goes to LDS, the wavefront runs mem_sort_dedup_patch on it
// (w_sort_dedup) unless a pair of regions could get as far as mem_patch_reg's alignment -- then lane 0 runs the one-thread pass on the
// list in HBM, which nothing has touched yet.
static __global__ void __launch_bounds__(64) k_dedup_heavy(KDedup f, int32_t *eh_pool)
{
	__shared__ Reg lds_list[DEDUP_LDS_REGS];
	__shared__ WaveScratch ws;
	__shared__ int next_h;
	const int lane = threadIdx.x, n_heavy = *f.n_heavy;
	for (;;) {
		if (lane == 0) next_h = atomicAdd(f.n_heavy + 1, 1);
		__syncthreads();
		const int h = next_h;
		__syncthreads();
		if (h >= n_heavy) break;
		const int r = f.heavy_list[h], g0 = f.occ_off[r], n = f.n_ext[r];
		{
			const uint32_t *src = (const uint32_t *)(f.regs + g0);
			uint32_t *dst = (uint32_t *)lds_list;
			for (int k = lane; k < n * (int)(sizeof(Reg) / 4); k += 64) dst[k] = src[k];
		}
		__syncthreads();
		const int m = w_sort_dedup(n, lds_list, f.tmp + g0, ws, f.ix.l_pac);
		if (m == -2) {
			if (lane == 0) f.one_thread(r, eh_pool + (size_t)blockIdx.x * f.eh_words);
		} else {
			for (int i = lane; i < m; i += 64) { Reg x = lds_list[i]; if (x.rid >= 0 && f.ix.ann_alt[x.rid]) x.is_alt = 1; f.regs[g0 + i] = x; }
			if (lane == 0) { f.clean[r] = n >= 2 ? 2 : 1; f.n_core[r] = m; }
		}
		__syncthreads();
	}
}
template <> void HipRT::run_dedup_heavy<KDedup>(const char *nm, int n_reads, const KDedup &f)
{
	Scope sc(*this, nm, n_reads);
	const int blocks = n_cu * 4;
	int32_t *eh_pool = alloc<int32_t>((size_t)blocks * f.eh_words + 16);
	hipLaunchKernelGGL(k_dedup_heavy, dim3(blocks), dim3(64), 0, stream, f, eh_pool);
	ARX_HIP_CHECK(hipGetLastError());
	wsort_report(stream, nm);
}

template <class F> struct ColdUsesSlots { static const bool value = true; };
template <> struct ColdUsesSlots<KRescueStep> { static const bool value = false; }; // no per-slot scratch: may take one item per lane
template <class F> void HipRT::launch_cold(const char *nm, int n, const F &f) { launch_cold_impl(nm, n, f, !ColdUsesSlots<F>::value); }
void HipRT::merge_sort_fail_cold(uint32_t *err) { hipLaunchKernelGGL(k_merge_sort_fail, dim3(1), dim3(1), 0, stream, err); } // this unit's copy of the flag
template void HipRT::launch_cold<KDedup>(const char *, int, const KDedup &);
template void HipRT::launch_cold<KRescueStep>(const char *, int, const KRescueStep &);

// ---- arx_selftest_wave_sort (include/arachne_amd.h): one random index array per workgroup, w_introsort against ks_introsort
struct SelfKeyLt { const int *k; __device__ bool operator()(int a, int b) const { return k[a] < k[b]; } };
static __global__ void __launch_bounds__(64) k_selftest_wsort(int n_cases, uint64_t seed, unsigned long long *n_bad)
{
	constexpr int MAXN = CHAIN_WAVE_MAX;
	__shared__ int key[MAXN], idx[MAXN], chk[MAXN], lpos[MAXN], rpos[MAXN];
	const int lane = threadIdx.x;
	for (int c = blockIdx.x; c < n_cases; c += gridDim.x) {
		uint64_t x = seed * 0x9E3779B97F4A7C15ull + (uint64_t)c * 0xD1B54A32D192ED03ull + 1;
		auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
		const int n = 2 + (int)(next() % (uint64_t)(c % 4 == 0 ? MAXN - 1 : c % 4 == 1 ? 255 : c % 4 == 2 ? 40 : 15));
		const int range = 1 + (int)(next() % (uint64_t)(c % 3 == 0 ? 4 : c % 3 == 1 ? n : 8 * n)); // few distinct keys ... hardly any ties
		const int shape = (int)(next() % 4); // random, ascending, descending, ascending with a few swaps (the lists the passes sort are nearly sorted)
		for (int i = lane; i < n; i += 64) {
			uint64_t y = x + (uint64_t)i * 0x2545F4914F6CDD1Dull; y ^= y >> 29; y *= 0xBF58476D1CE4E5B9ull; y ^= y >> 32;
			int k = (int)(y % (uint64_t)range);
			if (shape == 1) k = (int)((int64_t)i * range / n);
			else if (shape == 2) k = (int)((int64_t)(n - 1 - i) * range / n);
			else if (shape == 3) k = (y >> 40) % 8 == 0 ? k : (int)((int64_t)i * range / n);
			key[i] = k; idx[i] = i; chk[i] = i;
		}
		__syncthreads();
		SelfKeyLt lt; lt.k = key;
		if (lane == 0) ks_introsort(n, chk, lt);
		__syncthreads();
		if (n <= 256) w_introsort<256>(n, idx, lt, lpos, rpos);
		else w_introsort<MAXN>(n, idx, lt, lpos, rpos);
		bool bad = false;
		for (int i = lane; i < n; i += 64) bad = bad || idx[i] != chk[i];
		if (__ballot(bad) && lane == 0) atomicAdd(n_bad, 1ull);
		__syncthreads();
	}
}
} // namespace arx

extern "C" int arx_selftest_wave_sort(int32_t device, int32_t n_cases, int64_t seed, int64_t *n_bad)
{
	if (n_cases < 0 || !n_bad) return ARX_E_ARG;
	unsigned long long *d = nullptr, h = 0;
	if (hipSetDevice(device) != hipSuccess || hipMalloc(&d, 8) != hipSuccess) return ARX_E_DEVICE;
	int rc = ARX_OK;
	if (hipMemset(d, 0, 8) != hipSuccess) rc = ARX_E_DEVICE;
	if (rc == ARX_OK && n_cases > 0) {
		hipLaunchKernelGGL(arx::k_selftest_wsort, dim3(n_cases < 4096 ? n_cases : 4096), dim3(64), 0, 0, n_cases, (uint64_t)seed, d);
		if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = ARX_E_DEVICE;
	}
	if (rc == ARX_OK && hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = ARX_E_DEVICE;
	(void)hipFree(d);
	*n_bad = (int64_t)h;
	return rc;
}
