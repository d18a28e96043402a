// hip_rt.h -- the HIP runtime policy of Pipeline<RT>: device memory, kernel launches on one stream, device scan,
// per-kernel timing with HIP events.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <map>
#include <string>
#include <vector>
#include <stdexcept>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include "hip_sw_coop.h"
#include "hip_block.h"
#include "hip_nw_coop.h"
#include "hip_fm_coop.h"
#include "index_build.h"

namespace arx {

#define ARX_HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

// generic grid-stride launchers; slot = global thread index (always < max_slots) selects per-thread scratch
#ifndef ARX_ITEMS_WPE
#define ARX_ITEMS_WPE 1 // (experiments) wavefronts per SIMD the thread-per-item kernels are compiled for
#endif
// per-functor register budget (wavefronts per SIMD the kernel is compiled for); specialised where a measurement says so
template <class F> struct ItemsWpe { static constexpr int v = ARX_ITEMS_WPE; };
template <class F> __global__ void __launch_bounds__(64, ItemsWpe<F>::v) k_items(F f, int n)
{
	const int slot = blockIdx.x * blockDim.x + threadIdx.x;
	const long long step = (long long)gridDim.x * blockDim.x; // i + step must not wrap: launches of more than 2^30 items exist (arx_open: ARX_SA_DENSE=4 at GRCh38 size)
	for (long long i = slot; i < n; i += step) f((int)i, slot);
}
// DP kernels: each thread owns words [threadIdx.x + j*blockDim.x] of the block's LDS, i.e. a [column][lane] layout
template <class F> __global__ void __launch_bounds__(64) k_rows(F f, int n)
{
	extern __shared__ uint32_t lds_rows[];
	const int slot = blockIdx.x * blockDim.x + threadIdx.x;
	const long long step = (long long)gridDim.x * blockDim.x;
	for (long long i = slot; i < n; i += step) f((int)i, slot, lds_rows + threadIdx.x, (int)blockDim.x);
}

struct CastI64 { __host__ __device__ int64_t operator()(const int32_t &x) const { return (int64_t)x; } };

struct KernelTimer { double ms = 0; int64_t calls = 0, items = 0; };

std::string product_bwt_sa(const uint8_t *pac, size_t pac_bytes, int64_t l_pac, const uint64_t cnt_fwd[4], const std::string &prefix); // arx_index.hip

struct HipRT {
	static const char *name() { return "hip:gfx950"; }
	static BwtSaFn bwt_sa_fn() { return product_bwt_sa; } // arx_index_build: the suffix sort runs in HBM
	hipStream_t stream = 0;
	// side stream for launches that are one wavefront's tail (the heavy-item kernels): they run beside the launches that follow on the main
	// stream until aux_join()
	hipStream_t aux = 0; hipEvent_t ev_fork = 0, ev_join = 0; bool aux_pending = false;
	// Off by default: beside each other the launches shorten one batch alone (62.2 -> 59.5 ms) but cost 5-6 % of the throughput with three
	// batches in flight (7.1 against 7.6 M pairs/s, same box): the other batches' kernels already fill the chip while a tail runs, and the
	// cross-stream waits add bubbles.  ARX_AUX_STREAM=1 turns it on (latency-bound use: one batch at a time).
	bool aux_ok = getenv("ARX_AUX_STREAM") && atoi(getenv("ARX_AUX_STREAM")) != 0;
	template <class L> void on_aux(L f)
	{
		if (!aux_ok) { f(); return; }
		if (!aux) {
			ARX_HIP_CHECK(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
			ARX_HIP_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming)); ARX_HIP_CHECK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
		}
		ARX_HIP_CHECK(hipEventRecord(ev_fork, stream)); ARX_HIP_CHECK(hipStreamWaitEvent(aux, ev_fork, 0));
		hipStream_t main_stream = stream;
		stream = aux; f(); stream = main_stream;
		ARX_HIP_CHECK(hipEventRecord(ev_join, aux));
		aux_pending = true;
	}
	void aux_join() { if (aux_pending) { ARX_HIP_CHECK(hipStreamWaitEvent(stream, ev_join, 0)); aux_pending = false; } }
	int n_cu = 256;
	bool timing = false;
	std::map<std::string, KernelTimer> tm;
	void *scan_tmp = 0; size_t scan_tmp_bytes = 0; int64_t *d_total = 0; void *pinned = 0;

	int dev = 0;
	void bind() { (void)hipSetDevice(dev); }           // the current device is per host thread
	void set_timing(bool on) { timing = on; }
	std::string init(int device)
	{
		int n = 0;
		dev = device;
		if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return "no HIP device visible: libarachne_amd.so needs an MI355X (there is no CPU fallback)";
		if (device < 0 || device >= n) return "device index out of range";
		if (hipSetDevice(device) != hipSuccess) return "hipSetDevice failed";
		hipDeviceProp_t p;
		if (hipGetDeviceProperties(&p, device) != hipSuccess) return "hipGetDeviceProperties failed";
		n_cu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
		if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return "hipStreamCreate failed";
		(void)hipMalloc(&d_total, 8);
		(void)hipHostMalloc(&pinned, 64, hipHostMallocDefault);
		return "";
	}
	~HipRT()
	{
		if (sw_filter_stats && sw_tasks_seen) fprintf(stderr, "[arx] rescue alignments queued %lld, run after the pre-filter %lld\n", (long long)sw_tasks_seen, (long long)sw_tasks_run);
		for (auto &sl : slabs) (void)hipFree(sl.p);
		if (scan_tmp) hipFree(scan_tmp);
		if (d_total) hipFree(d_total);
		if (pinned) (void)hipHostFree(pinned);
		if (stage_buf) (void)hipHostFree(stage_buf);
		for (auto &p : pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
		for (auto e : free_events) (void)hipEventDestroy(e);
		if (aux) { hipStreamDestroy(aux); hipEventDestroy(ev_fork); hipEventDestroy(ev_join); }
		if (stream) hipStreamDestroy(stream);
	}
	// Work memory of a batch comes from a per-runtime arena: slabs obtained once with hipMalloc, bump-allocated, reset as
	// a whole when the batch is re-run.  Steady state therefore has no hipMalloc/hipFree at all -- both synchronise the
	// device and would serialise the batches that run on other streams.
	struct Slab { char *p; size_t cap, used; };
	std::vector<Slab> slabs;
	template <class T> T *alloc(size_t n)
	{
		size_t bytes = ((n ? n : 1) * sizeof(T) + 255) & ~(size_t)255;
		for (auto &sl : slabs) if (sl.cap - sl.used >= bytes) { char *r = sl.p + sl.used; sl.used += bytes; return (T *)r; }
		Slab sl; sl.cap = bytes > ((size_t)1 << 30) ? bytes : ((size_t)1 << 30); sl.used = bytes;
		ARX_HIP_CHECK(hipMalloc((void **)&sl.p, sl.cap));
		slabs.push_back(sl);
		return (T *)sl.p;
	}
	void free(void *) {}                       // arena memory is released by arena_reset()
	void arena_reset() { for (auto &sl : slabs) sl.used = 0; }
	std::vector<size_t> arena_mark() const { std::vector<size_t> m; for (auto &sl : slabs) m.push_back(sl.used); return m; }
	void arena_rewind(const std::vector<size_t> &m) { for (size_t i = 0; i < slabs.size(); ++i) slabs[i].used = i < m.size() ? m[i] : 0; }
	template <class T> T *palloc(size_t n) { void *p = 0; ARX_HIP_CHECK(hipMalloc(&p, (n ? n : 1) * sizeof(T))); return (T *)p; } // persistent
	void pfree(void *p) { if (p) (void)hipFree(p); }
	void h2d(void *d, const void *s, size_t bytes) { if (bytes) { ARX_HIP_CHECK(hipMemcpyAsync(d, s, bytes, hipMemcpyHostToDevice, stream)); ARX_HIP_CHECK(hipStreamSynchronize(stream)); } }
	void d2h(void *d, const void *s, size_t bytes)
	{
		if (!bytes) return;
		if (bytes <= 64 && pinned) { // small read-backs (round counters, error word, scan totals) go through pinned memory
			ARX_HIP_CHECK(hipMemcpyAsync(pinned, s, bytes, hipMemcpyDeviceToHost, stream));
			ARX_HIP_CHECK(hipStreamSynchronize(stream));
			memcpy(d, pinned, bytes);
			return;
		}
		ARX_HIP_CHECK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, stream));
		ARX_HIP_CHECK(hipStreamSynchronize(stream));
	}
	void d2h_async(void *d, const void *s, size_t bytes) { if (bytes) ARX_HIP_CHECK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, stream)); } // sync() before the data is read
	// staging for uploads: pinned host memory owned by the runtime, grown on demand.  stage() waits for the copies of the previous
	// use (the stream has long passed them when a batch is reset after its results were fetched); h2d_staged() only enqueues.
	void *stage_buf = 0; size_t stage_cap = 0;
	void *stage(size_t bytes)
	{
		ARX_HIP_CHECK(hipStreamSynchronize(stream));
		if (bytes > stage_cap) {
			if (stage_buf) (void)hipHostFree(stage_buf);
			stage_buf = 0; stage_cap = bytes + bytes / 8 + 4096;
			ARX_HIP_CHECK(hipHostMalloc(&stage_buf, stage_cap, hipHostMallocDefault));
		}
		return stage_buf;
	}
	void h2d_staged(void *d, const void *staged, size_t bytes) { if (bytes) ARX_HIP_CHECK(hipMemcpyAsync(d, staged, bytes, hipMemcpyHostToDevice, stream)); }
	// Host memory the caller has page-locked (arx_host_register): copies to and from it are DMA at PCIe speed without a staging copy --
	// hipMemcpyAsync finds that out by itself for the results (d2h); an upload from it skips the runtime's own pinned staging buffer.
	static int host_register(void *p, size_t bytes) { const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable); if (e != hipSuccess) (void)hipGetLastError(); return e == hipSuccess ? 0 : -1; }
	static int host_unregister(void *p) { const hipError_t e = hipHostUnregister(p); if (e != hipSuccess) (void)hipGetLastError(); return e == hipSuccess ? 0 : -1; }
	static bool host_pinned(const void *p)
	{
		hipPointerAttribute_t a;
		if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
		return a.type == hipMemoryTypeHost;
	}
	void h2d_pinned(void *d, const void *s, size_t bytes) // from page-locked caller memory; returns when the copy is done (the caller may reuse the array)
	{
		if (!bytes) return;
		ARX_HIP_CHECK(hipMemcpyAsync(d, s, bytes, hipMemcpyHostToDevice, stream));
		ARX_HIP_CHECK(hipStreamSynchronize(stream));
	}
	uint64_t free_bytes() const { size_t f = 0, t = 0; return hipMemGetInfo(&f, &t) == hipSuccess ? (uint64_t)f : 0; }
	void d2d(void *d, const void *s, size_t bytes) { if (bytes) ARX_HIP_CHECK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, stream)); }
	void memset0(void *d, size_t bytes) { ARX_HIP_CHECK(hipMemsetAsync(d, 0, bytes, stream)); }
	void memset_bytes(void *d, int v, size_t bytes) { ARX_HIP_CHECK(hipMemsetAsync(d, v, bytes, stream)); }
	void sync() { ARX_HIP_CHECK(hipStreamSynchronize(stream)); }

	// 8 resident 64-thread blocks per CU give every SIMD two waves of these latency-bound kernels
	int bpc = getenv("ARX_BPC") ? atoi(getenv("ARX_BPC")) : 16;           // resident 64-lane blocks per CU of the thread-per-item kernels (sizes their per-slot scratch)
	int coop_bpc = getenv("ARX_COOP_BPC") ? atoi(getenv("ARX_COOP_BPC")) : 64; // grid cap of the 16-lane DP kernels (no per-slot scratch; grid-stride)
	int ext_merge_below = getenv("ARX_EXT_MERGE") ? atoi(getenv("ARX_EXT_MERGE")) : 30000; // rounds with fewer extensions run all length classes in one launch
	int strat_bpc = getenv("ARX_STRAT_BPC") ? atoi(getenv("ARX_STRAT_BPC")) : 4 * ARX_SEED_WPE; // resident blocks per CU of the third seeding pass
	int max_blocks() const { return n_cu * bpc; }
	int coop_blocks(int n) const { int b = (n + 3) / 4, cap = n_cu * coop_bpc; return b < cap ? b : cap; }
	int max_slots() const { return max_blocks() * 64; }
	int max_slots_small() const { return n_cu * 64; }
	// seeding kernels: 4 * ARX_SEED_WPE resident blocks per CU (their register budget is compiled for that many waves per SIMD)
	int seed_bpc = getenv("ARX_SEED_BPC") ? atoi(getenv("ARX_SEED_BPC")) : 4 * ARX_SEED_WPE;
	int max_seed_slots() const { return n_cu * seed_bpc * 64; }
	// the row-parallel backward kernel needs 94 VGPRs: five wavefronts per SIMD fit, not only the four its launch bound asks for, so its grid is
	// 20 workgroups per CU (4.81 -> 4.62 ms alone; 24 and more lose again, and a build that forces six per SIMD spills: 7.2 ms)
	int seed_bwd_mid = getenv("ARX_SEED_BWD_MID") ? atoi(getenv("ARX_SEED_BWD_MID")) : 21; // longest list of the 21-lane bin of the backward sweeps (16: none)
	int seed_bwd_e_bpc = getenv("ARX_SEED_BWD_E_BPC") ? atoi(getenv("ARX_SEED_BWD_E_BPC")) : 32; // its resident workgroups per CU (62 VGPRs: eight wavefronts per SIMD fit)
	int seed_bwd_e_chunk = getenv("ARX_SEED_BWD_E_CHUNK") ? atoi(getenv("ARX_SEED_BWD_E_CHUNK")) : 256; // list entries a wavefront reserves per atomic (entry-parallel sweeps)
	bool seed_fit32 = !(getenv("ARX_SEED_FIT32") && atoi(getenv("ARX_SEED_FIT32")) == 0); // 0: the general (40-bit) arithmetic in the backward sweeps whatever the index (A/B)
	bool text_bwd = !(getenv("ARX_TEXT_BWD") && atoi(getenv("ARX_TEXT_BWD")) == 0);
	int seed_bwd_bpc = getenv("ARX_SEED_BWD_BPC") ? atoi(getenv("ARX_SEED_BWD_BPC")) : 20;
	int seed_row = SEED_ROW;                                   // LDS bytes per lane for its read
	void set_seed_read_len(int max_len) { seed_row = seed_row_bytes(max_len); }
	// the batch's reads as nibble rows of seed_row bytes (hip_fm_coop.h: k_pack_reads): what the seeding kernels stage into LDS
	const uint32_t *seed_qn = nullptr;
	void seed_prepare(const uint8_t *bases, const int32_t *base_off, const int32_t *lens, int n_reads)
	{
		if (sw_simple || n_reads <= 0) return;
		const int rw = seed_row >> 2;
		uint32_t *q = alloc<uint32_t>((size_t)n_reads * rw + 8);
		Scope sc(*this, "seed_pack", n_reads);
		const long long total = (long long)n_reads * rw;
		hipLaunchKernelGGL(k_pack_reads, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, bases, base_off, lens, n_reads, rw, q);
		ARX_HIP_CHECK(hipGetLastError());
		seed_qn = q;
	}

	// Kernel timing: a pair of HIP events around each launch on the launch stream, recorded without blocking and
	// resolved (hipEventElapsedTime) the next time the stream is known to be idle.
	struct Pending { hipEvent_t a, b; const char *nm; int64_t items; };
	std::vector<Pending> pending;
	std::vector<hipEvent_t> free_events;
	hipEvent_t get_event()
	{
		if (!free_events.empty()) { hipEvent_t e = free_events.back(); free_events.pop_back(); return e; }
		hipEvent_t e; ARX_HIP_CHECK(hipEventCreate(&e)); return e;
	}
	bool trace_launches = getenv("ARX_TRACE_LAUNCHES") != nullptr; // diagnostics: name every launch on stderr and wait for it (a fault then names its kernel)
	struct Scope {
		HipRT &rt; Pending p; bool on; const char *tn;
		Scope(HipRT &r, const char *n, int64_t it) : rt(r), on(r.timing), tn(r.trace_launches ? n : nullptr)
		{
			if (tn) { fprintf(stderr, "[arx launch] %s (%lld items) ...\n", tn, (long long)it); fflush(stderr); }
			if (!on) return;
			p.a = rt.get_event(); p.b = rt.get_event(); p.nm = n; p.items = it;
			(void)hipEventRecord(p.a, rt.stream);
		}
		~Scope()
		{
			if (on) { (void)hipEventRecord(p.b, rt.stream); rt.pending.push_back(p); }
			if (tn) { (void)hipStreamSynchronize(rt.stream); fprintf(stderr, "[arx launch] %s done\n", tn); fflush(stderr); }
		}
	};
	void resolve_timers()
	{
		if (pending.empty()) return;
		(void)hipStreamSynchronize(stream);
		static FILE *launch_log = getenv("ARX_LAUNCH_LOG") ? fopen(getenv("ARX_LAUNCH_LOG"), "a") : nullptr; // diagnostics: one line per launch
		for (auto &p : pending) {
			float ms = 0;
			(void)hipEventElapsedTime(&ms, p.a, p.b);
			if (launch_log) fprintf(launch_log, "%s\t%lld\t%.4f\n", p.nm, (long long)p.items, ms);
			KernelTimer &t = tm[p.nm]; t.ms += ms; ++t.calls; t.items += p.items;
			free_events.push_back(p.a); free_events.push_back(p.b);
		}
		pending.clear();
		if (launch_log) fflush(launch_log);
	}
	std::map<std::string, KernelTimer> &timers() { resolve_timers(); return tm; }
	void timers_reset(bool enable) { resolve_timers(); tm.clear(); timing = enable; }

	// for functors that use no per-slot scratch: one item per lane, as many blocks as that takes -- the hardware hands blocks to CUs
	// as they free up, which balances kernels whose items differ a lot in cost better than a fixed grid-stride assignment
	bool wide_ok = !(getenv("ARX_WIDE") && atoi(getenv("ARX_WIDE")) == 0);
	template <class F> void launch_wide(const char *nm, int n, const F &f)
	{
		if (n <= 0) return;
		if (!wide_ok) { launch(nm, n, f); return; }
		Scope sc(*this, nm, n);
		hipLaunchKernelGGL(k_items<F>, dim3((n + 63) / 64), dim3(64), 0, stream, f, n);
		ARX_HIP_CHECK(hipGetLastError());
	}
	template <class F> void launch(const char *nm, int n, const F &f)
	{
		if (n <= 0) return;
		Scope sc(*this, nm, n);
		int blocks = (n + 63) / 64; if (blocks > max_blocks()) blocks = max_blocks();
		hipLaunchKernelGGL(k_items<F>, dim3(blocks), dim3(64), 0, stream, f, n);
		ARX_HIP_CHECK(hipGetLastError());
	}
	// "cold" kernels (list bookkeeping: dedup, rescue_step) are instantiated in arx_cold.hip, a translation unit of its own (-O3 like
	// the rest since round 2; arx_dev.h ks_introsort has the story of the -O1 build they needed before)
	template <class F> void launch_cold(const char *nm, int n, const F &f);
	// ks_introsort's budget flag (arx_dev.h) of both device translation units into a batch's error word; no host round trip of its own
	void merge_sort_fail(uint32_t *err) { hipLaunchKernelGGL(k_merge_sort_fail, dim3(1), dim3(1), 0, stream, err); merge_sort_fail_cold(err); }
	void merge_sort_fail_cold(uint32_t *err); // arx_cold.hip
	// rescue replay of the pairs with long lists, lists staged in LDS (arx_cold.hip)
	bool rescue_heavy_ok() const { return !(getenv("ARX_RESCUE_HEAVY") && atoi(getenv("ARX_RESCUE_HEAVY")) == 0); }
	template <class F> void run_rescue_heavy(const char *nm, int n, const int32_t *list, const F &f);
	// chaining of the reads with many seed occurrences, one wavefront per read on a working set in LDS (arx_cold.hip); f.heavy_list / f.n_heavy
	bool chain_heavy_ok() const { return !(getenv("ARX_CHAIN_HEAVY") && atoi(getenv("ARX_CHAIN_HEAVY")) == 0); }
	template <class F> void run_chain_heavy(const char *nm, int n_reads, const F &f);
	bool rescue_heavy_attr_set = false;
	bool chain_heavy_attr_set = false; // the 128 KB dynamic-LDS opt-in of k_chain_heavy was made on this runtime's device
	bool dedup_heavy_ok() const { return !(getenv("ARX_DEDUP_HEAVY") && atoi(getenv("ARX_DEDUP_HEAVY")) == 0); }
	template <class F> void run_dedup_heavy(const char *nm, int n_reads, const F &f); // likewise the region lists of such reads (f.eh_words ints of scratch per workgroup)
	template <class F> void launch_cold_impl(const char *nm, int n, const F &f, bool wide = false)
	{
		if (n <= 0) return;
		Scope sc(*this, nm, n);
		int blocks = (n + 63) / 64; if (!(wide && wide_ok) && blocks > max_blocks()) blocks = max_blocks();
		hipLaunchKernelGGL(k_items<F>, dim3(blocks), dim3(64), 0, stream, f, n);
		ARX_HIP_CHECK(hipGetLastError());
	}
	// one work item per BLOCK_LANES-lane workgroup (hip_block.h); f(item, HipBlock&)
	// small (host, may be null): small[i] = 1 sends item i to a SMALL_LANES-lane workgroup
	template <class F> void launch_block(const char *nm, int n, const F &f, const uint8_t *small = nullptr)
	{
		if (n <= 0) return;
		Scope sc(*this, nm, n);
		int n_small = 0;
		if (small) for (int i = 0; i < n; ++i) n_small += small[i] ? 1 : 0;
		if (n_small == 0) {
			int blocks = n < n_cu * 8 ? n : n_cu * 8;
			hipLaunchKernelGGL((k_block_items<F, BLOCK_LANES, SORT_LDS>), dim3(blocks), dim3(BLOCK_LANES), 0, stream, f, n, (const uint8_t *)nullptr, 0);
		} else {
			uint8_t *d = alloc<uint8_t>((size_t)n + 8);
			h2d(d, small, (size_t)n);
			int blocks = n_small < n_cu * 32 ? n_small : n_cu * 32;
			hipLaunchKernelGGL((k_block_items<F, SMALL_LANES, SMALL_SORT>), dim3(blocks), dim3(SMALL_LANES), 0, stream, f, n, (const uint8_t *)d, 1);
			if (n_small < n) {
				blocks = n - n_small < n_cu * 8 ? n - n_small : n_cu * 8;
				hipLaunchKernelGGL((k_block_items<F, BLOCK_LANES, SORT_LDS>), dim3(blocks), dim3(BLOCK_LANES), 0, stream, f, n, (const uint8_t *)d, 0);
			}
		}
		ARX_HIP_CHECK(hipGetLastError());
	}
	template <class F> void launch_small(const char *nm, int n, const F &f)
	{
		if (n <= 0) return;
		Scope sc(*this, nm, n);
		int blocks = (n + 63) / 64; if (blocks > n_cu) blocks = n_cu;
		hipLaunchKernelGGL(k_items<F>, dim3(blocks), dim3(64), 0, stream, f, n);
		ARX_HIP_CHECK(hipGetLastError());
	}
	// rescue SW: 16 lanes per alignment (hip_sw_coop.h); ARX_SW_SIMPLE=1 selects the one-thread-per-alignment kernel for A/B runs
	bool sw_simple = getenv("ARX_SW_SIMPLE") != nullptr;
	template <class F> void run_sw_u8(const char *nm, int n, const F &f, int max_len)
	{
		if (n <= 0) return;
		if (sw_simple) { launch_rows(nm, n, f, 16 * ((max_len + 15) / 16)); return; }
		const int blocks = coop_blocks(n);
		int32_t *order = nullptr, *n_order = nullptr;
		if (sw_filter) { // tasks that provably stay below min_seed_len never reach the DP (dev_sw.h: sw_prefilter_serial)
			order = alloc<int32_t>((size_t)n + 1); n_order = order + n;
			memset0(n_order, 4);
			Scope sc(*this, "sw_filter", n);
			hipLaunchKernelGGL(k_sw_filter_g16, dim3(blocks), dim3(64), 0, stream, f.ix, f.bases, f.base_off, f.lens, f.tasks, f.res, n, order, n_order);
			ARX_HIP_CHECK(hipGetLastError());
		}
		{
			Scope sc(*this, nm, n);
			if (max_len <= 160) hipLaunchKernelGGL(k_sw_u8_g16<10>, dim3(blocks), dim3(64), 0, stream, f.ix, f.bases, f.base_off, f.lens, f.tasks, f.res, n, order, n_order);
			else if (max_len * OPT_A < 250) hipLaunchKernelGGL(k_sw_u8_g16<16>, dim3(blocks), dim3(64), 0, stream, f.ix, f.bases, f.base_off, f.lens, f.tasks, f.res, n, order, n_order);
			else hipLaunchKernelGGL(k_sw_u8_g16<32>, dim3(blocks), dim3(64), 0, stream, f.ix, f.bases, f.base_off, f.lens, f.tasks, f.res, n, order, n_order); // mates of 250+ bases: ksw_i16's eight stripes of up to 32 cells; shorter mates of the same batch take the byte form inside
			ARX_HIP_CHECK(hipGetLastError());
		}
		if (sw_filter_stats) { int32_t k = 0; d2h(&k, n_order, 4); sw_tasks_seen += n; sw_tasks_run += k; }
	}
	// Off by default: on the benchmark workload 99.6 % of the rescue alignments are real hits in repeat copies (nothing to drop, the
	// filter's 1 ms per batch is lost); on workloads with chimeric or unpaired reads it drops 40 % of them (profiles/r01/README.md).
	int sw_filter = getenv("ARX_SW_FILTER") ? atoi(getenv("ARX_SW_FILTER")) : 0;
	int sw_filter_stats = getenv("ARX_SW_FILTER_STATS") ? atoi(getenv("ARX_SW_FILTER_STATS")) : 0; // diagnostics: one extra host round trip per launch
	int64_t sw_tasks_seen = 0, sw_tasks_run = 0;
	// seeding: persistent lanes, items handed out in chunks (hip_fm_coop.h); f is one of pipeline.h's KSeedFwd1 / KSeedFwd2 / KSeedBwd,
	// f.scratch holds max_slots() forward lists
	int seed_batch = getenv("ARX_SEED_BATCH") ? atoi(getenv("ARX_SEED_BATCH")) : 48; // lanes that queue up before the slow bookkeeping runs
	int seed_bwd_budget = getenv("ARX_SEED_BWD_BUDGET") ? atoi(getenv("ARX_SEED_BWD_BUDGET")) : 128; // extensions a lane spends on one backward sweep before handing it to a wavefront (0: never)
	int seed_chunk = getenv("ARX_SEED_CHUNK") ? atoi(getenv("ARX_SEED_CHUNK")) : 64; // items a wavefront reserves per atomic
	int seed_bwd_chunk = getenv("ARX_SEED_BWD_CHUNK") ? atoi(getenv("ARX_SEED_BWD_CHUNK")) : (getenv("ARX_SEED_CHUNK") ? atoi(getenv("ARX_SEED_CHUNK")) : 32); // backward sweeps vary most in length: smaller reservations even out the end of the launch (64: 10.3 ms, 32: 9.4, 16: 9.7, 8: 10.3 per batch)
	// Backward sweeps: 2 (default) = row-parallel, one task per 16/32/64-lane group with the row's entries in registers (k_seed_bwd_g<GL>,
	// tasks binned by list length): 21.5 -> 10 ms per 667 k-read batch at GRCh38 size.  1 = the pipelined one-lane-per-task kernel
	// k_seed_bwd2 (51-61 of 64 lanes extending instead of 25-32, but no faster: profiles/r02/README.md).  0 = round 1's k_seed_bwd.
	// All three are bit-identical.
	int seed_bwd2 = getenv("ARX_SEED_BWD2") ? atoi(getenv("ARX_SEED_BWD2")) : 2;
	int seed_grant = getenv("ARX_SEED_GRANT") ? atoi(getenv("ARX_SEED_GRANT")) : 4; // first forward pass: lanes parked for a pool slice that trigger the hand-out (5.36 ms with none, 5.17 at 16, 4.94 at 4, 5.08 at 1)
	int seed_bwd_batch = getenv("ARX_SEED_BWD_BATCH") ? atoi(getenv("ARX_SEED_BWD_BATCH")) : 0; // 0: seed_batch
	// diagnostics (ARX_SEED_STATS=1): lane utilisation of the persistent-lane seeding kernels, printed per launch
	unsigned long long *seed_dbg_buf = nullptr;
	unsigned long long *seed_dbg()
	{
		if (!getenv("ARX_SEED_STATS")) return nullptr;
		if (!seed_dbg_buf) ARX_HIP_CHECK(hipMalloc((void **)&seed_dbg_buf, 32));
		memset0(seed_dbg_buf, 32);
		return seed_dbg_buf;
	}
	void seed_dbg_report(const char *nm, int n)
	{
		if (!seed_dbg_buf || !getenv("ARX_SEED_STATS")) return;
		unsigned long long h[4];
		d2h(h, seed_dbg_buf, 32);
		fprintf(stderr, "[arx seed stats] %s: %d items, %llu waves, %.0f iterations/wave, %.1f lanes extending per iteration, %.0f slow-path entries/wave\n", nm, n, h[3],
		        h[3] ? (double)h[0] / h[3] : 0.0, h[0] ? (double)h[1] / h[0] : 0.0, h[3] ? (double)h[2] / h[3] : 0.0);
	}
	template <class K> void launch_seed_kernel(const char *nm, K kern, int n, const SeedKArgs &A, int32_t *counter, int bpc_, int chunk_ = 0, int batch_ = 0, int grant_ = 64)
	{
		if (chunk_ <= 0) chunk_ = seed_chunk;
		if (batch_ <= 0) batch_ = seed_batch;
		memset0(counter, 4);
		Scope sc(*this, nm, n);
		int blocks = (n + 63) / 64; if (blocks > n_cu * bpc_) blocks = n_cu * bpc_;
		hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 64 * (size_t)seed_row, stream, A, n, counter, (batch_ & 0xff) | grant_ << 8, chunk_);
		ARX_HIP_CHECK(hipGetLastError());
	}
	template <class F> void run_seed_fwd1(const char *nm, int n, const F &f, int32_t *counter)
	{
		if (n <= 0) return;
		if (sw_simple) { launch(nm, n, f); return; }
		SeedKArgs A{f.ix, f.bases, f.base_off, f.lens, f.P, f.scratch, f.list_cap, f.first1, 0, nullptr, nullptr, 0, seed_row, seed_qn, f.read0, seed_dbg()};
		launch_seed_kernel(nm, k_seed_fwd1, n, A, counter, seed_bpc, 0, 0, seed_grant); // (the re-seeding pass has one extension per item: no grant step of its own)
		seed_dbg_report(nm, n);
	}
	template <class F> void run_seed_fwd2(const char *nm, int n, const F &f, int32_t *counter)
	{
		if (n <= 0) return;
		if (sw_simple) { launch(nm, n, f); return; }
		SeedKArgs A{f.ix, f.bases, f.base_off, f.lens, f.P, f.scratch, f.list_cap, nullptr, f.t0, nullptr, nullptr, 0, seed_row, seed_qn, 0, seed_dbg()};
		launch_seed_kernel(nm, k_seed_fwd2, n, A, counter, seed_bpc);
		seed_dbg_report(nm, n);
	}
	template <class F> void run_seed_bwd(const char *nm, int n, const F &f, int32_t *counter)
	{
		if (n <= 0) return;
		if (sw_simple) { launch(nm, n, f); return; }
		// sweeps longer than seed_bwd_budget extensions are finished by whole wavefronts (k_seed_bwd_wave)
		int32_t *heavy = alloc<int32_t>((size_t)n + 2);
		memset0(heavy + n, 4);
		SeedKArgs A{f.ix, f.bases, f.base_off, f.lens, f.P, nullptr, 0, nullptr, f.t0, heavy, heavy + n, seed_bwd_budget, seed_row, seed_qn, 0, seed_dbg()};
		if (!text_bwd) A.ix.isa40 = nullptr; // ARX_TEXT_BWD=0: every sweep walked to its end (k_seed_bwd_g hands nothing to KSeedBwdTail)
		if (seed_bwd2 == 3 && seed_row <= 132) { // entry-parallel sweeps (k_seed_bwd_e): one lane per list entry
			int32_t *ecnt = alloc<int32_t>((size_t)n + 2), *eoff = alloc<int32_t>((size_t)n + 2);
			Scope sc(*this, nm, n);
			hipLaunchKernelGGL(k_bwd_e_count, dim3((n + 255) / 256), dim3(256), 0, stream, f.P.tasks, f.t0, n, ecnt);
			const int64_t total = exclusive_scan(ecnt, eoff, n); // (waits for the stream)
			BwdItem *items = alloc<BwdItem>((size_t)total + 4);
			hipLaunchKernelGGL(k_bwd_e_expand, dim3((n + 255) / 256), dim3(256), 0, stream, f.P.tasks, f.P.pool, f.t0, n, eoff, items);
			if (total > 0) {
				memset0(counter, 4);
				int64_t blocks = (total + 63) / 64; if (blocks > (int64_t)n_cu * seed_bwd_e_bpc) blocks = (int64_t)n_cu * seed_bwd_e_bpc;
				hipLaunchKernelGGL(k_seed_bwd_e, dim3((unsigned)blocks), dim3(64), 0, stream, A, items, (int)total, counter, seed_bwd_e_chunk);
				hipLaunchKernelGGL(k_bwd_e_final, dim3((n + 255) / 256), dim3(256), 0, stream, f.P.tasks, f.P.pool, f.t0, n);
			}
			ARX_HIP_CHECK(hipGetLastError());
			seed_dbg_report(nm, (int)total);
			return;
		}
		if (seed_bwd2 == 2 && seed_row <= 132) { // row-parallel sweeps (k_seed_bwd_g<GL>): one task per 16/32/64-lane group, lists in registers
			uint8_t *flag = alloc<uint8_t>((size_t)n + 8);
			int32_t *bins = alloc<int32_t>(4 * (size_t)n + 8), *cnt = alloc<int32_t>(8); // cnt[0..3]: bin sizes, cnt[4..7]: the bins' item counters
			memset0(flag, (size_t)n);
			memset0(cnt, 32);
			const int cap = n_cu * seed_bwd_bpc;
			auto blocks_for = [&](int per_wave) { int b = (n + per_wave - 1) / per_wave; return b > cap ? cap : (b < 1 ? 1 : b); };
			const size_t xch = 64 * 32;
			{
				Scope sc(*this, nm, n);
				hipLaunchKernelGGL(k_bin_tasks, dim3((n + 255) / 256), dim3(256), 0, stream, f.P.tasks, f.t0, n, bins, bins + n, bins + 2 * (size_t)n, bins + 3 * (size_t)n, cnt, seed_bwd_mid);
				const bool fit32 = seed_fit32 && (((f.ix.L2[1] - f.ix.L2[0]) | (f.ix.L2[2] - f.ix.L2[1]) | (f.ix.L2[3] - f.ix.L2[2]) | (f.ix.L2[4] - f.ix.L2[3])) >> 32) == 0;
				if (fit32) hipLaunchKernelGGL(k_seed_bwd_g<true>, dim3(blocks_for(4)), dim3(64), ((4 * (size_t)seed_row + 31) & ~(size_t)31) + xch, stream, A, bins, n, cnt, flag);
				else hipLaunchKernelGGL(k_seed_bwd_g<false>, dim3(blocks_for(4)), dim3(64), ((4 * (size_t)seed_row + 31) & ~(size_t)31) + xch, stream, A, bins, n, cnt, flag);
				ARX_HIP_CHECK(hipGetLastError());
			}
			if (getenv("ARX_SEED_HIST")) { // diagnostics: forward-list lengths of this launch's tasks
				std::vector<SeedTask> ht((size_t)n);
				d2h(ht.data(), f.P.tasks + f.t0, (size_t)n * sizeof(SeedTask));
				long long hist[40] = {0};
				for (auto &k : ht) ++hist[k.n < 39 ? k.n : 39];
				fprintf(stderr, "[arx seed hist] %d tasks, list lengths 0..39+:", n);
				for (int i = 0; i < 40; ++i) fprintf(stderr, " %lld", hist[i]);
				fprintf(stderr, "\n");
			}
			if (getenv("ARX_SEED_STATS")) { int32_t h[4]; d2h(h, cnt, 16); fprintf(stderr, "[arx seed stats] backward tasks by list length: <= 16: %d, <= %d: %d, <= 32: %d, longer: %d\n", h[0], seed_bwd_mid, h[1], h[2], h[3]); }
			seed_dbg_report(nm, n);
			{
				Scope sc(*this, "seed_bwd_wave", n);
				hipLaunchKernelGGL(k_collect_heavy, dim3((n + 255) / 256), dim3(256), 0, stream, flag, n, f.t0, heavy, heavy + n);
				hipLaunchKernelGGL(k_seed_bwd_wave, dim3(n_cu * 16), dim3(64), 0, stream, A);
				if (A.ix.isa40) { // the sweeps k_seed_bwd_g left at a row of one interval with one occurrence (text mode)
					KSeedBwdTail kt{f.ix, f.bases, f.base_off, f.P.pool, f.P.tasks, f.t0, flag};
					hipLaunchKernelGGL(k_items<KSeedBwdTail>, dim3((n + 63) / 64), dim3(64), 0, stream, kt, n);
				}
				ARX_HIP_CHECK(hipGetLastError());
			}
			return;
		}
		if (seed_bwd2) { // pipelined refills (k_seed_bwd2): one wait on memory per iteration
			uint8_t *flag = alloc<uint8_t>((size_t)n + 8);
			memset0(flag, (size_t)n);
			memset0(counter, 4);
			{
				Scope sc(*this, nm, n);
				int blocks = (n + 63) / 64; if (blocks > n_cu * seed_bpc) blocks = n_cu * seed_bpc;
				hipLaunchKernelGGL(k_seed_bwd2, dim3(blocks), dim3(64), 64 * (size_t)seed_row, stream, A, n, counter, seed_bwd_chunk, flag);
				ARX_HIP_CHECK(hipGetLastError());
			}
			seed_dbg_report(nm, n);
			if (seed_bwd_budget > 0) {
				Scope sc(*this, "seed_bwd_wave", n);
				hipLaunchKernelGGL(k_collect_heavy, dim3((n + 255) / 256), dim3(256), 0, stream, flag, n, f.t0, heavy, heavy + n);
				hipLaunchKernelGGL(k_seed_bwd_wave, dim3(n_cu * 16), dim3(64), 0, stream, A);
				ARX_HIP_CHECK(hipGetLastError());
			}
			return;
		}
		launch_seed_kernel(nm, k_seed_bwd, n, A, counter, seed_bpc, seed_bwd_chunk, seed_bwd_batch);
		seed_dbg_report(nm, n);
		if (seed_bwd_budget > 0) {
			Scope sc(*this, "seed_bwd_wave", n);
			hipLaunchKernelGGL(k_seed_bwd_wave, dim3(n_cu * 16), dim3(64), 0, stream, A);
			ARX_HIP_CHECK(hipGetLastError());
		}
	}
	template <class F> void run_seed_strat(const char *nm, int n, const F &f, int32_t *counter)
	{
		if (n <= 0) return;
		if (sw_simple) { launch(nm, n, f); return; }
		memset0(counter, 4);
		Scope sc(*this, nm, n);
		StratArgs A{f.ix, f.bases, f.base_off, f.lens, f.strat, f.n_strat, seed_row, seed_qn};
		int blocks = (n + 63) / 64; if (blocks > n_cu * strat_bpc) blocks = n_cu * strat_bpc;
		hipLaunchKernelGGL(k_strat_dyn, dim3(blocks), dim3(64), 64 * (size_t)seed_row, stream, A, n, counter, seed_chunk);
		ARX_HIP_CHECK(hipGetLastError());
	}
	// locate: persistent lanes with wave-level work distribution (hip_fm_coop.h); 32 waves per CU to cover the miss latency
	template <class F> void run_locate(const char *nm, int n, const F &f, int32_t *counter)
	{
		if (n <= 0) return;
		if (sw_simple) { launch(nm, n, f); return; }
		if (f.ix.sa40) { launch_wide(nm, n, f); return; } // the whole suffix array is resident: one load per occurrence, no walk to balance
		memset0(counter, 4);
		Scope sc(*this, nm, n);
		int blocks = (n + 255) / 256; if (blocks > n_cu * 8) blocks = n_cu * 8;
		hipLaunchKernelGGL(k_locate_dyn, dim3(blocks), dim3(256), 0, stream, f.ix, f.occ_seed, n, counter);
		ARX_HIP_CHECK(hipGetLastError());
	}
	// banded extension: 16 lanes per extension (hip_sw_coop.h); the query-length classes share one launch
	template <class F> void run_extend(const char *nm, const int32_t *n_class, int stride, const F &f)
	{
		int total = 0;
		for (int c = 0; c < EXT_CLASSES; ++c) total += n_class[c];
		if (total <= 0) return;
		if (sw_simple) {
			for (int c = 0; c < EXT_CLASSES; ++c) { F fc = f; fc.tasks = f.tasks + (size_t)c * stride; launch_rows(nm, n_class[c], fc, MAX_READ_LEN + 2); }
			return;
		}
		if (total >= ext_merge_below) { // big round: one launch per class, each at the occupancy its own register tiling allows
			for (int c = 0; c < EXT_CLASSES; ++c) {
				const int nc = n_class[c];
				if (nc <= 0) continue;
				Scope sc(*this, nm, nc);
				const ExtTask *tk = f.tasks + (size_t)c * stride;
				const int blocks = coop_blocks(nc);
#define ARX_EXT_LAUNCH(CN, CO) do { if (ext_old) hipLaunchKernelGGL((k_extend_b16<CO, true>), dim3(blocks), dim3(64), 0, stream, f.ix, f.bases, tk, f.res, nc); \
                                    else hipLaunchKernelGGL((k_extend_b16<CN, false>), dim3(blocks), dim3(64), 0, stream, f.ix, f.bases, tk, f.res, nc); } while (0)
				switch (c) {
				case 0: ARX_EXT_LAUNCH(2, 4); break;
				case 1: ARX_EXT_LAUNCH(3, 4); break;
				case 2: ARX_EXT_LAUNCH(4, 4); break;
				case 3: ARX_EXT_LAUNCH(6, 7); break;
				case 4: ARX_EXT_LAUNCH(8, 10); break;
				case 5: ARX_EXT_LAUNCH(10, 10); break;
				default: ARX_EXT_LAUNCH(16, 16); break;
				}
#undef ARX_EXT_LAUNCH
				ARX_HIP_CHECK(hipGetLastError());
			}
			return;
		}
		Scope sc(*this, nm, total);
		ExtClassShape sh;
		const int cap = n_cu * coop_bpc;
		int blocks = 0;
		for (int c = 0; c < EXT_CLASSES; ++c) {
			sh.n[c] = n_class[c];
			int nb = (n_class[c] + 3) / 4;
			if (nb > 0 && (total + 3) / 4 > cap) { nb = (int)((int64_t)nb * cap / ((total + 3) / 4)); if (nb < 1) nb = 1; } // share the grid cap by class size
			sh.nb[c] = nb; blocks += nb;
		}
		if (ext_old) hipLaunchKernelGGL(k_extend_classes_b<true>, dim3(blocks), dim3(64), 0, stream, f.ix, f.bases, f.tasks, stride, f.res, sh);
		else hipLaunchKernelGGL(k_extend_classes_b<false>, dim3(blocks), dim3(64), 0, stream, f.ix, f.bases, f.tasks, stride, f.res, sh);
		ARX_HIP_CHECK(hipGetLastError());
	}
	bool ext_old = getenv("ARX_EXT_OLD") && atoi(getenv("ARX_EXT_OLD")) != 0; // A/B: round 2's extension kernel (ext2_g16) on the same class lists
	// CIGARs of the gapped regions: 16 lanes per region (hip_nw_coop.h); f is pipeline.h's KReg2Aln
	template <class F> void run_reg2aln_nw(const char *nm, int n, const int32_t *n_class, const F &f, uint8_t *zbuf, const int32_t *z_off)
	{
		if (n <= 0) return;
		if (sw_simple) { launch_small(nm, n, f); return; }
		// one launch per band class: the four groups of a wavefront then run the same tiling, and the class kernel holds only the tilings the
		// class can need (its own and the doubled band's): 5 / 4 / 3 / 2 / 2 wavefronts per SIMD instead of 2 for all
		int32_t *punt = alloc<int32_t>((size_t)n + 4), *n_punt = punt + n;
		memset0(n_punt, 16);
		for (int c = 0; c < NW_CLASSES; ++c) {
			const int nc = n_class[c];
			if (nc <= 0) continue;
			Scope sc(*this, nm, nc);
			NwArgs A{f.ix, f.bases, f.base_off, f.lens, f.preg_off, f.n_regs, f.n_reads, f.pregs, f.alns, f.cig, f.cig_w, zbuf, z_off, f.nw_list, f.err,
			         f.class_list + (size_t)c * f.class_stride, punt, n_punt};
			const dim3 grid(coop_blocks(nc)), blk(64);
			switch (c) {
			case 0: hipLaunchKernelGGL((k_reg2aln_nw_g16<1, 2>), grid, blk, 0, stream, A, nc, (const int32_t *)nullptr); break;
			case 1: hipLaunchKernelGGL((k_reg2aln_nw_g16<2, 4>), grid, blk, 0, stream, A, nc, (const int32_t *)nullptr); break;
			case 2: hipLaunchKernelGGL((k_reg2aln_nw_g16<4, 8>), grid, blk, 0, stream, A, nc, (const int32_t *)nullptr); break;
			case 3: hipLaunchKernelGGL((k_reg2aln_nw_g16<8, 16>), grid, blk, 0, stream, A, nc, (const int32_t *)nullptr); break;
			default: hipLaunchKernelGGL((k_reg2aln_nw_g16<16, 16>), grid, blk, 0, stream, A, nc, (const int32_t *)nullptr); break;
			}
			ARX_HIP_CHECK(hipGetLastError());
		}
		{ // what the class kernels handed on (a third band, or a doubled band past the class's second tiling): all tilings, length read on the device
			Scope sc(*this, nm, 0);
			NwArgs A{f.ix, f.bases, f.base_off, f.lens, f.preg_off, f.n_regs, f.n_reads, f.pregs, f.alns, f.cig, f.cig_w, zbuf, z_off, f.nw_list, f.err, punt, punt, n_punt + 1};
			hipLaunchKernelGGL((k_reg2aln_nw_g16<1, 16>), dim3(n_cu * 2), dim3(64), 0, stream, A, 0, (const int32_t *)n_punt);
			ARX_HIP_CHECK(hipGetLastError());
		}
	}
	template <class F> void launch_rows(const char *nm, int n, const F &f, int words_per_thread)
	{
		if (n <= 0) return;
		Scope sc(*this, nm, n);
		int blocks = (n + 63) / 64; if (blocks > max_blocks()) blocks = max_blocks();
		size_t lds = (size_t)words_per_thread * 64 * 4;
		hipLaunchKernelGGL(k_rows<F>, dim3(blocks), dim3(64), lds, stream, f, n);
		ARX_HIP_CHECK(hipGetLastError());
	}
	// out[0..n] = exclusive prefix sums of in[0..n); returns the total as int64
	int64_t exclusive_scan(const int32_t *in, int32_t *out, int n)
	{
		Scope sc(*this, "scan", n);
		size_t need = 0;
		hipcub::DeviceScan::ExclusiveSum(nullptr, need, in, out, n, stream);
		size_t need2 = 0;
		hipcub::TransformInputIterator<int64_t, CastI64, const int32_t *> it(in, CastI64());
		hipcub::DeviceReduce::Sum(nullptr, need2, it, d_total, n, stream);
		if (need2 > need) need = need2;
		if (need > scan_tmp_bytes) { if (scan_tmp) hipFree(scan_tmp); ARX_HIP_CHECK(hipMalloc(&scan_tmp, need)); scan_tmp_bytes = need; }
		size_t nb = scan_tmp_bytes;
		ARX_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, nb, in, out, n, stream));
		nb = scan_tmp_bytes;
		ARX_HIP_CHECK(hipcub::DeviceReduce::Sum(scan_tmp, nb, it, d_total, n, stream));
		int64_t total = 0;
		d2h(&total, d_total, 8);
		int32_t t32 = (int32_t)(total < ((int64_t)1 << 31) ? total : 0x7fffffff);
		ARX_HIP_CHECK(hipMemcpyAsync(out + n, &t32, 4, hipMemcpyHostToDevice, stream));
		ARX_HIP_CHECK(hipStreamSynchronize(stream));
		return total;
	}
};

} // namespace arx
