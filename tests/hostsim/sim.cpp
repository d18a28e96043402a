// tests/hostsim/sim.cpp -- TEST DOUBLE, not part of the product.
//
// There is no GPU in the development container, so the device-side logic (the functors in arachne_amd/csrc/dev_*.h and
// the stage sequence of pipeline.h) is also compiled here for the host with a sequential "runtime": each launch is a
// plain loop over work items.  It exports the same C symbols as libarachne_amd.so so the same Python parity tests can
// drive it (`-m "not gpu"`), which lets indexing and tie-order bugs be found before a GPU box is spent on them.
// libarachne_amd.so itself never contains this code: it is built from arx_api.hip only and refuses to open without a GPU.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <map>
#include <string>
#include <vector>
#define ARX_DEV
#define ARX_DEVI inline
#define ARX_HDI inline
#define ARX_ATOMIC_OR(p, v) (*(p) |= (v))
#define ARX_ATOMIC_INC(p) ((*(p))++)
#define ARX_ATOMIC_ADD(p, v) sim_fetch_add((p), (v))
#define ARX_ATOMIC_MIN(p, v) (*(p) = *(p) < (v) ? *(p) : (v))
#define ARX_ATOMIC_CAS(p, c, v) sim_cas((p), (c), (v))
#define ARX_ATOMIC_ADD64(p, v) (*(p) += (v))
#define ARX_ATOMIC_MIN64(p, v) (*(p) = *(p) < (v) ? *(p) : (v))
#define ARX_ATOMIC_MAX64(p, v) (*(p) = *(p) > (v) ? *(p) : (v))
#define ARX_LOAD_SHARED(p) (*(p))
static inline int sim_fetch_add(int32_t *p, int v) { int o = *p; *p += v; return o; }
static inline int sim_cas(int32_t *p, int c, int v) { int o = *p; if (o == c) *p = v; return o; }
static long sim_rescue_fast_hits = 0, sim_rescue_fast_fallbacks = 0;
static long sim_rescue_calls = 0, sim_rescue_ins = 0, sim_rescue_skipped = 0, sim_rescue_nsum = 0, sim_rescue_nmax = 0, sim_rescue_n2sum = 0;
static void sim_stat_rescue(int n, bool ins, int clean)
{
	++sim_rescue_calls; sim_rescue_ins += ins; sim_rescue_skipped += (!ins && clean); sim_rescue_nsum += n; sim_rescue_n2sum += (long)n * n; if (n > sim_rescue_nmax) sim_rescue_nmax = n;
}
struct SimStatPrinter { ~SimStatPrinter() { if (getenv("ARX_RESCUE_STATS")) fprintf(stderr, "[sim] rescue applies %ld, inserted %ld, dedup skipped %ld, mean n %.1f, mean n^2 %.1f, max n %ld; fast inserts %ld, fallbacks %ld\n", sim_rescue_calls, sim_rescue_ins, sim_rescue_skipped, sim_rescue_calls ? (double)sim_rescue_nsum / sim_rescue_calls : 0.0, sim_rescue_calls ? (double)sim_rescue_n2sum / sim_rescue_calls : 0.0, sim_rescue_nmax, sim_rescue_fast_hits, sim_rescue_fast_fallbacks); } } sim_stat_printer;
#define ARX_STAT_RESCUE(pair, n, inserts, clean) sim_stat_rescue((n), (inserts), (clean))
// the rescue pre-filter against the DP it replaces: a filtered task must score below min_seed_len (always checked)
static long sim_swf_tasks = 0, sim_swf_filtered = 0, sim_swf_low = 0;
static void sim_sw_filter_check(bool pass, int score)
{
	++sim_swf_tasks; sim_swf_filtered += !pass; sim_swf_low += score < 19;
	if (!pass && score >= 19) { fprintf(stderr, "[sim] rescue pre-filter dropped an alignment of score %d\n", score); abort(); }
}
struct SimSwfPrinter { ~SimSwfPrinter() { if (getenv("ARX_RESCUE_STATS")) fprintf(stderr, "[sim] rescue SWs %ld, below min_seed_len %ld, filtered %ld\n", sim_swf_tasks, sim_swf_low, sim_swf_filtered); } } sim_swf_printer;
#define ARX_SW_FILTER_CHECK(pass, score) sim_sw_filter_check((pass), (score))
static long sim_bwd_hist_n[8], sim_bwd_hist_ext[8], sim_bwd_ext_by_n[8], sim_bwd_max_ext;
static void sim_stat_bwd(int n, int ext)
{
	int a = 0; while (a < 7 && (1 << a) < n) ++a;
	int b = 0; while (b < 7 && (16 << b) <= ext) ++b;
	++sim_bwd_hist_n[a]; ++sim_bwd_hist_ext[b]; sim_bwd_ext_by_n[a] += ext; if (ext > sim_bwd_max_ext) sim_bwd_max_ext = ext;
}
struct SimBwdPrinter { ~SimBwdPrinter() { if (!getenv("ARX_BWD_STATS")) return; fprintf(stderr, "[sim] backward tasks by list length (<=1,2,4,..,128):"); for (int a = 0; a < 8; ++a) fprintf(stderr, " %ld", sim_bwd_hist_n[a]);
	fprintf(stderr, "\n[sim]   extensions spent there:"); for (int a = 0; a < 8; ++a) fprintf(stderr, " %ld", sim_bwd_ext_by_n[a]);
	fprintf(stderr, "\n[sim]   tasks by extensions (<16,<32,..,>=1024):"); for (int a = 0; a < 8; ++a) fprintf(stderr, " %ld", sim_bwd_hist_ext[a]); fprintf(stderr, "; max %ld\n", sim_bwd_max_ext); } } sim_bwd_printer;
#define ARX_STAT_BWD(n, ext) sim_stat_bwd((n), (ext))
// ARX_RESCUE_FAST=0 switches dedup_insert() off; ARX_RESCUE_CHECK=1 runs the general path next to it on a copy and aborts on any difference
static int sim_rescue_fast_f() { const char *e = getenv("ARX_RESCUE_FAST"); return e ? atoi(e) : 1; }
static int sim_rescue_check_f() { const char *e = getenv("ARX_RESCUE_CHECK"); return e ? atoi(e) : 0; }
#define ARX_RESCUE_FAST sim_rescue_fast_f()
#define ARX_RESCUE_CROSSCHECK_BEGIN(ma, n, b) const int sim_rescue_check = sim_rescue_check_f(); std::vector<Reg> chk_(ma, ma + (n)); if (sim_rescue_check) { int at_ = 0; while (at_ < (n) && !(chk_[at_].score < (b).score)) ++at_; chk_.insert(chk_.begin() + at_, (b)); }
#define ARX_RESCUE_CROSSCHECK_END(ix, ma, m) do { if ((m) >= 0) ++sim_rescue_fast_hits; else ++sim_rescue_fast_fallbacks; if (sim_rescue_check && (m) >= 0) { \
		std::vector<Reg> t_(chk_.size() + 1); std::vector<int> i_(chk_.size() + 1); \
		const int m2_ = sort_dedup_patch(ix, 0, (int)chk_.size(), chk_.data(), t_.data(), i_.data(), 0); \
		if (m2_ != (m) || memcmp(chk_.data(), (ma), sizeof(Reg) * (size_t)m2_)) { fprintf(stderr, "[sim] dedup_insert differs from mem_sort_dedup_patch: %d vs %d regions\n", (m), m2_); abort(); } } } while (0)
#include "../../arachne_amd/csrc/arx_dev.h"
#include "../../arachne_amd/csrc/index_build.h"

#include <algorithm>
namespace arx {
// sequential stand-in for hip_block.h's HipBlock: one "lane" runs every parallel-for in index order
// (ARX_SIM_PFOR=1: descending, =2: a stride permutation -- a phase whose result depends on the lane order is a race on the GPU)
constexpr int SMALL_LANES = 256, SMALL_SORT = 1024; // hip_block.h
struct SimBlock {
	int order = getenv("ARX_SIM_PFOR") ? atoi(getenv("ARX_SIM_PFOR")) : 0;
	template <class F> void pfor(int n, F f)
	{
		if (order == 1) { for (int i = n - 1; i >= 0; --i) f(i); return; }
		if (order == 2 && n > 2) { // i -> (i * step) mod n with gcd(step, n) = 1
			int64_t step = 7919; while (std::__gcd<int64_t>(step, n) != 1) ++step;
			for (int64_t i = 0; i < n; ++i) f((int)((i * step + 3) % n));
			return;
		}
		for (int i = 0; i < n; ++i) f(i);
	}
	template <class F> void single(F f) { f(); }
	int exclusive_scan(const int32_t *in, int32_t *out, int n) { int t = 0; for (int i = 0; i < n; ++i) { int x = in[i]; out[i] = t; t += x; } out[n] = t; return t; }
	void sort_kv(uint64_t *k, int32_t *v, int P)
	{
		std::vector<std::pair<std::pair<uint64_t, uint32_t>, int> > a(P);
		for (int i = 0; i < P; ++i) a[i] = std::make_pair(std::make_pair(k[i], (uint32_t)v[i]), i);
		std::sort(a.begin(), a.end());
		for (int i = 0; i < P; ++i) { k[i] = a[i].first.first; v[i] = (int32_t)a[i].first.second; }
	}
	template <class F> void argmax(int n, F keyf, uint64_t *key, int *idx)
	{
		uint64_t bk = 0; int bi = 0x7fffffff;
		for (int i = 0; i < n; ++i) { uint64_t x = keyf(i); if (x > bk) { bk = x; bi = i; } }
		*key = bk; *idx = bi;
	}
};
struct KernelTimer { double ms = 0; int64_t calls = 0, items = 0; };
struct SimRT {
	static const char *name() { return "hostsim"; }
	static arx::BwtSaFn bwt_sa_fn() { return arx::build_bwt_sa_host; }
	std::map<std::string, KernelTimer> tm;
	std::string init(int) { return ""; }
	void bind() {}
	void set_timing(bool) {}
	// poison fresh memory: hipMalloc does not zero either, so nothing may rely on it
	template <class T> T *alloc(size_t n) { size_t b = (n ? n : 1) * sizeof(T); void *p = malloc(b); memset(p, 0xAB, b); return (T *)p; }
	void free(void *p) { ::free(p); }
	void seed_prepare(const uint8_t *, const int32_t *, const int32_t *, int) {}
	bool rescue_heavy_ok() const { return getenv("ARX_SIM_RESCUE_HEAVY") != nullptr; } // the host double can run the split (same serial code on the flagged pairs)
	void aux_join() {}
	bool dedup_heavy_ok() const { return getenv("ARX_SIM_DEDUP_HEAVY") != nullptr; }
	template <class F> void run_dedup_heavy(const char *nm, int, const F &f) { tm[nm].calls++; for (int i = 0; i < *f.n_heavy; ++i) f.one_thread(f.heavy_list[i], f.eh); }
	bool chain_heavy_ok() const { return getenv("ARX_SIM_CHAIN_HEAVY") != nullptr; }
	template <class F> void run_chain_heavy(const char *nm, int, const F &f) { F g = f; g.heavy_list = nullptr; g.mid_list = nullptr; tm[nm].calls++; for (int i = 0; i < *f.n_heavy; ++i) g(f.heavy_list[i], 0); }
	template <class F> void run_rescue_heavy(const char *nm, int n, const int32_t *list, const F &f) { F g = f; g.heavy = nullptr; tm[nm].calls++; for (int i = 0; i < n; ++i) g(list[i], 0); }
	std::vector<uint8_t> stage_mem;
	void *stage(size_t bytes) { stage_mem.assign(bytes + 8, 0xCD); return stage_mem.data(); }
	void h2d_staged(void *d, const void *s, size_t bytes) { memcpy(d, s, bytes); }
	static int host_register(void *, size_t) { return 0; }
	static int host_unregister(void *) { return 0; }
	static bool host_pinned(const void *) { return false; }
	void h2d_pinned(void *d, const void *s, size_t bytes) { memcpy(d, s, bytes); }
	template <class T> T *palloc(size_t n) { return alloc<T>(n); }
	void pfree(void *p) { ::free(p); }
	void arena_reset() {}
	std::vector<size_t> arena_mark() const { return {}; }
	void arena_rewind(const std::vector<size_t> &) {}
	void h2d(void *d, const void *s, size_t b) { if (b) memcpy(d, s, b); }
	void d2h(void *d, const void *s, size_t b) { if (b) memcpy(d, s, b); }
	void d2h_async(void *d, const void *s, size_t b) { if (b) memcpy(d, s, b); }
	uint64_t free_bytes() const { return (uint64_t)1 << 32; } // test double: tables of up to 4^12 entries
	void d2d(void *d, const void *s, size_t b) { if (b) memcpy(d, s, b); }
	void memset0(void *d, size_t b) { memset(d, 0, b); }
	void memset_bytes(void *d, int v, size_t b) { memset(d, v, b); }
	void sync() {}
	int max_slots() const { return 1; }
	int max_slots_small() const { return 1; }
	int max_seed_slots() const { return 1; }
	void set_seed_read_len(int) {}
	std::map<std::string, KernelTimer> &timers() { return tm; }
	void timers_reset(bool) { tm.clear(); }
	template <class F> void launch(const char *nm, int n, const F &f) { tm[nm].calls++; tm[nm].items += n; for (int i = 0; i < n; ++i) f(i, 0); }
	template <class F> void launch_small(const char *nm, int n, const F &f) { launch(nm, n, f); }
	template <class F> void launch_wide(const char *nm, int n, const F &f) { launch(nm, n, f); }
	template <class F> void launch_block(const char *nm, int n, const F &f, const uint8_t * = nullptr) { tm[nm].calls++; tm[nm].items += n; SimBlock blk; for (int i = 0; i < n; ++i) f(i, blk); }
	template <class F> void launch_cold(const char *nm, int n, const F &f) { launch(nm, n, f); }
	void merge_sort_fail(uint32_t *err) { if (g_arx_sort_fail) *err |= ERR_INTERNAL; }
	template <class F> void run_sw_u8(const char *nm, int n, const F &f, int max_len) { launch_rows(nm, n, f, 16 * ((max_len + 15) / 16)); }
	template <class F> void run_seed_fwd1(const char *nm, int n, const F &f, int32_t *) { launch(nm, n, f); }
	template <class F> void run_seed_fwd2(const char *nm, int n, const F &f, int32_t *) { launch(nm, n, f); }
	template <class F> void run_seed_bwd(const char *nm, int n, const F &f, int32_t *) { launch(nm, n, f); }
	template <class F> void run_seed_strat(const char *nm, int n, const F &f, int32_t *) { launch(nm, n, f); }
	template <class F> void run_locate(const char *nm, int n, const F &f, int32_t *) { launch(nm, n, f); }
	template <class F> void run_extend(const char *nm, const int32_t *n_class, int stride, const F &f)
	{
		for (int c = 0; c < EXT_CLASSES; ++c) { F fc = f; fc.tasks = f.tasks + (size_t)c * stride; launch_rows(nm, n_class[c], fc, MAX_READ_LEN + 2); }
	}
	template <class F> void run_reg2aln_nw(const char *nm, int n, const int32_t *, const F &f, uint8_t *, const int32_t *) { launch(nm, n, f); }
	template <class F> void launch_rows(const char *nm, int n, const F &f, int words)
	{
		std::vector<uint32_t> row(words + 8);
		tm[nm].calls++; tm[nm].items += n;
		for (int i = 0; i < n; ++i) f(i, 0, row.data(), 1);
	}
	int64_t exclusive_scan(const int32_t *in, int32_t *out, int n)
	{
		int64_t t = 0;
		for (int i = 0; i < n; ++i) { out[i] = (int32_t)t; t += in[i]; }
		out[n] = (int32_t)t;
		return t;
	}
};
} // namespace arx

#include "../../arachne_amd/csrc/api_impl.h"
ARX_DEFINE_C_API(arx::SimRT)

// test entry: ksw_align2's two passes with the row's F by the plain recurrence (dev_sw.h u8_pass, exact_f: what the rescue kernel's scan
// computes) -- tests/test_sw_prefilter.py compares it with the restatement's striped pass + lazy-F loop
extern "C" void arx_test_sw_exact_f(const uint8_t *q, int qlen, const uint8_t *t, int tlen, int xtra, int exact_f, int32_t *out)
{
	std::vector<uint32_t> row((size_t)(16 * ((qlen + 15) / 16) + 16));
	std::vector<uint8_t> rowmax((size_t)tlen + 8);
	const arx::U8Res r = arx::u8_align(q, qlen, t, tlen, xtra, row.data(), 1, rowmax.data(), exact_f != 0);
	out[0] = r.score; out[1] = r.te; out[2] = r.qe; out[3] = r.score2; out[4] = r.te2; out[5] = r.tb; out[6] = r.qb;
}

// test entry: the rescue pre-filter alone (tests/test_sw_prefilter.py checks it against the oracle's ksw_align2)
extern "C" int arx_test_sw_prefilter(const uint8_t *q, int qlen, const uint8_t *t, int tlen) { return arx::sw_prefilter_serial(q, qlen, t, tlen) ? 1 : 0; }

// test entry: dedup_insert() (the one-scan insertion into a list that is a fixed point of mem_sort_dedup_patch) against the general
// path on random lists built to collide -- clustered regions on a few contigs with overlapping spans, so that both sides of the
// inserted region hold redundant neighbours, stoppers and survivors.  Returns the number of cases compared (< 0: first mismatch).
extern "C" long arx_test_dedup_insert(unsigned seed, int iters, long *n_fast, long *n_gone_cases, long *n_b_gone)
{
	using namespace arx;
	uint64_t x = 0x9E3779B97F4A7C15ull * (seed + 1);
	auto rnd = [&](int m) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (int)(x % (uint64_t)m); };
	auto make = [&](Reg &r) {
		r = Reg();
		r.rid = rnd(3);
		const int64_t base = 100000 * r.rid + 200 * rnd(6);           // a handful of clusters
		r.rb = base + rnd(40); r.re = r.rb + 100 + rnd(60);
		r.qb = rnd(30); r.qe = r.qb + 100 + rnd(20);
		r.score = 60 + rnd(60); r.truesc = r.score; r.secondary = -1; r.n_comp = 1;
	};
	IndexView ix = IndexView();
	long cases = 0;
	*n_fast = *n_gone_cases = *n_b_gone = 0;
	for (int it = 0; it < iters; ++it) {
		const int n0 = 2 + rnd(it % 7 == 0 ? 300 : 40);
		std::vector<Reg> a(n0 + 2), tmp(n0 + 2), chk;
		std::vector<int> idx(n0 + 2);
		for (int i = 0; i < n0; ++i) make(a[i]);
		int n = sort_dedup_patch(ix, 0, n0, a.data(), tmp.data(), idx.data(), 0);   // a fixed point of the pass
		if (n < 2) continue;
		Reg b; make(b);
		if (rnd(3) == 0) { const Reg &q = a[rnd(n)]; b.rid = q.rid; b.rb = q.rb + rnd(5) - 2; b.re = q.re + rnd(5) - 2; b.qb = q.qb; b.qe = q.qe; b.score = q.score + rnd(21) - 10; }
		chk.assign(a.begin(), a.begin() + n);
		{ int at = 0; while (at < n && !(chk[at].score < b.score)) ++at; chk.insert(chk.begin() + at, b); }
		std::vector<Reg> t2(n + 2); std::vector<int> i2(n + 2);
		const int m2 = sort_dedup_patch(ix, 0, n + 1, chk.data(), t2.data(), i2.data(), 0);
		a.resize(n + 2); tmp.resize(n + 2); idx.resize(n + 2);
		const int m = dedup_insert(b, a.data(), n, tmp.data(), idx.data());
		++cases;
		if (m < 0) continue; // tie: the caller takes the general path
		++*n_fast;
		if (m != n + 1) { if (m == n) ++*n_b_gone; else ++*n_gone_cases; }
		if (m != m2 || memcmp(a.data(), chk.data(), sizeof(Reg) * (size_t)m)) return -cases;
	}
	return cases;
}

// test entry: text mode's word-parallel comparison (dev_fm.h text_match_chunk on a row of 4-bit codes, what the wavefront kernels run) against
// gapfree_counts (four pairs per step on words) against the pair-by-pair walk of bwa_gen_cigar2's shortcut, on random packed strands: both
// strands, every phase of the region's start within a byte, regions that end at either end of a strand, lengths 1 .. 260, reads that mostly
// match with planted differences and ambiguous bases, unaligned read addresses.  Returns the cases compared (< 0: first mismatch).
extern "C" long arx_test_gapfree(unsigned seed, int iters)
{
	using namespace arx;
	uint64_t x = 0x9E3779B97F4A7C15ull * (seed + 7);
	auto rnd = [&](int m) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (int)(x % (uint64_t)m); };
	long cases = 0;
	for (int it = 0; it < iters; ++it) {
		const int64_t l_pac = 8 + rnd(it % 4 == 0 ? 40 : 900);
		std::vector<uint8_t> store((size_t)(l_pac / 4 + 1), 0); // no spare bytes behind the strand: a read past its last byte shows under a sanitizer
		uint8_t *pac = store.data();
		for (int64_t p = 0; p < l_pac; ++p) pac[p >> 2] |= (uint8_t)(rnd(4) << ((~p & 3) << 1));
		IndexView ix = IndexView();
		ix.pac = pac; ix.l_pac = l_pac; ix.seq_len = (uint64_t)(2 * l_pac);
		for (int rep = 0; rep < 8; ++rep) {
			const bool rev = rnd(2) == 1;
			int l = 1 + rnd(260); if (l > l_pac) l = (int)l_pac;
			const int edge = rnd(4);
			int64_t rb = edge == 0 ? 0 : edge == 1 ? l_pac - l : rnd((int)(l_pac - l + 1));
			if (rev) rb += l_pac;
			const int shift = rnd(4);
			std::vector<uint8_t> q((size_t)l + 8, 9);
			uint8_t *qp = q.data() + shift;
			const int p_diff = rnd(3) == 0 ? 2 : 20, p_amb = rnd(3) == 0 ? 3 : 40;
			for (int j = 0; j < l; ++j) {
				int b = ref_base(ix, rb + j);
				if (rnd(p_diff) == 0) b = (b + 1 + rnd(3)) & 3;
				if (rnd(p_amb) == 0) b = 4;
				qp[j] = (uint8_t)b;
			}
			int s_want = 0, mm_want = 0;
			for (int j = 0; j < l; ++j) { const int t = ref_base(ix, rb + j), c = qp[j]; s_want += sc_mat(t, c); mm_want += c != t; }
			int s_got = -12345, mm_got = -1;
			gapfree_counts(ix, qp, rb, l, &s_got, &mm_got);
			if (s_got != s_want || mm_got != mm_want) return -(cases + 1);
			++cases;
		}
	}
	return cases;
}

// the base-by-base form on random packed strands and reads: matching stretches of every length, both strands, chunks that end at the
// strand boundary and at either end of the text, ambiguous bases, reads that end inside the chunk.  Returns the cases compared (< 0: first mismatch).
extern "C" long arx_test_text_match(unsigned seed, int iters)
{
	using namespace arx;
	uint64_t x = 0x9E3779B97F4A7C15ull * (seed + 1);
	auto rnd = [&](int m) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (int)(x % (uint64_t)m); };
	long cases = 0;
	for (int it = 0; it < iters; ++it) {
		const int64_t l_pac = 40 + rnd(it % 5 == 0 ? 60 : 700);
		std::vector<uint8_t> store((size_t)(l_pac / 4 + 1 + 16 + 64), 0);
		uint8_t *pac = store.data() + 16;
		for (int64_t p = 0; p < l_pac; ++p) pac[p >> 2] |= (uint8_t)(rnd(4) << ((~p & 3) << 1));
		IndexView ix = IndexView();
		ix.pac = pac; ix.l_pac = l_pac; ix.seq_len = (uint64_t)(2 * l_pac);
		for (int rep = 0; rep < 8; ++rep) {
			const int len = 20 + rnd(236), i = rnd(len);
			const uint64_t t = (uint64_t)rnd((int)(2 * l_pac));
			std::vector<uint8_t> q((size_t)len + 80, 4);
			const int m_want = rnd(3) == 0 ? rnd(8) : rnd(140);          // equal bases from (i, t) on, then a difference
			for (int j = 0; j < len; ++j) q[(size_t)j] = (uint8_t)rnd(4);
			for (int j = 0; i + j < len && t + (uint64_t)j < ix.seq_len && j <= m_want; ++j) {
				const int b = ref_base(ix, (int64_t)t + j);
				q[(size_t)(i + j)] = (uint8_t)(j < m_want ? b : (b + 1 + rnd(3)) & 3);
			}
			if (rnd(6) == 0) q[(size_t)(i + rnd(len - i))] = 4;
			std::vector<uint32_t> rowbuf((size_t)(len + 80) / 8 + 12, 0x44444444u);
			uint8_t *row = (uint8_t *)rowbuf.data();
			for (int j = 0; j < len + 72; ++j) { const int b = j < len ? q[(size_t)j] : 4; row[j >> 1] = (uint8_t)((row[j >> 1] & ~(15 << ((j & 1) << 2))) | b << ((j & 1) << 2)); }
			const uint32_t *w = (const uint32_t *)ix.pac + text_chunk_word(ix, t);
			bool more_a = false, more_b = false;
			const int a = text_match_chunk(ix, w[0], w[1], w[2], w[3], t, QBytes{q.data()}, i, len, &more_a);
			const int b = text_match_chunk(ix, w[0], w[1], w[2], w[3], t, QNibbles{row}, i, len, &more_b);
			++cases;
			if (a != b || more_a != more_b) return -cases;
			// and the base-by-base form against the text itself
			int m = 0;
			while (m < a + 1 && i + m < len && t + (uint64_t)m < ix.seq_len && q[(size_t)(i + m)] == ref_base(ix, (int64_t)t + m)) ++m;
			if (m < a) return -cases;
			if (m == a + 1 && !more_a) return -cases; // it stopped although the next base is equal, within the read and the text
		}
	}
	return cases;
}

// test entry: an extension's arithmetic with the children's sizes in 32 bits (dev_fm.h ext_finish<true>, what k_seed_bwd_g<true> runs) against the
// general 40-bit form and against extend1, on intervals of the context's index: every symbol as a start, then random walks of extensions in
// both directions.  Returns the extensions compared (< 0: first mismatch; 0: the index does not qualify).
extern "C" long arx_test_ext_fit32(arx_ctx *h, unsigned seed, int walks)
{
	using namespace arx;
	const IndexView &ix = ((Context<SimRT> *)h)->ix;
	if (!occ_counts_fit32(ix)) return 0;
	uint64_t x = 0x9E3779B97F4A7C15ull * (seed + 3);
	auto rnd = [&](int m) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (int)(x % (uint64_t)m); };
	long n = 0;
	for (int w = 0; w < walks; ++w) {
		Biv ik = set_intv(ix, rnd(4));
		for (int step = 0; step < 40 && ik.s > 0; ++step) {
			const int back = rnd(2), c = rnd(4);
			ExtLoad L;
			ext_issue(ix, ik, back, true, L);
			const Biv a = ext_finish<true>(ix, ik, back, c, L), b = ext_finish<false>(ix, ik, back, c, L), e = extend1(ix, ik, back, c);
			++n;
			if (a.k != b.k || a.l != b.l || a.s != b.s || a.k != e.k || a.l != e.l || a.s != e.s) return -n;
			ik = a;
		}
	}
	return n;
}

// test entry: the backward comparison of text mode (dev_fm.h text_match_back: the entry-parallel backward sweeps) against the text itself
extern "C" long arx_test_text_match_back(unsigned seed, int iters)
{
	using namespace arx;
	uint64_t x = 0x9E3779B97F4A7C15ull * (seed + 7);
	auto rnd = [&](int m) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (int)(x % (uint64_t)m); };
	long cases = 0;
	for (int it = 0; it < iters; ++it) {
		const int64_t l_pac = 40 + rnd(it % 5 == 0 ? 60 : 700);
		std::vector<uint8_t> store((size_t)(l_pac / 4 + 1 + 16 + 64), 0);
		uint8_t *pac = store.data() + 16;
		for (int64_t p = 0; p < l_pac; ++p) pac[p >> 2] |= (uint8_t)(rnd(4) << ((~p & 3) << 1));
		IndexView ix = IndexView();
		ix.pac = pac; ix.l_pac = l_pac; ix.seq_len = (uint64_t)(2 * l_pac);
		for (int rep = 0; rep < 8; ++rep) {
			const int len = 20 + rnd(236), i = rnd(len);
			const uint64_t t = (uint64_t)rnd((int)(2 * l_pac));
			std::vector<uint8_t> q((size_t)len + 8, 4);
			const int m_want = rnd(3) == 0 ? rnd(6) : rnd(60);          // equal bases from (i, t) downwards, then a difference
			for (int j = 0; j < len; ++j) q[(size_t)j] = (uint8_t)rnd(4);
			for (int j = 0; i - j >= 0 && (int64_t)t - j >= 0 && j <= m_want; ++j) {
				const int b = ref_base(ix, (int64_t)t - j);
				q[(size_t)(i - j)] = (uint8_t)(j < m_want ? b : (b + 1 + rnd(3)) & 3);
			}
			if (rnd(6) == 0) q[(size_t)rnd(i + 1)] = 4;
			std::vector<uint32_t> rowbuf((size_t)(len + 16) / 8 + 4, 0x44444444u);
			uint8_t *row = (uint8_t *)rowbuf.data();
			for (int j = 0; j < len; ++j) row[j >> 1] = (uint8_t)((row[j >> 1] & ~(15 << ((j & 1) << 2))) | q[(size_t)j] << ((j & 1) << 2));
			// walk as the kernel does: chunk after chunk until the comparison stops
			int ii = i; uint64_t tt = t; int total = 0; bool more = true;
			while (more) {
				const uint32_t *w = (const uint32_t *)ix.pac + text_chunk_word_back(ix, tt);
				const int wi = ii >> 3;
				auto qword = [&](int k) { return rowbuf[(size_t)(wi - k > 0 ? wi - k : 0)]; };
				const int m = text_match_back(ix, w[0], w[1], w[2], w[3], tt, qword(0), qword(1), qword(2), qword(3), ii, &more);
				total += m; ii -= m; tt -= (uint64_t)m;
				if (more && (ii < 0 || m == 0)) return -(cases + 1);   // `more` promises progress and room
			}
			int truth = 0;
			while (i - truth >= 0 && (int64_t)t - truth >= 0 && q[(size_t)(i - truth)] < 4 && q[(size_t)(i - truth)] == ref_base(ix, (int64_t)t - truth)) ++truth;
			++cases;
			if (total != truth) return -cases;
		}
	}
	return cases;
}

// the wavefront routines exist on the GPU only (include/arachne_amd.h): nothing to test here
extern "C" int arx_selftest_wave_sort(int32_t, int32_t, int64_t, int64_t *n_bad) { if (n_bad) *n_bad = 0; return ARX_E_DEVICE; }
