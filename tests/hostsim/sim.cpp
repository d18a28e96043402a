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
#include "../../arachne_amd/csrc/arx_dev.h"

namespace arx {
struct KernelTimer { double ms = 0; int64_t calls = 0, items = 0; };
struct SimRT {
	static const char *name() { return "hostsim"; }
	std::map<std::string, KernelTimer> tm;
	std::string init(int) { return ""; }
	void bind() {}
	void set_timing(bool) {}
	// poison fresh memory: hipMalloc does not zero either, so nothing may rely on it
	template <class T> T *alloc(size_t n) { size_t b = (n ? n : 1) * sizeof(T); void *p = malloc(b); memset(p, 0xAB, b); return (T *)p; }
	void free(void *p) { ::free(p); }
	template <class T> T *palloc(size_t n) { return alloc<T>(n); }
	void pfree(void *p) { ::free(p); }
	void arena_reset() {}
	std::vector<size_t> arena_mark() const { return {}; }
	void arena_rewind(const std::vector<size_t> &) {}
	void h2d(void *d, const void *s, size_t b) { if (b) memcpy(d, s, b); }
	void d2h(void *d, const void *s, size_t b) { if (b) memcpy(d, s, b); }
	void memset0(void *d, size_t b) { memset(d, 0, b); }
	void sync() {}
	int max_slots() const { return 1; }
	int max_slots_small() const { return 1; }
	std::map<std::string, KernelTimer> &timers() { return tm; }
	void timers_reset(bool) { tm.clear(); }
	template <class F> void launch(const char *nm, int n, const F &f) { tm[nm].calls++; tm[nm].items += n; for (int i = 0; i < n; ++i) f(i, 0); }
	template <class F> void launch_small(const char *nm, int n, const F &f) { launch(nm, n, f); }
	template <class F> void launch_cold(const char *nm, int n, const F &f) { launch(nm, n, f); }
	template <class F> void run_sw_u8(const char *nm, int n, const F &f, int max_len) { launch_rows(nm, n, f, 16 * ((max_len + 15) / 16)); }
	template <class F> void run_locate(const char *nm, int n, const F &f, int32_t *) { launch(nm, n, f); }
	template <class F> void run_extend(const char *nm, int n, const F &f, int max_len) { launch_rows(nm, n, f, max_len + 1); }
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
