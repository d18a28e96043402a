// tools/calib_fetch.hip -- calibration of rocprofv3's FETCH_SIZE for the access pattern of the seeding kernel
// (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern").  Every lane reads ONE random, distinct 64-byte block of a 2 GiB table with the same four 16-byte loads as
// load_block() in arachne_amd/csrc/dev_fm.h; the table is larger than the Infinity Cache and no block is read twice, so
// the algorithmic byte count is exactly n_blocks * 64.  Build: hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

struct alignas(16) Q16 { uint32_t x, y, z, w; };

__global__ void k_calib_blocks(const uint32_t *tab, uint64_t n_tab_blocks, uint64_t n, uint64_t mul, uint64_t add, uint32_t *sink)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t b = (i * mul + add) & (n_tab_blocks - 1); // odd multiplier, power-of-two table: a permutation, every block once
	const Q16 *p = (const Q16 *)(tab + b * 16);
	const Q16 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
	const uint32_t s = a0.x ^ a0.w ^ a1.y ^ a1.z ^ a2.x ^ a2.w ^ a3.y ^ a3.z;
	if (s == 0x12345678u) sink[0] = s; // keep the loads
}

int main()
{
	const uint64_t n_tab_blocks = 1ull << 25; // 2 GiB
	const uint64_t n = 1ull << 24;            // 16 Mi blocks read = 1 GiB algorithmic
	uint32_t *tab = nullptr, *sink = nullptr;
	if (hipMalloc((void **)&tab, n_tab_blocks * 64) != hipSuccess || hipMalloc((void **)&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
	hipMemset(tab, 1, n_tab_blocks * 64);
	hipDeviceSynchronize();
	for (int rep = 0; rep < 3; ++rep) {
		hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
		hipEventRecord(a);
		hipLaunchKernelGGL(k_calib_blocks, dim3((unsigned)(n / 256)), dim3(256), 0, 0, tab, n_tab_blocks, n, 0x9E3779B97F4A7C15ull | 1, 12345ull + rep * 7919, sink);
		hipEventRecord(b); hipEventSynchronize(b);
		float ms = 0; hipEventElapsedTime(&ms, a, b);
		printf("calib launch %d: %.3f ms, %.1f GB/s algorithmic (%llu blocks x 64 B)\n", rep, ms, n * 64 / (ms * 1e6), (unsigned long long)n);
	}
	return 0;
}
