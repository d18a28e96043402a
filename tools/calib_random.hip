// tools/calib_random.hip -- what the memory system of an MI355X delivers for the FM-index access pattern at GRCh38 footprint:
// every lane walks a chain of DEPENDENT random 64-byte block reads (next index = hash of the previous one, mixed with the loaded
// data so that the load must return first), `nb` blocks per step in flight per lane (1 or 2: a bi-interval extension touches two),
// tables of 1 .. 8 GiB, 16 waves of 64 lanes per CU resident (4 per SIMD, as the seeding kernels).  tools/calib_fetch.hip reads its
// table with a fixed stride (an odd multiplier): that spreads perfectly over channels and banks and overstates what random rows get.
// Build: hipcc --offload-arch=gfx950 -O3 tools/calib_random.hip -o calib_random
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct alignas(16) Q16 { uint32_t x, y, z, w; };
__device__ __forceinline__ uint64_t mix(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
template <int NB> __global__ void __launch_bounds__(64, 4) k_walk(const uint32_t *tab, uint64_t n_blocks, int steps, uint32_t *sink)
{
	uint64_t s = mix((uint64_t)blockIdx.x * 64 + threadIdx.x);
	uint32_t acc = 0;
	for (int i = 0; i < steps; ++i) {
		uint64_t b[NB]; Q16 h0[NB], h1[NB]; uint64_t q[NB];
#pragma unroll
		for (int u = 0; u < NB; ++u) { b[u] = mix(s + u) % n_blocks; const Q16 *p = (const Q16 *)(tab + b[u] * 16); h0[u] = p[0]; h1[u] = p[1]; q[u] = *(const uint64_t *)(tab + b[u] * 16 + 8 + 2 * (int)(s & 3)); } // head 32 B + one quarter: what a count reads
		uint32_t d = 0;
#pragma unroll
		for (int u = 0; u < NB; ++u) d ^= h0[u].x ^ h1[u].w ^ (uint32_t)q[u];
		acc ^= d;
		s = mix(s ^ (d & 0)); // depends on the loads (the table holds a constant)
	}
	if (acc == 0x12345678u) sink[0] = acc;
}
int main()
{
	uint32_t *tab = nullptr, *sink = nullptr;
	const uint64_t max_blocks = 1ull << 27; // 8 GiB
	if (hipMalloc((void **)&tab, max_blocks * 64) != hipSuccess || hipMalloc((void **)&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
	(void)hipMemset(tab, 1, max_blocks * 64);
	(void)hipDeviceSynchronize();
	const int steps = 400, blocks = 256 * 16;
	const double gib[] = {0.25, 1, 2, 2.5, 3, 3.5, 4, 6, 8};
	for (double g : gib)
		for (int nb = 1; nb <= 2; nb += 1) {
			const uint64_t n_blocks = (uint64_t)(g * (1ull << 30) / 64);
			hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
			float best = 1e9f;
			for (int rep = 0; rep < 3; ++rep) {
				(void)hipEventRecord(a);
				if (nb == 1) hipLaunchKernelGGL(k_walk<1>, dim3(blocks), dim3(64), 0, 0, tab, n_blocks, steps, sink);
				else hipLaunchKernelGGL(k_walk<2>, dim3(blocks), dim3(64), 0, 0, tab, n_blocks, steps, sink);
				(void)hipEventRecord(b); (void)hipEventSynchronize(b);
				float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
				if (ms < best) best = ms;
			}
			const double nblk = (double)blocks * 64 * steps * nb;
			printf("table %5.2f GiB, %d block(s) in flight per lane: %.2f ms, %.1f G blocks/s, %.0f GB/s (64 B per block), %.2f us per step\n",
			       g, nb, best, nblk / (best * 1e6), nblk * 64 / (best * 1e6), best * 1e3 / steps);
		}
	return 0;
}
