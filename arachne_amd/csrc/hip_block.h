// hip_block.h -- workgroup-cooperative primitives for kernels that give one work item (a barcode) to a whole 1024-lane
// workgroup: strided parallel-for with barrier, block scan, bitonic key/value sort, arg-max.  The device logic written
// against this handle (dev_rfa.h) keeps its control flow uniform over the workgroup.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace arx {

#ifndef ARX_BLOCK_LANES
#define ARX_BLOCK_LANES 1024 // 6.4 ms (256) -> 4.3 (512) -> 3.8 (1024) per 350 k-pair batch: a barcode is one workgroup and the kernel is latency-bound
#endif
constexpr int BLOCK_LANES = ARX_BLOCK_LANES, SORT_LDS = 4096;
// A barcode of TELLseq size (tens of pairs) does not need 1024 lanes and 60 KB of LDS: small barcodes run in workgroups of
// SMALL_LANES lanes with a SMALL_SORT-entry sort buffer (15 KB: ten workgroups per CU instead of two).
constexpr int SMALL_LANES = 256, SMALL_SORT = 1024;

template <int BLOCK_LANES, int SORT_LDS> struct HipBlockT {
	int tid;
	uint64_t *l64; // BLOCK_LANES words of LDS
	int32_t *l32;  // BLOCK_LANES + 1 words of LDS

	template <class F> __device__ __forceinline__ void pfor(int n, F f)
	{
		for (int i = tid; i < n; i += BLOCK_LANES) f(i);
		__syncthreads();
	}
	template <class F> __device__ __forceinline__ void single(F f)
	{
		if (tid == 0) f();
		__syncthreads();
	}
	// out[i] = sum of in[0..i), i in [0, n]; returns the total.  Every lane scans one contiguous chunk.
	__device__ int exclusive_scan(const int32_t *in, int32_t *out, int n)
	{
		const int per = (n + BLOCK_LANES - 1) / BLOCK_LANES, b = tid * per, e = b + per < n ? b + per : n;
		int sum = 0;
		for (int i = b; i < e; ++i) sum += in[i];
		l32[tid] = sum;
		__syncthreads();
		if (tid < 64) { // one wave scans the partial sums, BLOCK_LANES / 64 per lane
			constexpr int PER = BLOCK_LANES / 64;
			int a[PER], t = 0;
#pragma unroll
			for (int x = 0; x < PER; ++x) { a[x] = l32[PER * tid + x]; t += a[x]; }
			int incl = t;
			for (int d = 1; d < 64; d <<= 1) { int o = __shfl_up(incl, d, 64); if (tid >= d) incl += o; }
			int ex = incl - t;
#pragma unroll
			for (int x = 0; x < PER; ++x) { l32[PER * tid + x] = ex; ex += a[x]; }
			if (tid == 63) l32[BLOCK_LANES] = incl;
		}
		__syncthreads();
		int run = l32[tid];
		const int total = l32[BLOCK_LANES];
		for (int i = b; i < e; ++i) { const int x = in[i]; out[i] = run; run += x; }
		if (tid == 0) out[n] = total;
		__syncthreads();
		return total;
	}
	// ascending bitonic sort of P = 2^k (key, value) pairs by (key, (uint32)value); up to SORT_LDS pairs are sorted in LDS
	uint64_t *sk; int32_t *sv; // SORT_LDS entries of LDS each
	__device__ void sort_kv(uint64_t *k, int32_t *v, int P)
	{
		if (P <= SORT_LDS) {
			for (int t = tid; t < P; t += BLOCK_LANES) { sk[t] = k[t]; sv[t] = v[t]; }
			__syncthreads();
			sort_kv_in(sk, sv, P);
			for (int t = tid; t < P; t += BLOCK_LANES) { k[t] = sk[t]; v[t] = sv[t]; }
			__syncthreads();
		} else sort_kv_in(k, v, P);
	}
	__device__ void sort_kv_in(uint64_t *k, int32_t *v, int P)
	{
		for (int size = 2; size <= P; size <<= 1)
			for (int stride = size >> 1; stride > 0; stride >>= 1) {
				for (int t = tid; t < (P >> 1); t += BLOCK_LANES) {
					const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1)), hi = lo | stride;
					const bool up = (lo & size) == 0;
					const uint64_t ka = k[lo], kb = k[hi];
					const uint32_t va = (uint32_t)v[lo], vb = (uint32_t)v[hi];
					const bool gt = ka > kb || (ka == kb && va > vb);
					if (gt == up) { k[lo] = kb; k[hi] = ka; v[lo] = (int32_t)vb; v[hi] = (int32_t)va; }
				}
				__syncthreads();
			}
	}
	// largest key over i in [0, n) (0 = none), ties to the smallest i; every lane gets the result
	template <class F> __device__ void argmax(int n, F keyf, uint64_t *key, int *idx)
	{
		uint64_t bk = 0; int bi = 0x7fffffff;
		for (int i = tid; i < n; i += BLOCK_LANES) { const uint64_t x = keyf(i); if (x > bk) { bk = x; bi = i; } } // i ascends: the first maximum stays
		for (int d = 32; d > 0; d >>= 1) {
			const uint64_t ok = (uint64_t)__shfl_xor((unsigned long long)bk, d, 64); const int oi = __shfl_xor(bi, d, 64);
			if (ok > bk || (ok == bk && oi < bi)) { bk = ok; bi = oi; }
		}
		if ((tid & 63) == 0) { l64[tid >> 6] = bk; l32[tid >> 6] = bi; }
		__syncthreads();
		bk = l64[0]; bi = l32[0];
		for (int w = 1; w < BLOCK_LANES / 64; ++w) { const uint64_t ok = l64[w]; const int oi = l32[w]; if (ok > bk || (ok == bk && oi < bi)) { bk = ok; bi = oi; } }
		__syncthreads(); // the LDS words are reused by the next call
		*key = bk; *idx = bi;
	}
};

using HipBlock = HipBlockT<BLOCK_LANES, SORT_LDS>;

// cls / want: only the items whose class byte equals `want` (cls == nullptr: all)
template <class F, int LANES, int SORT> __global__ __launch_bounds__(LANES) void k_block_items(F f, int n, const uint8_t *cls, int want)
{
	__shared__ uint64_t l64[LANES];
	__shared__ int32_t l32[LANES + 1];
	__shared__ uint64_t sk[SORT];
	__shared__ int32_t sv[SORT];
	HipBlockT<LANES, SORT> blk{(int)threadIdx.x, l64, l32, sk, sv};
	for (int b = blockIdx.x; b < n; b += gridDim.x) {
		if (cls && cls[b] != want) continue; // uniform over the workgroup
		f(b, blk);
		__syncthreads();
	}
}

} // namespace arx
