// hip_index_build.h -- the BWT and the sampled suffix array of `bwa index`, built in HBM (gfx950).
//
// What it replaces: bwt_pac2bwt (bwtindex.c:61-125: is_bwt below 50 Mbp, bwt_bwtgen's ropes above, :271), bwt_bwtupdate_core
// (bwtindex.c:151-173) and bwt_cal_sa (bwt.c:62-84) of the reference.  The BWT of a text is unique, so any suffix sorter gives
// the reference's bytes; the reference's run single-threaded for about an hour on GRCh38.  Here the 2 * l_pac suffixes of
// forward + reverse-complement text are sorted on the device with the 288 GB of HBM as workspace:
//
//   1. text: 2 bits per base, 32 bases per 64-bit word, first base in the top bits -- the 32-mer at any position is two words and a
//      funnel shift.
//   2. bucket: histogram of the leading 12-mers (2^24 buckets, 64-bit counters), exclusive scan, scatter of every suffix into its
//      bucket (one atomic per suffix) -> the suffix array is sorted by its first 12 bases and cut into chunks of whole buckets.
//   3. chunk sort: per chunk a radix sort (rocprim, 64-bit keys = the 32-mer, 64-bit values = the suffix) -> sorted by 32 bases;
//      rank[suffix] = 1 + index of the first suffix with the same 32-mer (row 0 belongs to the empty suffix, is.c:208-223).
//      A suffix closer than 32 bases to the end sees 'A' beyond the text; the ties that creates are resolved in step 4 because
//      a position beyond the text ranks below every suffix, the further the lower.
//   4. prefix doubling (Larsson-Sadakane) on the suffixes that still share their 32-mer with another: sort each group by
//      rank[suffix + h], h = 32, 64, ...; groups of one leave the work list.  On a genome with repeats of a few per cent
//      divergence that list is well under a per cent of the suffixes and is gone after a handful of rounds; the list is processed
//      in slices of whole groups so that neither its length nor the key width (group number | rank) limits the genome.
//   5. emit: BWT symbol of every row = text[SA[row] - 1] (the row of suffix 0 removed, its index = primary), packed 16 per word
//      and interleaved with the running counts every 128 symbols exactly as bwt_bwtupdate_core lays them out; SA of every 32nd
//      row.  Both stream to the files through pinned staging.
//   6. (ARX_INDEX_VERIFY, default on) every adjacent pair of the suffix array is compared base by base on the device and rank[] is
//      checked to be its inverse: a wrong order cannot reach the files.
//
// Memory: 16 bytes per suffix (SA + rank) + 24 bytes per chunk element + 40 per slice element; GRCh38 (6.2 G suffixes): ~125 GB.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <chrono>
#include <stdexcept>
#include <string>
#include <vector>

namespace arx {
namespace gpuidx {

#define ARX_IDX_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string("index build: ") + #x + ": " + hipGetErrorString(e_)); } while (0)

struct DevBuf {
	void *p = nullptr; size_t bytes = 0;
	DevBuf() {}
	explicit DevBuf(size_t b) { alloc(b); }
	DevBuf(const DevBuf &) = delete; DevBuf &operator=(const DevBuf &) = delete;
	void alloc(size_t b)
	{
		release(); bytes = b ? b : 8;
		if (hipMalloc(&p, bytes) != hipSuccess) {
			size_t fr = 0, to = 0; (void)hipMemGetInfo(&fr, &to);
			p = nullptr;
			throw std::runtime_error("index build: out of device memory (wanted " + std::to_string(bytes >> 20) + " MiB, " + std::to_string(fr >> 20) + " of " + std::to_string(to >> 20) + " MiB free)");
		}
	}
	void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
	~DevBuf() { release(); }
	template <class T> T *as() const { return (T *)p; }
};

typedef unsigned long long u64;
constexpr int BUCKET_BASES = 12;
constexpr u64 N_BUCKETS = 1ull << (2 * BUCKET_BASES);

// A dispatch holds fewer than 2^32 work-items (the AQL packet's grid size is 32-bit) and GRCh38 has 6.2 G suffixes: every kernel is a
// grid-stride loop over at most 2^30 lanes.
__device__ __forceinline__ u64 gid() { return (u64)blockIdx.x * blockDim.x + threadIdx.x; }
#define ARX_GRID_LOOP(i, n) for (u64 i = gid(), stride_ = (u64)gridDim.x * blockDim.x; i < (n); i += stride_)
inline dim3 grid_for(u64 n, int block = 256) { u64 b = (n + block - 1) / block; if (b > ((u64)1 << 22)) b = (u64)1 << 22; if (b < 1) b = 1; return dim3((unsigned)b); }

// 32 bases from position i; beyond the text the words are zero ('A')
__device__ __forceinline__ u64 kmer32(const u64 *T, u64 i)
{
	const u64 w = i >> 5; const int sh = (int)(i & 31) << 1;
	const u64 a = T[w];
	if (sh == 0) return a;
	return a << sh | T[w + 1] >> (64 - sh);
}
__device__ __forceinline__ int base_at(const u64 *T, u64 i) { return (int)(T[i >> 5] >> (62 - ((i & 31) << 1))) & 3; }

// text word j from the forward strand's .pac bytes (4 bases per byte, first base in the top bits, bntseq.c:225):
// T[i] = fwd[i] for i < l_pac, 3 - fwd[n - 1 - i] above (bntseq.c:299-305 via bwtindex.c:78-85)
__global__ void k_pack_text(const uint8_t *pac, u64 l_pac, u64 *T, u64 n_words)
{
	const u64 n = 2 * l_pac;
	ARX_GRID_LOOP(j, n_words) {
		u64 w = 0;
		for (int s = 0; s < 32; ++s) {
			const u64 i = 32 * j + s;
			int c = 0;
			if (i < l_pac) c = pac[i >> 2] >> ((~i & 3) << 1) & 3;
			else if (i < n) { const u64 m = n - 1 - i; c = 3 - (pac[m >> 2] >> ((~m & 3) << 1) & 3); }
			w |= (u64)c << (62 - 2 * s);
		}
		T[j] = w;
	}
}

__global__ void k_hist(const u64 *T, u64 n, u64 *hist)
{
	ARX_GRID_LOOP(i, n) atomicAdd(&hist[kmer32(T, i) >> (64 - 2 * BUCKET_BASES)], 1ull);
}
__global__ void k_scatter(const u64 *T, u64 n, u64 *cursor, u64 *SA)
{
	ARX_GRID_LOOP(i, n) SA[atomicAdd(&cursor[kmer32(T, i) >> (64 - 2 * BUCKET_BASES)], 1ull)] = i;
}
__global__ void k_keys(const u64 *T, const u64 *sa, u64 cnt, u64 *keys)
{
	ARX_GRID_LOOP(t, cnt) keys[t] = kmer32(T, sa[t]);
}
// hv[t] = global index + 1 of t if it starts a group, else 0 (an inclusive max-scan turns it into the rank of every element)
__global__ void k_heads(const u64 *keys, u64 cnt, u64 base, u64 *hv)
{
	ARX_GRID_LOOP(t, cnt) hv[t] = (t == 0 || keys[t] != keys[t - 1]) ? base + t + 1 : 0;
}
// after the chunk sort: rank and SA go to their arrays; flag = the element shares its group with another
__global__ void k_chunk_apply(const u64 *sa_sorted, const u64 *rank, u64 cnt, u64 base, u64 *SA, u64 *ISA, u64 *flag)
{
	ARX_GRID_LOOP(t, cnt) {
		const u64 s = sa_sorted[t], r = rank[t];
		SA[base + t] = s; ISA[s] = r;
		const bool single = r == base + t + 1 && (t + 1 == cnt || rank[t + 1] == base + t + 2);
		flag[t] = single ? 0 : 1;
	}
}
__global__ void k_compact(const u64 *flag, const u64 *off, u64 cnt, u64 base, const u64 *sa_sorted, u64 *u_pos, u64 *u_sa)
{
	ARX_GRID_LOOP(t, cnt) if (flag[t]) { u_pos[off[t]] = base + t; u_sa[off[t]] = sa_sorted[t]; }
}

// ---- prefix doubling on the work list (u_pos: index into SA, increasing; u_sa: the suffix there)
__global__ void k_rank_of(const u64 *u_sa, u64 m, const u64 *ISA, u64 *r)
{
	ARX_GRID_LOOP(t, m) r[t] = ISA[u_sa[t]];
}
// bound[k] = first group start at or after k * S (atomic min over the group starts of [k * S, (k + 1) * S))
__global__ void k_slice_bounds(const u64 *r, u64 m, u64 S, u64 *bound)
{
	ARX_GRID_LOOP(t, m) if (t == 0 || r[t] != r[t - 1]) atomicMin(&bound[t / S], t);
}
__global__ void k_head_flags(const u64 *r, u64 cnt, u64 *hf)
{
	ARX_GRID_LOOP(t, cnt) hf[t] = (t == 0 || r[t] != r[t - 1]) ? 1 : 0;
}
// key = (group number within the slice) << kb | rank of the suffix h further on, shifted so that positions beyond the text
// (the further the smaller) stay non-negative: n + x for x >= 0 beyond ranks below rank(n) = 0 -> h - x
__global__ void k_comp_keys(const u64 *u_sa, const u64 *gnum, u64 cnt, const u64 *ISA, u64 n, u64 h, int kb, u64 *comp)
{
	ARX_GRID_LOOP(t, cnt) {
		const u64 p = u_sa[t] + h;
		const u64 v = p >= n ? h - (p - n) : h + ISA[p];
		comp[t] = (gnum[t] - 1) << kb | v;
	}
}
__global__ void k_heads2(const u64 *comp, const u64 *u_pos, u64 cnt, u64 *hv)
{
	ARX_GRID_LOOP(t, cnt) hv[t] = (t == 0 || comp[t] != comp[t - 1]) ? u_pos[t] + 1 : 0;
}
__global__ void k_round_apply(const u64 *sa_sorted, const u64 *rank, const u64 *u_pos, u64 cnt, u64 *SA, u64 *ISA, u64 *flag)
{
	ARX_GRID_LOOP(t, cnt) {
		const u64 s = sa_sorted[t], r = rank[t];
		ISA[s] = r; SA[u_pos[t]] = s;
		const bool single = r == u_pos[t] + 1 && (t + 1 == cnt || rank[t + 1] == u_pos[t + 1] + 1);
		flag[t] = single ? 0 : 1;
	}
}
__global__ void k_compact2(const u64 *flag, const u64 *off, u64 cnt, const u64 *u_pos, const u64 *sa_sorted, u64 *o_pos, u64 *o_sa)
{
	ARX_GRID_LOOP(t, cnt) if (flag[t]) { o_pos[off[t]] = u_pos[t]; o_sa[off[t]] = sa_sorted[t]; }
}

// ---- emit
// row r of the full matrix (0..n): suffix n for r = 0, SA[r - 1] otherwise; the BWT string skips the row of suffix 0 (= primary)
__global__ void k_bwt_words(const u64 *T, const u64 *SA, u64 n, u64 primary, u64 n_words, uint32_t *words, uint32_t *wcnt)
{
	ARX_GRID_LOOP(w, n_words) {
		uint32_t x = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
		for (int s = 0; s < 16; ++s) {
			const u64 k = 16 * w + s;
			if (k >= n) break;
			const u64 r = k + (k >= primary);
			const u64 suf = r == 0 ? n : SA[r - 1];
			const int b = base_at(T, suf - 1);
			x |= (uint32_t)b << (30 - 2 * s);
			c0 += b == 0; c1 += b == 1; c2 += b == 2; c3 += b == 3;
		}
		words[w] = x;
		wcnt[w] = c0 | c1 << 8 | c2 << 16 | c3 << 24;
	}
}
__global__ void k_block_counts(const uint32_t *wcnt, u64 n_words, u64 n_blocks, u64 *c0, u64 *c1, u64 *c2, u64 *c3)
{
	ARX_GRID_LOOP(b, n_blocks) {
		uint32_t s = 0;
		for (int j = 0; j < 8; ++j) { const u64 w = 8 * b + j; if (w < n_words) s += wcnt[w]; } // <= 128 per byte lane: no carry
		c0[b] = s & 0xff; c1[b] = (s >> 8) & 0xff; c2[b] = (s >> 16) & 0xff; c3[b] = s >> 24;
	}
}
// the file layout (bwtindex.c:151-173): before every 128 symbols the four running counts as u64, then the 8 words; after the last
// symbol the totals.  c0..c3 hold the exclusive prefix sums per block; tot[] the totals.
__device__ __forceinline__ void put_u64(uint32_t *o, u64 v) { o[0] = (uint32_t)v; o[1] = (uint32_t)(v >> 32); } // the totals may sit at an odd word
__global__ void k_interleave(const uint32_t *words, u64 n_words, u64 n_blocks, const u64 *c0, const u64 *c1, const u64 *c2, const u64 *c3,
                             const u64 *tot, uint32_t *out)
{
	ARX_GRID_LOOP(b, n_blocks + 1) {
		if (b == n_blocks) { uint32_t *o = out + 8 * n_blocks + n_words; for (int c = 0; c < 4; ++c) put_u64(o + 2 * c, tot[c]); continue; }
		uint32_t *o = out + 16 * b;
		put_u64(o, c0[b]); put_u64(o + 2, c1[b]); put_u64(o + 4, c2[b]); put_u64(o + 6, c3[b]);
		for (int j = 0; j < 8; ++j) { const u64 w = 8 * b + j; if (w < n_words) o[8 + j] = words[w]; }
	}
}
__global__ void k_sa_sample(const u64 *SA, u64 n_sa, u64 intv, u64 *out) // out[i - 1] = row i * intv, i = 1 .. n_sa - 1
{
	ARX_GRID_LOOP(k, n_sa - 1) { const u64 i = k + 1; out[i - 1] = SA[i * intv - 1]; }
}

// ---- verification: SA[j - 1] < SA[j] as suffixes of the text ($ at the end is the smallest symbol), and rank is SA's inverse
__device__ __forceinline__ bool verify_row(const u64 *T, const u64 *SA, const u64 *ISA, u64 n, u64 j) // true: row j is in order
{
	const u64 b = SA[j];
	if (b >= n || ISA[b] != j + 1) return false;
	if (j == 0) return true;
	const u64 a = SA[j - 1];
	if (a >= n) return true; // counted at its own row
	for (u64 d = 0;; d += 32) {
		const u64 ra = n - (a + d), rb = n - (b + d); // bases left in each suffix (> 0 on entry)
		const u64 ka = kmer32(T, a + d), kb = kmer32(T, b + d);
		const u64 lim = ra < rb ? ra : rb;
		if (lim >= 32) {
			if (ka != kb) return ka < kb;
			if (lim == 32) return ra < rb; // the one that ends here is the smaller
			continue;
		}
		const u64 mask = ~0ull << (64 - 2 * lim); // lim in 1..31 real bases in both
		if ((ka & mask) != (kb & mask)) return (ka & mask) < (kb & mask);
		return ra < rb; // equal up to the end of the shorter: a must be the shorter one
	}
}
__global__ void k_verify(const u64 *T, const u64 *SA, const u64 *ISA, u64 n, u64 *bad)
{
	ARX_GRID_LOOP(j, n) if (!verify_row(T, SA, ISA, n, j)) atomicAdd(bad, 1ull);
}

struct MaxOp { __host__ __device__ u64 operator()(const u64 &a, const u64 &b) const { return a > b ? a : b; } };

struct Timer {
	std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
	double lap() { auto t = std::chrono::steady_clock::now(); double s = std::chrono::duration<double>(t - t0).count(); t0 = t; return s; }
};

struct Scratch { // rocprim temporary storage, grown on demand
	DevBuf buf;
	void *get(size_t bytes) { if (bytes > buf.bytes) buf.alloc(bytes + (bytes >> 2)); return buf.p; }
};

inline void scan_max(Scratch &sc, u64 *data, u64 cnt, hipStream_t st)
{
	size_t tb = 0;
	ARX_IDX_CHECK(rocprim::inclusive_scan(nullptr, tb, data, data, (size_t)cnt, MaxOp(), st));
	void *tmp = sc.get(tb);
	ARX_IDX_CHECK(rocprim::inclusive_scan(tmp, tb, data, data, (size_t)cnt, MaxOp(), st));
}
inline void scan_incl_sum(Scratch &sc, u64 *data, u64 cnt, hipStream_t st)
{
	size_t tb = 0;
	ARX_IDX_CHECK(rocprim::inclusive_scan(nullptr, tb, data, data, (size_t)cnt, rocprim::plus<u64>(), st));
	void *tmp = sc.get(tb);
	ARX_IDX_CHECK(rocprim::inclusive_scan(tmp, tb, data, data, (size_t)cnt, rocprim::plus<u64>(), st));
}
inline void scan_excl_sum(Scratch &sc, const u64 *in, u64 *out, u64 cnt, hipStream_t st)
{
	size_t tb = 0;
	ARX_IDX_CHECK(rocprim::exclusive_scan(nullptr, tb, in, out, (u64)0, (size_t)cnt, rocprim::plus<u64>(), st));
	void *tmp = sc.get(tb);
	ARX_IDX_CHECK(rocprim::exclusive_scan(tmp, tb, in, out, (u64)0, (size_t)cnt, rocprim::plus<u64>(), st));
}
inline void sort_pairs(Scratch &sc, const u64 *kin, u64 *kout, const u64 *vin, u64 *vout, u64 cnt, int bits, hipStream_t st)
{
	size_t tb = 0;
	if (bits < 1) bits = 1;
	if (bits > 64) bits = 64;
	ARX_IDX_CHECK(rocprim::radix_sort_pairs(nullptr, tb, kin, kout, vin, vout, (size_t)cnt, 0u, (unsigned)bits, st));
	void *tmp = sc.get(tb);
	ARX_IDX_CHECK(rocprim::radix_sort_pairs(tmp, tb, kin, kout, vin, vout, (size_t)cnt, 0u, (unsigned)bits, st));
}
inline u64 read_u64(const u64 *d, hipStream_t st)
{
	u64 v = 0;
	ARX_IDX_CHECK(hipMemcpyAsync(&v, d, 8, hipMemcpyDeviceToHost, st));
	ARX_IDX_CHECK(hipStreamSynchronize(st));
	return v;
}
inline int bits_for(u64 x) { int b = 0; while (b < 64 && (x >> b)) ++b; return b; } // smallest b with x < 2^b

// device -> file through two pinned buffers
inline void stream_to_file(FILE *o, const void *dev, size_t bytes, hipStream_t st)
{
	const size_t CH = (size_t)64 << 20;
	void *pin[2] = {nullptr, nullptr};
	ARX_IDX_CHECK(hipHostMalloc(&pin[0], CH, hipHostMallocDefault));
	ARX_IDX_CHECK(hipHostMalloc(&pin[1], CH, hipHostMallocDefault));
	hipEvent_t ev[2];
	ARX_IDX_CHECK(hipEventCreate(&ev[0])); ARX_IDX_CHECK(hipEventCreate(&ev[1]));
	size_t done = 0, issued = 0; int k = 0;
	size_t len[2] = {0, 0};
	bool ok = true;
	if (bytes) { len[0] = bytes < CH ? bytes : CH; (void)hipMemcpyAsync(pin[0], dev, len[0], hipMemcpyDeviceToHost, st); (void)hipEventRecord(ev[0], st); issued = len[0]; }
	while (done < bytes) {
		const int nx = k ^ 1;
		if (issued < bytes) { // next chunk flies while this one is written
			len[nx] = bytes - issued < CH ? bytes - issued : CH;
			(void)hipMemcpyAsync(pin[nx], (const char *)dev + issued, len[nx], hipMemcpyDeviceToHost, st); (void)hipEventRecord(ev[nx], st);
			issued += len[nx];
		}
		if (hipEventSynchronize(ev[k]) != hipSuccess) { ok = false; break; }
		if (fwrite(pin[k], 1, len[k], o) != len[k]) { ok = false; break; }
		done += len[k]; k = nx;
	}
	(void)hipStreamSynchronize(st);
	(void)hipEventDestroy(ev[0]); (void)hipEventDestroy(ev[1]);
	(void)hipHostFree(pin[0]); (void)hipHostFree(pin[1]);
	if (!ok) throw std::runtime_error("index build: writing the index files failed");
}

struct DeviceBuildStats { double s_text = 0, s_bucket = 0, s_chunks = 0, s_rounds = 0, s_verify = 0, s_emit = 0; int n_chunks = 0, n_rounds = 0; u64 unresolved0 = 0; };

// pac: forward strand (l_pac bases, 4 per byte); cnt_fwd[c]: occurrences of base c on the forward strand.
// Writes <prefix>.bwt and <prefix>.sa.  Returns "" or an error message.
inline std::string build_bwt_sa_device(const uint8_t *pac, size_t pac_bytes, int64_t l_pac_, const uint64_t cnt_fwd[4], const std::string &prefix,
                                       int device = -1, DeviceBuildStats *stats = nullptr)
{
	const u64 l_pac = (u64)l_pac_, n = 2 * l_pac;
	const bool verbose = getenv("ARX_INDEX_VERBOSE") != nullptr;
	const bool verify = !(getenv("ARX_INDEX_VERIFY") && atoi(getenv("ARX_INDEX_VERIFY")) == 0);
	u64 chunk_cap = getenv("ARX_INDEX_CHUNK") ? strtoull(getenv("ARX_INDEX_CHUNK"), nullptr, 10) : (u64)512 << 20;
	u64 slice_cap = getenv("ARX_INDEX_SLICE") ? strtoull(getenv("ARX_INDEX_SLICE"), nullptr, 10) : (u64)256 << 20;
	if (chunk_cap < 1) chunk_cap = 1;
	if (slice_cap < 1) slice_cap = 1;
	DeviceBuildStats st_;
	DeviceBuildStats &S = stats ? *stats : st_;
	try {
		int ndev = 0;
		if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return "no HIP device visible";
		if (device >= 0) ARX_IDX_CHECK(hipSetDevice(device));
		hipStream_t st = 0; // the null stream: this is a stand-alone tool step, not part of a batch
		Timer tm;
		Scratch sc;
		// 1. text
		const u64 n_tw = (n + 31) / 32 + 2;
		DevBuf dT(n_tw * 8);
		{
			DevBuf dpac(pac_bytes + 8);
			ARX_IDX_CHECK(hipMemcpy(dpac.p, pac, pac_bytes, hipMemcpyHostToDevice));
			hipLaunchKernelGGL(k_pack_text, grid_for(n_tw), dim3(256), 0, st, dpac.as<uint8_t>(), l_pac, dT.as<u64>(), n_tw);
			ARX_IDX_CHECK(hipGetLastError());
			ARX_IDX_CHECK(hipStreamSynchronize(st));
		}
		const u64 *T = dT.as<u64>();
		S.s_text = tm.lap();
		// 2. buckets
		DevBuf dSA(n * 8), dISA(n * 8);
		u64 *SA = dSA.as<u64>(), *ISA = dISA.as<u64>();
		std::vector<u64> bstart(N_BUCKETS + 1);
		{
			DevBuf dh(N_BUCKETS * 8), dc(N_BUCKETS * 8);
			ARX_IDX_CHECK(hipMemsetAsync(dh.p, 0, N_BUCKETS * 8, st));
			hipLaunchKernelGGL(k_hist, grid_for(n), dim3(256), 0, st, T, n, dh.as<u64>());
			ARX_IDX_CHECK(hipGetLastError());
			scan_excl_sum(sc, dh.as<u64>(), dc.as<u64>(), N_BUCKETS, st);
			ARX_IDX_CHECK(hipMemcpyAsync(bstart.data(), dc.p, N_BUCKETS * 8, hipMemcpyDeviceToHost, st));
			ARX_IDX_CHECK(hipStreamSynchronize(st));
			bstart[N_BUCKETS] = n;
			hipLaunchKernelGGL(k_scatter, grid_for(n), dim3(256), 0, st, T, n, dc.as<u64>(), SA);
			ARX_IDX_CHECK(hipGetLastError());
			ARX_IDX_CHECK(hipStreamSynchronize(st));
		}
		S.s_bucket = tm.lap();
		// 3. chunks of whole buckets
		struct Seg { DevBuf pos, sa; u64 m = 0; };
		std::vector<Seg *> segs;
		struct SegGuard { std::vector<Seg *> &v; ~SegGuard() { for (Seg *s : v) delete s; } } seg_guard{segs};
		u64 m_total = 0;
		{
			u64 biggest = 0;
			for (u64 b = 0; b < N_BUCKETS; ++b) { const u64 c = bstart[b + 1] - bstart[b]; if (c > biggest) biggest = c; }
			const u64 cap = biggest > chunk_cap ? biggest : chunk_cap;
			DevBuf dk1(cap * 8), dk2(cap * 8), dv2(cap * 8);
			u64 b0 = 0;
			while (b0 < N_BUCKETS) {
				u64 b1 = b0 + 1;
				while (b1 < N_BUCKETS && bstart[b1 + 1] - bstart[b0] <= cap) ++b1;
				const u64 base = bstart[b0], cnt = bstart[b1] - base;
				b0 = b1;
				if (cnt == 0) continue;
				++S.n_chunks;
				u64 *k1 = dk1.as<u64>(), *k2 = dk2.as<u64>(), *v2 = dv2.as<u64>();
				hipLaunchKernelGGL(k_keys, grid_for(cnt), dim3(256), 0, st, T, SA + base, cnt, k1);
				sort_pairs(sc, k1, k2, SA + base, v2, cnt, 64, st);
				hipLaunchKernelGGL(k_heads, grid_for(cnt), dim3(256), 0, st, k2, cnt, base, k1);
				scan_max(sc, k1, cnt, st);                          // k1 = rank
				hipLaunchKernelGGL(k_chunk_apply, grid_for(cnt), dim3(256), 0, st, v2, k1, cnt, base, SA, ISA, k2); // k2 = flag
				ARX_IDX_CHECK(hipGetLastError());
				const u64 last_flag = read_u64(k2 + cnt - 1, st);
				scan_excl_sum(sc, k2, k1, cnt, st);                 // k1 = offsets
				const u64 m = read_u64(k1 + cnt - 1, st) + last_flag;
				if (m) {
					Seg *sg = new Seg(); segs.push_back(sg);
					sg->m = m; sg->pos.alloc(m * 8); sg->sa.alloc(m * 8);
					hipLaunchKernelGGL(k_compact, grid_for(cnt), dim3(256), 0, st, k2, k1, cnt, base, v2, sg->pos.as<u64>(), sg->sa.as<u64>());
					ARX_IDX_CHECK(hipGetLastError());
					m_total += m;
				}
				ARX_IDX_CHECK(hipStreamSynchronize(st));
			}
		}
		S.s_chunks = tm.lap(); S.unresolved0 = m_total;
		if (verbose) fprintf(stderr, "[arx index] n=%llu text %.2fs buckets %.2fs %d chunk sorts %.2fs, %llu suffixes (%.3f%%) share their 32-mer\n", n, S.s_text, S.s_bucket, S.n_chunks, S.s_chunks, m_total, 100.0 * m_total / (double)n);
		// 4. prefix doubling on the work list
		{
			u64 m = m_total;
			DevBuf upos[2], usa[2];
			int cur = 0;
			if (m) {
				upos[0].alloc(m * 8); usa[0].alloc(m * 8);
				u64 o = 0;
				for (Seg *sg : segs) {
					ARX_IDX_CHECK(hipMemcpyAsync(upos[0].as<u64>() + o, sg->pos.p, sg->m * 8, hipMemcpyDeviceToDevice, st));
					ARX_IDX_CHECK(hipMemcpyAsync(usa[0].as<u64>() + o, sg->sa.p, sg->m * 8, hipMemcpyDeviceToDevice, st));
					o += sg->m;
				}
				ARX_IDX_CHECK(hipStreamSynchronize(st));
			}
			for (Seg *sg : segs) delete sg;
			segs.clear();
			for (u64 h = 32; m > 0; h <<= 1) {
				if (h > 2 * n + 64) return "index build: prefix doubling did not converge (internal error)";
				++S.n_rounds;
				const int kb = bits_for(n + h);
				DevBuf dr(m * 8);
				u64 *r = dr.as<u64>();
				hipLaunchKernelGGL(k_rank_of, grid_for(m), dim3(256), 0, st, usa[cur].as<u64>(), m, ISA, r);
				// slices of whole groups, about slice_cap elements each (a group longer than that is one slice); the key of a slice
				// is (group number in the slice) << kb | rank: the group number must fit in 64 - kb bits
				u64 Sl = slice_cap;
				if (kb < 63 && ((u64)1 << (63 - kb)) < Sl) Sl = (u64)1 << (63 - kb);
				const u64 n_sl = (m + Sl - 1) / Sl;
				std::vector<u64> bound(n_sl + 1);
				{
					DevBuf db(n_sl * 8);
					ARX_IDX_CHECK(hipMemsetAsync(db.p, 0xff, n_sl * 8, st));
					hipLaunchKernelGGL(k_slice_bounds, grid_for(m), dim3(256), 0, st, r, m, Sl, db.as<u64>());
					ARX_IDX_CHECK(hipMemcpyAsync(bound.data(), db.p, n_sl * 8, hipMemcpyDeviceToHost, st));
					ARX_IDX_CHECK(hipStreamSynchronize(st));
				}
				std::vector<u64> cuts;
				for (u64 k = 0; k < n_sl; ++k) if (bound[k] != ~0ull) cuts.push_back(bound[k]);
				cuts.push_back(m); // cuts[0] == 0: element 0 starts a group
				u64 widest = 0;
				for (size_t k = 0; k + 1 < cuts.size(); ++k) if (cuts[k + 1] - cuts[k] > widest) widest = cuts[k + 1] - cuts[k];
				if (kb + bits_for(widest) > 64) return "index build: a group of equal suffixes is too long for the 64-bit sort key";
				const int nx = cur ^ 1;
				upos[nx].alloc(m * 8); usa[nx].alloc(m * 8);
				DevBuf da(widest * 8), dbb(widest * 8), dc(widest * 8), dd(widest * 8);
				u64 m_next = 0;
				for (size_t k = 0; k + 1 < cuts.size(); ++k) {
					const u64 a = cuts[k], cnt = cuts[k + 1] - a;
					u64 *A = da.as<u64>(), *B = dbb.as<u64>(), *Cc = dc.as<u64>(), *D = dd.as<u64>();
					const u64 *sa_in = usa[cur].as<u64>() + a, *pos_in = upos[cur].as<u64>() + a;
					hipLaunchKernelGGL(k_head_flags, grid_for(cnt), dim3(256), 0, st, r + a, cnt, A);
					scan_incl_sum(sc, A, cnt, st);                                                     // A = group number (1-based)
					hipLaunchKernelGGL(k_comp_keys, grid_for(cnt), dim3(256), 0, st, sa_in, A, cnt, ISA, n, h, kb, B); // B = keys
					sort_pairs(sc, B, Cc, sa_in, D, cnt, kb + bits_for(cnt), st);                      // Cc = sorted keys, D = sorted suffixes
					hipLaunchKernelGGL(k_heads2, grid_for(cnt), dim3(256), 0, st, Cc, pos_in, cnt, A);
					scan_max(sc, A, cnt, st);                                                          // A = new rank
					hipLaunchKernelGGL(k_round_apply, grid_for(cnt), dim3(256), 0, st, D, A, pos_in, cnt, SA, ISA, B); // B = flag
					ARX_IDX_CHECK(hipGetLastError());
					const u64 last_flag = read_u64(B + cnt - 1, st);
					scan_excl_sum(sc, B, Cc, cnt, st);                                                 // Cc = offsets
					const u64 keep = read_u64(Cc + cnt - 1, st) + last_flag;
					if (keep) {
						hipLaunchKernelGGL(k_compact2, grid_for(cnt), dim3(256), 0, st, B, Cc, cnt, pos_in, D, upos[nx].as<u64>() + m_next, usa[nx].as<u64>() + m_next);
						ARX_IDX_CHECK(hipGetLastError());
					}
					m_next += keep;
					ARX_IDX_CHECK(hipStreamSynchronize(st));
				}
				if (verbose) fprintf(stderr, "[arx index] round h=%llu: %llu -> %llu unresolved, %zu slice(s)\n", h, m, m_next, cuts.size() - 1);
				upos[cur].release(); usa[cur].release();
				cur = nx; m = m_next;
			}
		}
		S.s_rounds = tm.lap();
		// 6. verification
		if (verify) {
			DevBuf dbad(8);
			ARX_IDX_CHECK(hipMemsetAsync(dbad.p, 0, 8, st));
			hipLaunchKernelGGL(k_verify, grid_for(n), dim3(256), 0, st, T, SA, ISA, n, dbad.as<u64>());
			ARX_IDX_CHECK(hipGetLastError());
			const u64 bad = read_u64(dbad.as<u64>(), st);
			if (bad) return "index build: suffix array verification failed (" + std::to_string(bad) + " rows out of order)";
			S.s_verify = tm.lap();
		}
		// 5. emit
		const u64 primary = read_u64(ISA, st);
		dISA.release();
		u64 L2[5] = {0, 0, 0, 0, 0};
		for (int c = 0; c < 4; ++c) L2[c + 1] = L2[c] + cnt_fwd[c] + cnt_fwd[3 - c];
		{
			const u64 n_words = (n + 15) >> 4, n_blocks = (n + 127) / 128, out_words = n_words + (n_blocks + 1) * 8;
			DevBuf dwords(n_words * 4), dwcnt(n_words * 4), dout(out_words * 4 + 64);
			DevBuf c0(n_blocks * 8 + 8), c1(n_blocks * 8 + 8), c2(n_blocks * 8 + 8), c3(n_blocks * 8 + 8), dtot(32);
			hipLaunchKernelGGL(k_bwt_words, grid_for(n_words), dim3(256), 0, st, T, SA, n, primary, n_words, dwords.as<uint32_t>(), dwcnt.as<uint32_t>());
			hipLaunchKernelGGL(k_block_counts, grid_for(n_blocks), dim3(256), 0, st, dwcnt.as<uint32_t>(), n_words, n_blocks, c0.as<u64>(), c1.as<u64>(), c2.as<u64>(), c3.as<u64>());
			ARX_IDX_CHECK(hipGetLastError());
			u64 tot[4];
			DevBuf *cs[4] = {&c0, &c1, &c2, &c3};
			for (int c = 0; c < 4; ++c) {
				u64 *p = cs[c]->as<u64>();
				const u64 last = read_u64(p + n_blocks - 1, st);
				scan_excl_sum(sc, p, p, n_blocks, st);
				tot[c] = read_u64(p + n_blocks - 1, st) + last;
				if (tot[c] != cnt_fwd[c] + cnt_fwd[3 - c]) return "index build: BWT symbol counts do not match the text (internal error)";
			}
			ARX_IDX_CHECK(hipMemcpyAsync(dtot.p, tot, 32, hipMemcpyHostToDevice, st));
			ARX_IDX_CHECK(hipMemsetAsync(dout.p, 0, out_words * 4, st));
			hipLaunchKernelGGL(k_interleave, grid_for(n_blocks + 1), dim3(256), 0, st, dwords.as<uint32_t>(), n_words, n_blocks, c0.as<u64>(), c1.as<u64>(), c2.as<u64>(), c3.as<u64>(), dtot.as<u64>(), dout.as<uint32_t>());
			ARX_IDX_CHECK(hipGetLastError());
			ARX_IDX_CHECK(hipStreamSynchronize(st));
			FILE *o = fopen((prefix + ".bwt").c_str(), "wb");
			if (!o) return "cannot write " + prefix + ".bwt";
			fwrite(&primary, 8, 1, o); fwrite(L2 + 1, 8, 4, o);
			try { stream_to_file(o, dout.p, out_words * 4, st); } catch (...) { fclose(o); throw; }
			if (fclose(o) != 0) return "cannot write " + prefix + ".bwt";
		}
		{
			const u64 intv = 32, n_sa = (n + intv) / intv, seq_len = n;
			DevBuf ds((n_sa ? n_sa : 1) * 8);
			if (n_sa > 1) hipLaunchKernelGGL(k_sa_sample, grid_for(n_sa - 1), dim3(256), 0, st, SA, n_sa, intv, ds.as<u64>());
			ARX_IDX_CHECK(hipGetLastError());
			FILE *o = fopen((prefix + ".sa").c_str(), "wb");
			if (!o) return "cannot write " + prefix + ".sa";
			fwrite(&primary, 8, 1, o); fwrite(L2 + 1, 8, 4, o); fwrite(&intv, 8, 1, o); fwrite(&seq_len, 8, 1, o);
			try { stream_to_file(o, ds.p, (n_sa - 1) * 8, st); } catch (...) { fclose(o); throw; }
			if (fclose(o) != 0) return "cannot write " + prefix + ".sa";
		}
		S.s_emit = tm.lap();
		if (verbose) fprintf(stderr, "[arx index] %d doubling rounds %.2fs, verify %.2fs, emit %.2fs\n", S.n_rounds, S.s_rounds, S.s_verify, S.s_emit);
	} catch (const std::exception &ex) {
		return ex.what();
	}
	return "";
}

} // namespace gpuidx
} // namespace arx
