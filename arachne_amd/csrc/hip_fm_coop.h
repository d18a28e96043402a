// hip_fm_coop.h -- gfx950-only FM-index kernels.
//
// k_locate_dyn: sampled-SA locate (bwt_sa, bwt.c:86-96).  The number of LF steps per occurrence is geometric (mean
// sa_intv - 1 = 31), so a static one-occurrence-per-lane mapping leaves most lanes of a wave idle while the longest walk
// finishes.  Here every lane is a worker that takes its next occurrence as soon as its walk ends: a wave reserves chunks
// of the occurrence array with ONE atomic per chunk and hands them out to its idle lanes by ballot rank, so each
// iteration of the loop is one LF step (= one random 64-byte Occ block) for all 64 lanes.
#pragma once
#include "arx_dev.h"
#include "dev_fm.h"

namespace arx {

constexpr int LOCATE_CHUNK = 1024;

static __global__ void __launch_bounds__(256) k_locate_dyn(IndexView ix, Seed *occ, int n, int32_t *counter)
{
	const uint64_t mask = (uint64_t)ix.sa_intv - 1;
	int g = -1;                    // occurrence this lane is walking, -1 = idle
	uint64_t k = 0, steps = 0;
	int pool_next = 0, pool_end = 0; // wave-uniform: the reserved chunk
	bool exhausted = false;          // wave-uniform
	for (;;) {
		const unsigned long long idle = __ballot(g < 0);
		if (idle) {
			if (pool_next == pool_end && !exhausted) {
				int base = 0;
				if (__lane_id() == 0) base = atomicAdd(counter, LOCATE_CHUNK);
				base = __shfl(base, 0);
				if (base >= n) { exhausted = true; pool_next = pool_end = n; }
				else { pool_next = base; pool_end = base + LOCATE_CHUNK < n ? base + LOCATE_CHUNK : n; }
			}
			const int avail = pool_end - pool_next;
			if (avail > 0) {
				const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0));
				if (g < 0 && rank < avail) { g = pool_next + rank; k = (uint64_t)occ[g].rbeg; steps = 0; }
				const int need = __builtin_popcountll(idle);
				pool_next += need < avail ? need : avail;
			} else if (exhausted && idle == ~0ull) break; // nothing left to hand out and nobody is walking (every wave is 64 lanes wide here)
		}
		if (g >= 0) {
			if (k & mask) { k = lf_step(ix, k); ++steps; }
			else { occ[g].rbeg = (int64_t)(steps + ix.sa[k / (uint64_t)ix.sa_intv]); g = -1; }
		}
	}
}

// Persistent-lane kernels of the seeding passes (mem_collect_intv).  A read's SMEM search is a serial chain of several
// hundred bidirectional extensions whose count varies a lot between reads (repeats), so a static one-item-per-lane mapping
// makes every wavefront wait for its slowest item.  Here a lane takes its next item (a read, or a forward / backward task,
// dev_fm.h) as soon as the previous one is finished: a wave reserves chunks of the item range with ONE atomic and deals them
// out to its idle lanes by ballot rank.  Every iteration of the loop is one extension for all busy lanes: the lane program
// runs each lane's bookkeeping up to its next extension, the lanes reconverge on extend1() (two random 64-byte Occ blocks).
// The item's read is staged in LDS (4-bit codes, row stride 33 words so that the 64 lanes hit different banks): a search
// reads one base per extension, which would otherwise be a dependent HBM access in front of the Occ block loads.
// What happens between two runs of extensions (exporting a finished forward list, taking the next start or item) is rare per
// lane but long, and a wavefront pays for a divergent path whenever ONE lane is in it; lanes therefore wait in front of it
// until `batch` of them do (or nobody can extend), and the wave runs that code once for all of them.
constexpr int SEED_ROW = 132; // bytes per lane for the longest read: 256 bases + pad
// The row is as long as the batch's longest read needs (an odd number of words): LDS must not be what limits the waves per SIMD.
inline int seed_row_bytes(int max_len) { int w = ((max_len + 1) / 2 + 3) / 4; if (!(w & 1)) ++w; return w * 4 < SEED_ROW ? w * 4 : SEED_ROW; }
#ifndef ARX_SEED_WPE
#define ARX_SEED_WPE 4 // waves per SIMD the seeding kernels are compiled for (register budget 512 / WPE)
#endif
#ifndef ARX_SEED_FWD_WPE
#define ARX_SEED_FWD_WPE 4 // ... and the two forward kernels (5 and 6 were measured: the spills cost more than the wavefronts bring, profiles/r03/README.md)
#endif

// The reads of a batch as the seeding kernels want them: 4-bit codes, two per byte (low nibble first), one fixed-stride row of `row`
// bytes per read.  Packed once per batch (k_pack_reads); a lane that takes an item then copies ITS row into its LDS row with
// independent dword loads -- two round trips to memory for the whole wave.  (Round 1 staged the rows cooperatively from the byte-per-base
// array, one taken item after the other with four dependent loads each: ~40 us per refill of a wave, which forced the refills to be rare
// -- 48 parked lanes -- and left 25-32 of 64 lanes extending on average.)
static __global__ void __launch_bounds__(256) k_pack_reads(const uint8_t *bases, const int32_t *base_off, const int32_t *lens, int n_reads, int row_words, uint32_t *qn)
{
	const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	const int r = (int)(g / row_words), wd = (int)(g % row_words);
	if (r >= n_reads) return;
	const int len = lens[r];
	const uint8_t *b = bases + base_off[r];
	uint32_t x = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const int i = 2 * (4 * wd + k);
		const uint32_t lo = i < len && len <= MAX_READ_LEN ? b[i] : 4, hi = i + 1 < len && len <= MAX_READ_LEN ? b[i + 1] : 4;
		x |= (lo | hi << 4) << (8 * k);
	}
	qn[(size_t)r * row_words + wd] = x;
}

struct ItemFeeder {
	int pool_next = 0, pool_end = 0; bool exhausted = false; // wave-uniform
	// item < 0 marks an idle lane.  tasks != nullptr: items are task ids (t0 + item), the read to stage is the task's.
	// Returns false when nothing is left and the whole wave is idle.
	__device__ bool deal(int &item, bool &took, const SeedTask *tasks, int t0, const uint32_t *qn, int n, int32_t *counter, int chunk, uint8_t *q_lds, int row, int read0 = 0)
	{
		const int lane = threadIdx.x;
		took = false;
		const unsigned long long idle = __ballot(item < 0);
		if (!idle) return true;
		if (pool_next == pool_end && !exhausted) {
			int base = 0;
			if (lane == 0) base = atomicAdd(counter, chunk);
			base = __shfl(base, 0);
			if (base >= n) { exhausted = true; pool_next = pool_end = n; }
			else { pool_next = base; pool_end = base + chunk < n ? base + chunk : n; }
		}
		const int avail = pool_end - pool_next;
		if (avail <= 0) return !(exhausted && idle == ~0ull); // every wave is 64 lanes wide here
		const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0));
		took = item < 0 && rank < avail;
		if (took) item = pool_next + rank;
		const int need = __builtin_popcountll(idle);
		pool_next += need < avail ? need : avail;
		if (took) { // this lane's read into this lane's LDS row (row: an odd number of words, so the lanes hit different banks)
			const int rs = tasks ? tasks[t0 + item].read : read0 + item;
			const int rw = row >> 2;
			const uint32_t *src = qn + (size_t)rs * rw;
			uint32_t *dst = (uint32_t *)(q_lds + lane * row);
			for (int w0 = 0; w0 < rw; w0 += 8) {
				uint32_t v[8];
#pragma unroll
				for (int u = 0; u < 8; ++u) v[u] = src[w0 + u < rw ? w0 + u : rw - 1];
#pragma unroll
				for (int u = 0; u < 8; ++u) if (w0 + u < rw) dst[w0 + u] = v[u];
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		return true;
	}
};

struct SeedKArgs { // shared by the three kernels
	IndexView ix; const uint8_t *bases; const int32_t *base_off, *lens; SeedPools P; Biv *scratch; int list_cap; int32_t *first1; int t0;
	int32_t *heavy; int32_t *n_heavy; int budget; // backward sweeps that exceed `budget` extensions are queued here for k_seed_bwd_wave
	int row;                                       // bytes of LDS per lane for its read (seed_row_bytes)
	const uint32_t *qn;                            // the batch's reads as nibble rows (k_pack_reads), row / 4 words each
	int read0;                                     // first pass: item i is read read0 + i
	unsigned long long *dbg;                       // (diagnostics, ARX_SEED_STATS) [0] wave-iterations, [1] lane-extensions, [2] slow-path entries, [3] waves
};

// Lane programs: begin(item) after the read is staged (false: nothing to do), advance(req, rb, rc, slow_ok) -> has a request /
// parked or done, consume(req, ok), done(), finish().  A program that needs pool entries (and a task id) parks with want() > 0;
// the driver serves all such lanes of the wave with ONE atomic per cursor (a per-lane atomic on the batch-wide cursors was the
// bottleneck of the forward kernels) and calls granted().
struct FwdProg1 { // first pass: the forward extensions of one read, start after start
	// A finished forward list becomes a backward task once it has a pool slice, and slices are handed out in one step for the wavefront.
	// The lane does not wait for that: the list stays where it is (`pending`, one of the lane's two list buffers) and the next extension
	// starts at once in the other buffer -- where the next start lies is known from the list itself (bwt.c:356).  Only a lane that finishes
	// a second list before the first one has its slice parks, and a few parked lanes are enough to trigger the hand-out (grant step of
	// persistent_lanes()).  (Parking after every extension until 48 lanes had parked left 30 of 64 lanes extending.)
	static constexpr bool ALLOCATES = true, NEW_TASK = true, AUX = true;
	const SeedKArgs &A; Biv *list; QNibbles q; FwdLane<QNibbles> ln; int r, len, x, head, last, cur, pend_n, pend_x, pend_buf; bool extending, awaiting, over;
	int pend_def; uint32_t pend_code; bool want_tab; // the pending list's owed prefix (FwdLane::n_def) and its k-mer; the lane has asked for a table entry
	__device__ FwdProg1(const SeedKArgs &a, Biv *l, QNibbles qq) : A(a), list(l), q(qq), r(-1), len(0), x(0), head(-1), last(-1), cur(0), pend_n(0), pend_x(0), pend_buf(0), extending(false), awaiting(false), over(true), pend_def(0), pend_code(0), want_tab(false) {}
	__device__ Biv *buf(int b) const { return list + b * (A.list_cap >> 1); }
	__device__ bool begin(int item)
	{
		r = A.read0 + item; len = A.lens[r]; x = 0; head = last = -1; extending = false; awaiting = false; over = false; pend_n = 0; cur = 0; pend_def = 0; want_tab = false;
		if (len > MAX_READ_LEN) { atomicOr(A.P.err, ERR_READ_TOO_LONG); len = 0; }
		if (len < OPT_MIN_SEED_LEN) { A.first1[r] = -1; over = true; return false; }
		return true;
	}
	__device__ void shelve() // the finished list of ln becomes the pending one; the next start is where its longest match ends
	{
		pend_n = ln.n; pend_x = x; pend_buf = cur; pend_def = ln.n_def; pend_code = ln.code; x = ln.ret(); cur ^= 1; extending = false;
	}
	__device__ bool advance(Biv *req, int *rb, int *rc, bool)
	{
		*rb = 0;
		while (!over && !awaiting) {
			if (extending) {
				if (ln.advance(req, rc)) return true;
				if (pend_n > 0) { awaiting = true; break; } // two finished lists: wait for the slice of the first
				shelve();
			}
			while (x < len && q.at(x) > 3) ++x;
			if (x >= len) { over = true; break; }
			uint64_t code = 0;
			extending = true;
			if (ln.start_jump(A.ix, len, q, x, 1, buf(cur), &code)) { want_tab = true; req->k = code; *rc = -1; return true; } // the interval of the first K bases: one table load (consume() below)
		}
		return false;
	}
	__device__ int want() const { return 3 * (pend_n + pend_def); } // an upper bound while a prefix is owed: the depths whose size does not change take no entry
	__device__ bool parked() const { return awaiting; }
	__device__ int export_off() const { return (int)(buf(pend_buf) - A.scratch); } // where the list to export lies, in entries from A.scratch
	__device__ int export_def() const { return pend_def; }
	__device__ uint32_t export_code() const { return pend_code; }
	__device__ int export_x() const { return pend_x; }
	__device__ void granted(int off, int t, int n_act) // n_act: entries the wavefront wrote (the lane's own plus the owed prefix)
	{
		const int n = n_act, n_res = pend_n + pend_def;
		pend_n = 0; pend_def = 0;
		if (t >= A.P.task_cap || (int64_t)off + 3 * n_res > A.P.pool_cap) {
			atomicOr(A.P.err, ERR_POOL_OVERFLOW);
			if (t < A.P.task_cap) { SeedTask e = SeedTask(); e.read = r; e.next = -1; A.P.tasks[t] = e; } // the id is taken: leave an empty task the later kernels skip
			over = true; awaiting = false;
			return;
		}
		SeedTask k = SeedTask(); k.read = r; k.x = pend_x; k.min_intv = 1; k.off = off; k.n = n; k.nm = 0; k.next = -1;
		A.P.tasks[t] = k; // (the list itself was copied to pool + off by the wavefront, persistent_lanes())
		if (last >= 0) A.P.tasks[last].next = t; else head = t;
		last = t;
		if (awaiting) { awaiting = false; shelve(); } // the parked second list moves up
	}
	__device__ void consume(const Biv &, const Biv &ok) { ln.consume(ok); }
	__device__ void consume_aux(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) // the answer to a request with *rc < 0: the k-mer's table entry, or what text mode asked for (dev_fm.h)
	{
		if (want_tab) { want_tab = false; ln.take_jump(A.ix, ktab_unpack((uint64_t)w1 << 32 | w0, (uint64_t)w3 << 32 | w2)); }
		else ln.consume_aux(A.ix, w0, w1, w2, w3);
	}
	__device__ bool done() const { return over && pend_n == 0; }
	__device__ void finish() { A.first1[r] = head; }
};

struct FwdProg2 { // re-seeding: the forward extension of one task
	// With the per-depth tables the extension starts at the interval of its first K bases like a first-pass one (FwdLane::start_jump): a
	// re-seeding walk (min_intv = occurrences of the SMEM + 1) ends a few bases beyond the table's depth, so the table takes 13 of its ~17
	// dependent steps; the list prefix it owes is expanded by the grant step as in the first pass.
	static constexpr bool ALLOCATES = true, NEW_TASK = false, AUX = true;
	const SeedKArgs &A; Biv *list; QNibbles q; FwdLane<QNibbles> ln; int t, x; bool awaiting, over, want_tab; uint32_t tab_code;
	__device__ FwdProg2(const SeedKArgs &a, Biv *l, QNibbles qq) : A(a), list(l), q(qq), t(-1), x(0), awaiting(false), over(true), want_tab(false), tab_code(0) {}
	__device__ bool begin(int item)
	{
		t = A.t0 + item;
		const SeedTask k = A.P.tasks[t];
		x = k.x;
		uint64_t code = 0;
		want_tab = ln.start_jump(A.ix, A.lens[k.read], q, k.x, k.min_intv, list, &code);
		tab_code = (uint32_t)code;
		over = false; awaiting = false;
		return true;
	}
	__device__ bool advance(Biv *req, int *rb, int *rc, bool slow_ok)
	{
		*rb = 0;
		if (over || awaiting) return false;
		if (want_tab) { req->k = tab_code; *rc = -1; return true; } // the interval of the first K bases: one table load (consume_aux)
		if (ln.advance(req, rc)) return true;
		(void)slow_ok;
		awaiting = true; // the list wants its pool slice: handed out in the wavefront's next grant step
		return false;
	}
	__device__ int want() const { return awaiting ? 3 * (ln.n + ln.n_def) : 0; } // (an upper bound while a prefix is owed, as in FwdProg1)
	__device__ bool parked() const { return awaiting; }
	__device__ int export_off() const { return (int)(list - A.scratch); }
	__device__ int export_def() const { return ln.n_def; }
	__device__ uint32_t export_code() const { return ln.code; }
	__device__ int export_x() const { return x; }
	__device__ void granted(int off, int, int n_act)
	{
		awaiting = false; over = true;
		if ((int64_t)off + 3 * (ln.n + ln.n_def) > A.P.pool_cap) { atomicOr(A.P.err, ERR_POOL_OVERFLOW); return; } // n stays 0: the task is skipped
		A.P.tasks[t].off = off; A.P.tasks[t].n = n_act; // the list itself (own entries + owed prefix) was written to pool + off by the wavefront, persistent_lanes()
	}
	__device__ void consume(const Biv &, const Biv &ok) { ln.consume(ok); }
	__device__ void consume_aux(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
	{
		want_tab = false;
		ln.take_jump(A.ix, ktab_unpack((uint64_t)w1 << 32 | w0, (uint64_t)w3 << 32 | w2));
	}
	__device__ bool done() const { return over; }
	__device__ void finish() {}
};

struct BwdProg { // the backward sweep of one task
	static constexpr bool ALLOCATES = false, NEW_TASK = false, AUX = false;
	const SeedKArgs &A; QNibbles q; BwdLane<QNibbles> ln; int t;
	__device__ BwdProg(const SeedKArgs &a, Biv *, QNibbles qq) : A(a), q(qq), t(-1) { ln.finished = true; }
	__device__ bool begin(int item)
	{
		t = A.t0 + item;
		const SeedTask k = A.P.tasks[t];
		if (k.n == 0) { ln.finished = true; return false; }
		if (k.x == 0) { // nothing lies before the read: the longest forward match is the SMEM (the sweep's c < 0 case at i = -1)
			A.P.pool[k.off + 2 * k.n] = A.P.pool[k.off];
			A.P.tasks[t].nm = 1;
			ln.finished = true;
			return false;
		}
		ln.start(q, k, A.P.pool);
		return true;
	}
	__device__ bool advance(Biv *req, int *rb, int *rc, bool) { *rb = 1; return ln.advance(req, rc, A.budget); }
	__device__ int want() const { return 0; }
	__device__ bool parked() const { return false; }
	__device__ int export_off() const { return 0; }
	__device__ int export_def() const { return 0; }
	__device__ uint32_t export_code() const { return 0; }
	__device__ int export_x() const { return 0; }
	__device__ void granted(int, int, int) {}
	__device__ void consume(const Biv &req, const Biv &ok) { ln.consume(req, ok); }
	__device__ void consume_aux(uint32_t, uint32_t, uint32_t, uint32_t) {}
	__device__ bool done() const { return ln.finished; }
	__device__ void finish()
	{
		SeedTask &k = A.P.tasks[t];
		k.nm = ln.nm;
		if (ln.handed) { // the rest of this sweep goes to a whole wavefront
			k.flip = ln.prev != A.P.pool + k.off; k.row = ln.i; k.n_prev = ln.n_prev; k.mls = ln.mem_last_start;
			A.heavy[atomicAdd(A.n_heavy, 1)] = t;
		}
	}
};

// The rest of a long backward sweep, one task per wavefront: the entries of a row are extended side by side (lane j takes
// prev[j]), and what the serial loop does with the results in order (bwt.c:336-345: the first interval that falls below
// min_intv before any survivor becomes an SMEM; a survivor is kept unless it has the size of the survivor before it) is
// done with ballots and shuffles.  A row then costs two trips to HBM however many entries it has.
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) { return (uint64_t)__shfl((int)(v >> 32), src) << 32 | (uint32_t)__shfl((int)v, src); }

static __global__ void __launch_bounds__(64) k_seed_bwd_wave(SeedKArgs A)
{
	__shared__ __attribute__((aligned(4))) uint8_t q_lds[SEED_ROW + 4];
	const int lane = threadIdx.x;
	const int n_heavy = *A.n_heavy;
	for (int h = blockIdx.x; h < n_heavy; h += gridDim.x) {
		const int t = A.heavy[h];
		const SeedTask k = A.P.tasks[t];
		__builtin_amdgcn_wave_barrier();
		if (lane < (A.row >> 2)) ((uint32_t *)q_lds)[lane] = A.qn[(size_t)k.read * (A.row >> 2) + lane]; // the task's read: one word per lane
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		const QNibbles q{q_lds};
		Biv *prev = A.P.pool + k.off + (k.flip ? k.n : 0), *curr = A.P.pool + k.off + (k.flip ? 0 : k.n), *mem = A.P.pool + k.off + 2 * k.n;
		int i = k.row, n_prev = k.n_prev, nm = k.nm, mls = k.mls;
		const uint64_t min_intv = (uint64_t)k.min_intv;
		for (;;) {
			if (i < -1) break;
			const int c = i < 0 ? -1 : (q.at(i) < 4 ? q.at(i) : -1);
			if (c < 0) {
				if (n_prev > 0 && (nm == 0 || i + 1 < mls)) { if (lane == 0) { Biv x = prev[0]; x.info |= (uint64_t)(i + 1) << 32; mem[nm] = x; } ++nm; mls = i + 1; }
				break;
			}
			int n_curr = 0;
			bool carry_valid = false;
			uint64_t carry_s = 0;
			for (int base = 0; base < n_prev; base += 64) {
				const int j = base + lane;
				const bool valid = j < n_prev;
				Biv req = Biv(), ok = Biv();
				if (valid) { req = prev[j]; ok = extend1(A.ix, req, 1, c); }
				const bool keep = valid && ok.s >= min_intv;
				const unsigned long long km = __ballot(keep), fm = __ballot(valid && !keep);
				if (n_curr == 0 && fm) { // the first interval that died, if no survivor precedes it in the row
					const int jf = __builtin_ctzll(fm);
					if ((km & ((1ull << jf) - 1)) == 0 && (nm == 0 || i + 1 < mls)) {
						if (lane == jf) { Biv x = req; x.info |= (uint64_t)(i + 1) << 32; mem[nm] = x; }
						++nm; mls = i + 1;
					}
				}
				const unsigned long long below = km & ((1ull << lane) - 1);
				const int p = below ? 63 - __builtin_clzll(below) : 0;
				const uint64_t ps = shfl_u64(ok.s, p);
				const uint64_t prev_s = below ? ps : carry_s;
				const bool push = keep && (!(below || carry_valid) || ok.s != prev_s);
				const unsigned long long pm = __ballot(push);
				if (push) { Biv x = ok; x.info = req.info; curr[n_curr + __builtin_popcountll(pm & ((1ull << lane) - 1))] = x; }
				n_curr += __builtin_popcountll(pm);
				if (km) { carry_s = shfl_u64(ok.s, 63 - __builtin_clzll(km)); carry_valid = true; }
			}
			if (n_curr == 0) break;
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // the next row reads what this one wrote
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			Biv *sw = curr; curr = prev; prev = sw; n_prev = n_curr;
			--i;
		}
		if (lane == 0) A.P.tasks[t].nm = nm;
	}
}

template <class Prog, bool BY_TASK>
__device__ __forceinline__ void persistent_lanes(const SeedKArgs &A, int n, int32_t *counter, int batch_grant, int chunk, uint8_t *q_lds)
{
	const int lane = threadIdx.x, batch = batch_grant & 0xff, grant = (batch_grant >> 8) > 0 ? batch_grant >> 8 : 64; // (hip_rt.h packs the two thresholds)
	Prog prog(A, A.scratch + (size_t)(blockIdx.x * 64 + lane) * A.list_cap, QNibbles{q_lds + lane * A.row});
	ItemFeeder feed;
	int item = -1; // what this lane is working on, -1 = idle
	Biv req = Biv();
	int rb = 0, rc = 0;
	bool have_req = false;
	unsigned long long n_it = 0, n_ext = 0, n_slow = 0;
	for (;;) {
		if (item >= 0 && !have_req) have_req = prog.advance(&req, &rb, &rc, false); // cheap part: the next request of a running extension
		const int waiting = __builtin_popcountll(__ballot(!have_req));
		if (A.dbg) { ++n_it; n_ext += 64 - waiting; }
		// two steps off the extension loop, each for many lanes at once: the GRANT step hands pool slices (and task ids) to the lanes with a
		// finished list -- cheap: two atomics and the copies -- as soon as `grant` lanes are parked for one; the REFILL step (finish reads,
		// deal new ones, stage their rows) waits for `batch` lanes without work
		const bool refill = waiting >= batch || waiting == 64;
		const bool grants = Prog::ALLOCATES && (refill || __builtin_popcountll(__ballot(item >= 0 && prog.parked())) >= grant);
		if (grants) { // pool slices (and task ids) for every lane that has a finished list: one atomic per cursor and wave
			const int amt = item >= 0 ? prog.want() : 0;
			const unsigned long long askers = __ballot(amt > 0);
			if (askers) {
				int incl = amt;
				for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
				const int total = __shfl(incl, 63);
				int base = 0, tbase = 0;
				if (lane == 0) {
					base = atomicAdd(A.P.cursors, total);
					if (Prog::NEW_TASK) tbase = atomicAdd(A.P.cursors + 1, __builtin_popcountll(askers));
				}
				base = __shfl(base, 0); tbase = __shfl(tbase, 0);
				const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(askers >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)askers, 0));
				// The finished forward lists go to their pool slices, longest first (bwt.c:322), copied by the whole wavefront: four lists per
				// step, one per 16-lane quarter (most lists have 16 entries or fewer), their loads in flight together.  A lane copying its own
				// list waits for a round trip per entry, ~18 of them, with the rest of the wavefront waiting for it.
				__shared__ int g_n[64], g_off[64], g_src[64], g_def[64], g_x[64], g_act[64];
				__shared__ uint32_t g_code[64];
				if (amt > 0) {
					const int off = base + incl - amt, ndef = prog.export_def();
					g_n[rank] = (int64_t)off + amt <= A.P.pool_cap ? amt / 3 - ndef : -1; // the lane's own entries; -1: granted() raises the error, nothing is copied
					g_off[rank] = off; g_src[rank] = prog.export_off();
					g_def[rank] = ndef; g_code[rank] = prog.export_code(); g_x[rank] = prog.export_x();
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				const int n_ask = __builtin_popcountll(askers), quarter = lane >> 4, el = lane & 15;
				for (int k0 = 0; k0 < n_ask; k0 += 4) {
					const int k = k0 + quarter;
					const bool live = k < n_ask && g_n[k] >= 0;
					const int n_a = live ? g_n[k] : 0, off_a = live ? g_off[k] : 0, ndef = live ? g_def[k] : 0;
					if (live) {
						const Biv *src = A.scratch + g_src[k];
						for (int e = el; e < n_a; e += 16) A.P.pool[off_a + e] = src[n_a - 1 - e];
					}
					// The list prefix the lane did not store (dev_fm.h FwdLane): depth d = el + 1 of this list's k-mer from the tables, all depths of the
					// four lists side by side; depth d takes an entry where the size changes from d to d + 1 (bwt.c:308-313), the entries follow the lane's
					// own in order of decreasing depth ("longest first", bwt.c:322)
					int n_pre = 0;
					if (__ballot(ndef > 0)) { // (wave-uniform)
						const int d = el + 1;
						Biv td = Biv();
						if (ndef > 0 && d <= ndef + 1) td = klv_load(A.ix, d, (uint64_t)(g_code[k] & (uint32_t)((1ull << (2 * d)) - 1ull)));
						const uint64_t s_next = shfl_u64(td.s, (lane & 48) | ((el + 1) & 15)); // the size at depth d + 1, from the lane beside
						const bool flag = ndef > 0 && d <= ndef && td.s != s_next;
						const unsigned m16 = (unsigned)(__ballot(flag) >> (lane & 48)) & 0xffffu;
						n_pre = __builtin_popcount(m16);
						if (flag) { td.info = (uint64_t)(g_x[k] + d); A.P.pool[off_a + n_a + __builtin_popcount(m16 >> (el + 1))] = td; }
					}
					if (live && el == 0) g_act[k] = n_a + n_pre;
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				__builtin_amdgcn_wave_barrier(); // (g_act is read below; the LDS words are rewritten by the next grant step)
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				if (amt > 0) prog.granted(base + incl - amt, tbase + rank, g_n[rank] >= 0 ? g_act[rank] : 0);
				__builtin_amdgcn_wave_barrier();
			}
			if (item >= 0 && !have_req) have_req = prog.advance(&req, &rb, &rc, true); // a lane whose parked list just got its slice goes on
		}
		if (refill) {
			++n_slow;
			if (item >= 0 && !have_req && prog.done()) { prog.finish(); item = -1; }
			bool took;
			if (!feed.deal(item, took, BY_TASK ? A.P.tasks : nullptr, A.t0, A.qn, n, counter, chunk, q_lds, A.row, A.read0)) break;
			if (took && !prog.begin(item)) item = -1; // nothing to do for this item; the lane asks again next time round
			if (item >= 0 && !have_req) have_req = prog.advance(&req, &rb, &rc, true); // an item that ends here is finished the next time round
		}
		if (Prog::AUX && (A.ix.klv || A.ix.isa40)) { // forward passes with k-mer tables and / or text mode: a lane may ask for 16 bytes from
			// somewhere (rc < 0: a table entry, a suffix-array entry, reference text; dev_fm.h aux_addr) where the others ask for an extension; that
			// load is issued for every lane (the first Occ block for those that want none: a cached line) ahead of the Occ loads, so that one
			// wait covers both kinds
			const bool is_aux = have_req && rc < 0;
			const uint32_t *ap = is_aux ? aux_addr(A.ix, rc, req.k) : A.ix.bwt;
			const uint32_t x0 = ap[0], x1 = ap[1], x2 = ap[2], x3 = ap[3];
			if (have_req) { if (is_aux) prog.consume_aux(x0, x1, x2, x3); else prog.consume(req, extend1(A.ix, req, rb, rc)); have_req = false; }
		} else if (have_req) { prog.consume(req, extend1(A.ix, req, rb, rc)); have_req = false; }
	}
	if (A.dbg && lane == 0) { atomicAdd(A.dbg, n_it); atomicAdd(A.dbg + 1, n_ext); atomicAdd(A.dbg + 2, n_slow); atomicAdd(A.dbg + 3, 1ull); }
}

static __global__ void __launch_bounds__(64, ARX_SEED_FWD_WPE) k_seed_fwd1(SeedKArgs A, int n, int32_t *counter, int batch, int chunk)
{
	extern __shared__ uint8_t q_lds[]; // 64 rows of A.row bytes
	persistent_lanes<FwdProg1, false>(A, n, counter, batch, chunk, q_lds);
}
static __global__ void __launch_bounds__(64, ARX_SEED_FWD_WPE) k_seed_fwd2(SeedKArgs A, int n, int32_t *counter, int batch, int chunk)
{
	extern __shared__ uint8_t q_lds[]; // 64 rows of A.row bytes
	persistent_lanes<FwdProg2, true>(A, n, counter, batch, chunk, q_lds);
}
static __global__ void __launch_bounds__(64, ARX_SEED_WPE) k_seed_bwd(SeedKArgs A, int n, int32_t *counter, int batch, int chunk)
{
	extern __shared__ uint8_t q_lds[]; // 64 rows of A.row bytes
	persistent_lanes<BwdProg, true>(A, n, counter, batch, chunk, q_lds);
}

// ---- the backward sweeps again, as a pipeline: ONE wait on memory per loop iteration, with everything any lane needs next in flight.
// In k_seed_bwd a wavefront that refills its lanes stops for five dependent round trips (chunk reservation, task -> read id, read row,
// task, first interval) while its extending lanes stand by; at GRCh38 size a round trip under load is several microseconds, a refill
// cost ~22 us, and making refills rare (48 parked lanes) left 25-32 of 64 lanes extending (measured, profiles/r02).  Here a lane being
// refilled walks through stages, one per iteration, its loads riding along with the Occ loads of the lanes that extend:
//   stage 0 idle -> takes the next task id from the wave's reservation (ballot rank)          -> stage 1
//   stage 1      -> loads its SeedTask                                                        -> stage 2
//   stage 2      -> its read row goes to LDS by LDS-direct loads (no registers), prev[0] loads -> stage 3
//   stage 3      running: BwdLane::advance / ext_issue | wait | ext_finish / consume
// Sweeps that exceed the budget are flagged (no atomic in the loop); k_collect_heavy lists them for k_seed_bwd_wave.
static __global__ void __launch_bounds__(64, ARX_SEED_WPE) k_seed_bwd2(SeedKArgs A, int n, int32_t *counter, int chunk, uint8_t *heavy_flag)
{
	extern __shared__ uint8_t q_lds[]; // (row / 4) words x 64 lanes, word-major
	const int lane = threadIdx.x, rw = A.row >> 2;
	const QNibblesT q{q_lds + lane * 4};
	BwdLane<QNibblesT> ln;
	ln.finished = true;
	int stage = 0, t = -1;
	SeedTask k = SeedTask();
	Biv req = Biv(), nxt = Biv();
	int rc = 0;
	bool have_req = false;
	int pool_next = 0, pool_end = 0; bool exhausted = false; // wave-uniform: the reserved chunk of items
	unsigned long long n_it = 0, n_ext = 0;
	for (;;) {
		// A. running lanes: the next request, or the sweep is over
		if (stage == 3 && !have_req) {
			have_req = ln.advance_nx(&req, &rc, A.budget, true, nxt);
			if (!have_req) {
				SeedTask &kt = A.P.tasks[t];
				kt.nm = ln.nm;
				if (ln.handed) { kt.flip = ln.prev != A.P.pool + k.off; kt.row = ln.i; kt.n_prev = ln.n_prev; kt.mls = ln.mem_last_start; heavy_flag[t - A.t0] = 1; }
				stage = 0;
			}
		}
		if (A.dbg) { ++n_it; n_ext += __builtin_popcountll(__ballot(have_req)); }
		// B. idle lanes take items
		const unsigned long long idle = __ballot(stage == 0);
		if (idle) {
			if (pool_next == pool_end && !exhausted) {
				int base = 0;
				if (lane == 0) base = atomicAdd(counter, chunk); // a dozen reservations per wave and launch: not worth hiding
				base = __shfl(base, 0);
				if (base >= n) { exhausted = true; pool_next = pool_end = n; }
				else { pool_next = base; pool_end = base + chunk < n ? base + chunk : n; }
			}
			const int avail = pool_end - pool_next;
			if (avail > 0) {
				const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0));
				if (stage == 0 && rank < avail) { t = A.t0 + pool_next + rank; stage = 1; }
				const int need = __builtin_popcountll(idle);
				pool_next += need < avail ? need : avail;
			} else if (exhausted && idle == ~0ull) break;
		}
		// C. everything this iteration needs from memory (the scheduling barriers keep every load above every use: left alone, the
		//    scheduler computes on the Occ blocks before it issues the other loads -- fewer registers, two round trips per iteration)
		__builtin_amdgcn_sched_barrier(0);
		ExtLoad L;
		ext_issue(A.ix, req, 1, have_req, L);
		nxt = A.P.pool[(stage == 3 && ln.has_next_entry()) ? (ln.prev - A.P.pool) + ln.j + 1 : 0]; // the sweep's next list entry rides along
		const SeedTask kt2 = A.P.tasks[stage == 1 ? t : A.t0];   // lanes in other stages load the first task / pool entry 0 and drop it
		const Biv p0 = A.P.pool[stage == 2 ? k.off : 0];
		if (__ballot(stage == 2)) {
			if (stage == 2) {
				const uint32_t *src = A.qn + (size_t)k.read * rw;
				for (int w = 0; w < rw; ++w)
					__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + w), (__attribute__((address_space(3))) void *)(q_lds + w * 256), 4, 0, 0);
			}
		}
		__builtin_amdgcn_sched_barrier(0);
		// D. (the first use below waits)  E. consume
		if (have_req) { ln.consume(req, ext_finish(A.ix, req, 1, rc, L)); have_req = false; }
		if (stage == 2) {
			if (k.x == 0) { // nothing lies before the read: the longest forward match is the SMEM (the sweep's c < 0 case at i = -1)
				A.P.pool[k.off + 2 * k.n] = p0;
				A.P.tasks[t].nm = 1;
				stage = 0;
			} else { ln.start_with(q, k, A.P.pool, p0); stage = 3; }
		} else if (stage == 1) { k = kt2; stage = k.n == 0 ? 0 : 2; }
	}
	if (A.dbg && lane == 0) { atomicAdd(A.dbg, n_it); atomicAdd(A.dbg + 1, n_ext); atomicAdd(A.dbg + 3, 1ull); }
}
// ---- the backward sweeps, row-parallel: one task per 16-lane group, the entries of a row side by side in the group's lanes.
// A sweep extends every interval of the current list by one base, keeps the survivors (dropping one that has the size of the survivor
// before it) and repeats with the next base to the left.  k_seed_bwd gives the sweep to ONE lane: ~100 dependent extensions per task,
// every list entry beyond the first read from and written to the batch-wide pool (measured at GRCh38 size: 0.9 32-byte stores per
// extension, 3.6 GB of writes next to 14 GB of reads per launch, scattered partial-line writes).  Here lane j of the group holds entry
// j in registers: a row is ONE trip to memory whatever its length, a task is as many trips as it has rows (5-10), and the lists never
// touch memory -- the pool is read once (the forward list, <= 16 entries) and written only with the SMEMs found.  The order-dependent
// rules of the serial loop (bwt.c:336-345: the first interval that falls below min_intv before any survivor becomes an SMEM; a survivor
// is kept unless it has the size of the survivor before it) are ballots within the group, as in k_seed_bwd_wave; survivors move to the
// group's low lanes through LDS.  Refills are pipelined as in k_seed_bwd2: a group that takes a task loads it in one iteration, its read
// row and list in the next, riding along with the Occ loads of the groups that extend; one wait on memory per iteration.
// Tasks whose forward list is longer than 16 go to k_seed_bwd_wave (64 lanes) from their first row.
// GL = lanes per group (16, 21, 32 or 64): tasks are binned by the length of their forward list (k_bin_tasks) and each bin runs with the
// smallest group that holds its rows; `list` / `n_list` = the bin's task ids and their number (device memory).  GL = 21 (three groups and an
// idle lane): at GRCh38 size a forward list has an entry for nearly every one of the ~16 depths it takes to get to one occurrence, so more
// than half of the first pass's lists are 17-20 entries long -- just too long for a quarter of a wavefront, and half a wavefront leaves most
// of its lanes empty from the first row on.
template <int GL, bool FIT32>
__device__ __forceinline__ void seed_bwd_g_body(const SeedKArgs &A, const int32_t *list, const int32_t *n_list, int32_t *counter, int chunk, uint8_t *heavy_flag)
{
	constexpr int NG = 64 / GL, NW = (33 + GL - 1) / GL; // groups per wave; row words a lane copies (a row has at most 33 words)
	const int n = *n_list;
	extern __shared__ uint8_t lds_g16[];                      // NG read rows (A.row bytes each), then 64 exchange slots of 32 bytes
	const int lane = threadIdx.x, g = lane / GL, gl = lane % GL, rw = A.row >> 2;
	uint8_t *q_row = lds_g16 + g * A.row;
	Biv *xch = (Biv *)(lds_g16 + ((NG * A.row + 31) & ~31)) + g * GL;
	const QNibbles q{q_row};
	const bool in_group = g < NG;                             // (GL = 21: lane 63 belongs to no group; it goes along and never takes a task)
	const unsigned long long gmask = GL == 64 ? ~0ull : in_group ? ((1ull << (GL & 63)) - 1) << (GL * g) : 0ull;
	const unsigned long long lowmask = gmask & ((1ull << lane) - 1); // the lanes of this group below this one
	int stage = 0, t = -1;                                    // group-uniform: 0 idle, 1 task id taken, 2 task loaded, 3 running
	SeedTask k = SeedTask();
	Biv ent = Biv();                                          // this lane's entry of the current row (valid: gl < n_prev)
	int i = 0, n_prev = 0, nm = 0, mls = 0, c = 0;            // group-uniform sweep state
	int pool_next = 0, pool_end = 0; bool exhausted = false;  // wave-uniform
	unsigned long long n_it = 0, n_ext = 0;
	for (;;) {
		// A. running groups: the base to extend by, or the sweep is over
		bool ext = false;
		// text mode (dev_fm.h bwd_text_tail): a row that is ONE interval with ONE occurrence is not walked here -- three dependent loads and a
		// comparison with the reference text finish it, one thread per such sweep after this kernel (KSeedBwdTail); the group is free for the next task
		const bool to_tail = A.ix.isa40 && stage == 3 && n_prev == 1 && k.min_intv == 1 && i >= -1 && (__ballot(gl == 0 && ent.s == 1) & gmask);
		if (to_tail) {
			if (gl == 0) { SeedTask &kt = A.P.tasks[t]; A.P.pool[k.off + k.n] = ent; kt.row = i; kt.nm = nm; kt.mls = mls; heavy_flag[t - A.t0] = 2; }
			stage = 0;
		}
		if (stage == 3) {
			bool over = i < -1;
			if (!over) {
				c = i < 0 ? -1 : (q.at(i) < 4 ? q.at(i) : -1);
				if (c < 0) { // nothing can be extended: the longest interval survives if it is not contained (bwt.c:326-331)
					if (n_prev > 0 && (nm == 0 || i + 1 < mls)) {
						if (gl == 0) { Biv x = ent; x.info |= (uint64_t)(i + 1) << 32; A.P.pool[k.off + 2 * k.n + nm] = x; }
						++nm; mls = i + 1;
					}
					over = true;
				}
			}
			if (over) { if (gl == 0) A.P.tasks[t].nm = nm; stage = 0; }
			else ext = gl < n_prev;
		}
		if (A.dbg) { ++n_it; n_ext += __builtin_popcountll(__ballot(ext)); }
		// B. idle groups take tasks
		const unsigned long long idle = __ballot(stage == 0 && gl == 0 && in_group);
		if (idle) {
			if (pool_next == pool_end && !exhausted) {
				int base = 0;
				if (lane == 0) base = atomicAdd(counter, chunk);
				base = __shfl(base, 0);
				if (base >= n) { exhausted = true; pool_next = pool_end = n; }
				else { pool_next = base; pool_end = base + chunk < n ? base + chunk : n; }
			}
			const int avail = pool_end - pool_next;
			if (avail > 0) {
				const int rank = __builtin_popcountll(idle & ((1ull << (GL * g)) - 1)); // idle groups below this one
				if (stage == 0 && in_group && rank < avail) { t = list[pool_next + rank]; stage = 1; } // (in_group: the lane that belongs to no group must not take the slot after the last idle group's)
				const int need = __builtin_popcountll(idle);
				pool_next += need < avail ? need : avail;
			} else if (exhausted && __ballot(stage != 0) == 0) break;
		}
		// C. everything this iteration needs from memory
		__builtin_amdgcn_sched_barrier(0);
		ExtLoad L;
		ext_issue(A.ix, ent, 1, ext, L);
		SeedTask kt2 = SeedTask();
		Biv e0 = Biv();
		uint32_t wv[NW];
#pragma unroll
		for (int u = 0; u < NW; ++u) wv[u] = 0;
		if (__ballot(stage == 1 || stage == 2)) { // (wave-uniform) some group is taking a task: its loads ride along with the Occ loads of the others
			kt2 = A.P.tasks[stage == 1 ? t : A.t0];
			e0 = A.P.pool[(stage == 2 && gl < k.n) ? k.off + gl : 0];       // the forward list, longest match first (bwt.c:322)
			const uint32_t *src = A.qn + (size_t)(stage == 2 ? k.read : 0) * rw;
#pragma unroll
			for (int u = 0; u < NW; ++u) wv[u] = src[gl + GL * u < rw ? gl + GL * u : 0];
		}
		__builtin_amdgcn_sched_barrier(0);
		// D. (the first use waits)  E. consume
		Biv ok = Biv();
		if (ext) ok = ext_finish<FIT32>(A.ix, ent, 1, c, L);
		const bool keep = ext && ok.s >= (uint64_t)k.min_intv;
		// ballots stay wave-wide; this group's part is cut out with masks that do not change (no shifts by the group's position)
		const unsigned long long km = __ballot(keep) & gmask, fm = __ballot(ext && !keep) & gmask;
		const unsigned long long below = km & lowmask;
		const int p = below ? 63 - __builtin_clzll(below) : 0; // lane of the nearest survivor below this one
		const uint64_t ps = shfl_u64(ok.s, p);
		const bool push = keep && (!below || ok.s != ps);
		const unsigned long long pm = __ballot(push) & gmask;
		if (stage == 3) {
			if (fm) { // the first interval that died, if no survivor precedes it in the row
				const int jf = __builtin_ctzll(fm);
				if ((km & ((1ull << jf) - 1)) == 0 && (nm == 0 || i + 1 < mls)) {
					if (lane == jf) { Biv x = ent; x.info |= (uint64_t)(i + 1) << 32; A.P.pool[k.off + 2 * k.n + nm] = x; }
					++nm; mls = i + 1;
				}
			}
			if (push) { Biv x = ok; x.info = ent.info; xch[__builtin_popcountll(pm & lowmask)] = x; }
		}
		if (stage == 2) { // the read row of the group and its first list
			uint32_t *dst = (uint32_t *)q_row;
#pragma unroll
			for (int u = 0; u < NW; ++u) if (gl + GL * u < rw) dst[gl + GL * u] = wv[u];
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		if (stage == 3) {
			const int n_curr = __builtin_popcountll(pm);
			if (n_curr == 0) { if (gl == 0) A.P.tasks[t].nm = nm; stage = 0; }
			else { if (gl < n_curr) ent = xch[gl]; n_prev = n_curr; --i; }
		} else if (stage == 2) {
			if (k.x == 0) { // nothing lies before the read: the longest forward match is the SMEM
				if (gl == 0) { A.P.pool[k.off + 2 * k.n] = e0; A.P.tasks[t].nm = 1; }
				stage = 0;
			} else { ent = e0; n_prev = k.n; i = k.x - 1; nm = 0; mls = 0; stage = 3; }
		} else if (stage == 1) {
			k = kt2;
			if (k.n == 0) stage = 0;
			else if (k.n > GL) { // too long for this group size (only the 64-lane bin meets such lists): k_seed_bwd_wave walks them in chunks of 64
				if (gl == 0) { SeedTask &kt = A.P.tasks[t]; kt.flip = 0; kt.row = k.x - 1; kt.n_prev = k.n; kt.mls = 0; heavy_flag[t - A.t0] = 1; }
				stage = 0;
			} else stage = 2;
		}
		__builtin_amdgcn_wave_barrier(); // the exchange slots are rewritten next iteration
	}
	if (A.dbg && lane == 0) { atomicAdd(A.dbg, n_it); atomicAdd(A.dbg + 1, n_ext); atomicAdd(A.dbg + 3, 1ull); }
}
// One launch for the three bins: every wavefront works through the 16-lane bin, then the 32-lane bin, then the 64-lane bin, moving on
// as soon as a bin has nothing left to hand out -- the end of one bin overlaps the start of the next instead of leaving the chip to the
// last few groups (three launches: 2.4 + 1.95 + 0.3 ms, each with its own tail).
// FIT32: every symbol occurs fewer than 2^32 times in the text (dev_fm.h occ_counts_fit32; the host checks the index header): the sizes of an
// extension's children in 32 bits, the 40-bit count for one symbol only
template <bool FIT32>
static __global__ void __launch_bounds__(64, ARX_SEED_WPE) k_seed_bwd_g(SeedKArgs A, const int32_t *bins, int n, int32_t *cnt, uint8_t *heavy_flag)
{
	seed_bwd_g_body<16, FIT32>(A, bins, cnt, cnt + 4, 32, heavy_flag);
	__builtin_amdgcn_wave_barrier();
	seed_bwd_g_body<21, FIT32>(A, bins + n, cnt + 1, cnt + 5, 24, heavy_flag);
	__builtin_amdgcn_wave_barrier();
	seed_bwd_g_body<32, FIT32>(A, bins + 2 * (size_t)n, cnt + 2, cnt + 6, 16, heavy_flag);
	__builtin_amdgcn_wave_barrier();
	seed_bwd_g_body<64, FIT32>(A, bins + 3 * (size_t)n, cnt + 3, cnt + 7, 2, heavy_flag);
}
// ---- the backward sweeps, entry-parallel (end of round 3; ARX_SEED_BWD2=3).  What bwt_smem1a's backward loop computes (bwt.c:322-348) is,
// for every interval of the forward list on its own, how far to the left it can be extended before it holds fewer than min_intv occurrences:
//  * an interval's fate does not depend on the others -- extending is a function of the interval and the base;
//  * a longer match's occurrences are among a shorter one's, so the intervals give up in list order (longest first) and an interval that
//    gives up finds every longer one gone: "no survivor precedes it" always holds, and it becomes an SMEM iff its start lies strictly left
//    of the last SMEM's (the "contained" test; intervals that give up in the same row leave the SMEM to the longest);
//  * a survivor that is dropped for having the size of the survivor before it (bwt.c:341) has that one's occurrences, hence its fate:
//    it would give up in the same row and lose the same test.
// So every list entry is an item of its own for one lane: no rows, no groups, no lanes idling beside a short list; an entry that is down to
// ONE occurrence (first pass) goes on by comparing text (dev_fm.h text_match_back), the way the forward extensions do.  A lane walks through
// stages: 0 idle -> 1 its item's descriptor -> 2 the list entry -> 3 extending | 4 suffix-array entry -> 5 text -> 6 the inverse's entry;
// one wait on memory per iteration, with every load any lane needs in flight.  KBwdEFinal then applies the start test per task, in list order.
// Results are those of the row-parallel kernel bit for bit.
struct BwdItem { uint32_t src, dst, read, xm; }; // pool index of the list entry, of its result (the task's second list), the read, x | min_intv << 16

static __global__ void __launch_bounds__(256) k_bwd_e_count(const SeedTask *tasks, int t0, int n, int32_t *cnt)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { const SeedTask k = tasks[t0 + i]; cnt[i] = k.x > 0 ? k.n : 0; } // x == 0: nothing lies before the read (k_bwd_e_expand emits the longest match)
}
static __global__ void __launch_bounds__(256) k_bwd_e_expand(SeedTask *tasks, Biv *pool, int t0, int n, const int32_t *off, BwdItem *items)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const SeedTask k = tasks[t0 + i];
	if (k.n == 0) return;
	if (k.x == 0) { pool[k.off + 2 * k.n] = pool[k.off]; tasks[t0 + i].nm = 1; return; }
	BwdItem it; it.read = (uint32_t)k.read; it.xm = (uint32_t)k.x | (uint32_t)k.min_intv << 16;
	for (int j = 0; j < k.n; ++j) { it.src = (uint32_t)(k.off + j); it.dst = (uint32_t)(k.off + k.n + j); items[off[i] + j] = it; }
}
static __global__ void __launch_bounds__(256) k_bwd_e_final(SeedTask *tasks, Biv *pool, int t0, int n)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const SeedTask k = tasks[t0 + i];
	if (k.n == 0 || k.x == 0) return;
	int nm = 0, mls = 0;
	for (int j = 0; j < k.n; ++j) { // longest match first (bwt.c:322): an interval is an SMEM iff it starts left of the last one found (bwt.c:329)
		const Biv r = pool[k.off + k.n + j];
		const int start = (int)(r.info >> 32);
		if (nm == 0 || start < mls) { pool[k.off + 2 * k.n + nm] = r; ++nm; mls = start; }
	}
	tasks[t0 + i].nm = nm;
}

static __global__ void __launch_bounds__(64, ARX_SEED_WPE) k_seed_bwd_e(SeedKArgs A, const BwdItem *items, int n, int32_t *counter, int chunk)
{
	const int lane = threadIdx.x, rw = A.row >> 2;
	const bool text = A.ix.sa40 && A.ix.isa40;
	int stage = 0, item = 0, i = 0, min_intv = 1;
	uint32_t d_src = 0, d_dst = 0, d_read = 0, d_xm = 0;
	Biv cur = Biv();                 // info: the end of the match (the start is added when the entry is through)
	uint64_t tpos = 0; bool moved = false;
	int pool_next = 0, pool_end = 0; bool exhausted = false; // wave-uniform: the reserved chunk of items
	unsigned long long n_it = 0, n_ext = 0;
	auto finish = [&]() { Biv r = cur; r.info = (uint64_t)(uint32_t)cur.info | (uint64_t)(i + 1) << 32; A.P.pool[d_dst] = r; stage = 0; };
	for (;;) {
		// A. an extending lane that has run off the read is through; one whose interval holds one occurrence goes on in the text
		if (stage == 3) {
			if (i < 0) finish();
			else if (text && min_intv == 1 && cur.s == 1) { stage = 4; tpos = cur.k; }
		}
		if (A.dbg) { ++n_it; n_ext += __builtin_popcountll(__ballot(stage >= 3)); }
		// B. idle lanes take items
		const unsigned long long idle = __ballot(stage == 0);
		if (idle) {
			if (pool_next == pool_end && !exhausted) {
				int base = 0;
				if (lane == 0) base = atomicAdd(counter, chunk);
				base = __shfl(base, 0);
				if (base >= n) { exhausted = true; pool_next = pool_end = n; }
				else { pool_next = base; pool_end = base + chunk < n ? base + chunk : n; }
			}
			const int avail = pool_end - pool_next;
			if (avail > 0) {
				const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0));
				if (stage == 0 && rank < avail) { item = pool_next + rank; stage = 1; }
				const int need = __builtin_popcountll(idle);
				pool_next += need < avail ? need : avail;
			} else if (exhausted && idle == ~0ull) break;
		}
		// C. everything this iteration needs from memory
		__builtin_amdgcn_sched_barrier(0);
		ExtLoad L;
		ext_issue(A.ix, cur, 1, stage == 3, L);
		const uint32_t *ap = stage == 1 ? (const uint32_t *)(items + item) : stage == 4 ? aux_addr(A.ix, RC_SA, tpos)
		                   : stage == 5 ? (const uint32_t *)A.ix.pac + text_chunk_word_back(A.ix, tpos) : stage == 6 ? aux_addr(A.ix, RC_ISA, tpos) : A.ix.bwt;
		const uint32_t x0 = ap[0], x1 = ap[1], x2 = ap[2], x3 = ap[3];
		const bool reading = stage == 3 || stage == 5;
		const uint32_t *qrow = A.qn + (reading ? (size_t)d_read * rw : 0);
		const int wi = reading ? i >> 3 : 0;
		const uint32_t q0 = qrow[wi];
		uint32_t q1 = 0, q2 = 0, q3 = 0;
		if (__ballot(stage == 5)) { q1 = qrow[wi > 0 ? wi - 1 : 0]; q2 = qrow[wi > 1 ? wi - 2 : 0]; q3 = qrow[wi > 2 ? wi - 3 : 0]; } // (wave-uniform)
		Biv e = Biv();
		if (__ballot(stage == 2)) e = A.P.pool[stage == 2 ? d_src : 0];
		__builtin_amdgcn_sched_barrier(0);
		// D. (the first use waits)  consume
		if (stage == 3) {
			const int c = (int)((q0 >> (4 * (i & 7))) & 15u);
			if (c > 3) finish();
			else {
				Biv ok = ext_finish(A.ix, cur, 1, c, L);
				if (ok.s < (uint64_t)min_intv) finish();
				else { ok.info = cur.info; cur = ok; --i; }
			}
		} else if (stage == 1) { d_src = x0; d_dst = x1; d_read = x2; d_xm = x3; stage = 2; }
		else if (stage == 2) { cur = e; i = (int)(d_xm & 0xffffu) - 1; min_intv = (int)(d_xm >> 16); stage = 3; }
		else if (stage == 4) {
			const uint64_t p = p40_decode(x0, x1, tpos);
			if (p == 0) finish(); // the match lies at the start of the text: nothing extends it
			else { tpos = p - 1; moved = false; stage = 5; }
		} else if (stage == 5) {
			bool more;
			const int m = text_match_back(A.ix, x0, x1, x2, x3, tpos, q0, q1, q2, q3, i, &more);
			i -= m; tpos -= (uint64_t)m; moved = moved || m > 0;
			if (!more) { if (moved) { tpos += 1; stage = 6; } else finish(); } // the match now lies at tpos + 1: its row is the new k (l stays)
		} else if (stage == 6) { cur.k = p40_decode(x0, x1, tpos); finish(); }
	}
	if (A.dbg && lane == 0) { atomicAdd(A.dbg, n_it); atomicAdd(A.dbg + 1, n_ext); atomicAdd(A.dbg + 3, 1ull); }
}

// task ids of [t0, t0 + n) by the group size their forward list needs: bins[0] <= 16 entries, [1] <= 21, [2] <= 32, [3] the rest; cnt[4]
static __global__ void __launch_bounds__(256) k_bin_tasks(const SeedTask *tasks, int t0, int n, int32_t *bin0, int32_t *bin1, int32_t *bin2, int32_t *bin3, int32_t *cnt, int mid)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	const int len = i < n ? tasks[t0 + i].n : 0; // 0: nothing to sweep (nm stays 0)
	const int cls = len == 0 ? -1 : len <= 16 ? 0 : len <= mid ? 1 : len <= 32 ? 2 : 3; // mid = 21 (16: the 21-lane bin stays empty)
	int32_t *const bins[4] = {bin0, bin1, bin2, bin3};
#pragma unroll
	for (int c = 0; c < 4; ++c) { // one atomic per wavefront and bin
		const unsigned long long m = __ballot(cls == c);
		if (!m) continue;
		int base = 0;
		if ((int)(threadIdx.x & 63) == __builtin_ctzll(m)) base = atomicAdd(cnt + c, __builtin_popcountll(m));
		base = __shfl(base, __builtin_ctzll(m));
		if (cls == c) bins[c][base + __builtin_popcountll(m & ((1ull << (threadIdx.x & 63)) - 1))] = t0 + i;
	}
}
static __global__ void __launch_bounds__(256) k_collect_heavy(const uint8_t *flag, int n, int t0, int32_t *heavy, int32_t *n_heavy)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && flag[i] == 1) heavy[atomicAdd(n_heavy, 1)] = t0 + i; // (2: text mode's tails, KSeedBwdTail)
}

struct StratArgs { IndexView ix; const uint8_t *bases; const int32_t *base_off, *lens; Biv *strat; int32_t *n_strat; int row; const uint32_t *qn; };

// third pass: forward extensions only, no lists -- the loop body is little more than extend1()
static __global__ void __launch_bounds__(64, ARX_SEED_WPE) k_strat_dyn(StratArgs A, int n, int32_t *counter, int chunk)
{
	extern __shared__ uint8_t q_lds[];
	const int lane = threadIdx.x;
	StratLane<QNibbles> ln;
	ln.finished = true;
	ItemFeeder feed;
	int r = -1;
	for (;;) {
		if (__ballot(r < 0)) {
			bool took;
			if (!feed.deal(r, took, nullptr, 0, A.qn, n, counter, chunk, q_lds, A.row)) break;
			if (took) {
				const int len = A.lens[r];
				if (len >= OPT_MIN_SEED_LEN && len <= MAX_READ_LEN) ln.start(len, QNibbles{q_lds + lane * A.row}, A.strat + (size_t)r * CAP_STRAT);
				else { A.n_strat[r] = 0; r = -1; }
			}
		}
		Biv req = Biv();
		int rc = 0;
		bool need = false;
		if (r >= 0) {
			need = ln.advance(A.ix, &req, &rc);
			if (!need) { A.n_strat[r] = ln.n; r = -1; }
		}
		if (need) {
			if (rc < 0) ln.consume_tab(A.ix, ktab_load(A.ix, req.k)); // a new start: the interval of its first K bases (dev_fm.h: k-mer table)
			else ln.consume(extend1(A.ix, req, 0, rc));
		}
	}
}

} // namespace arx
