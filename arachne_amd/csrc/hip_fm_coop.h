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

// k_seed_dyn / k_strat_dyn: the seeding passes of mem_collect_intv with persistent lanes.  A read's SMEM search is a serial
// chain of several hundred bidirectional extensions whose count varies a lot between reads (repeats), so a static
// one-read-per-lane mapping makes every wavefront wait for its slowest read.  Here a lane takes its next read as soon as the
// previous one is finished (wave-level chunk reservation as in k_locate_dyn), and every iteration of the loop is one
// extension for all busy lanes: the lane program (SeedLane / StratLane, dev_fm.h) runs each lane's bookkeeping up to its next
// extension, the lanes reconverge on extend1().  The read's bases are staged in LDS (4-bit codes, row stride 33 words so
// that the 64 lanes hit different banks): the search reads one base per extension, which would otherwise be a dependent HBM
// access in front of the Occ block loads.
constexpr int SEED_ROW = 132; // bytes per lane: 256 bases + pad

struct SeedArgs {
	IndexView ix; const uint8_t *bases; const int32_t *base_off, *lens;
	Biv *intv; int32_t *n_intv; Biv *scratch; int list_cap; uint32_t *err;
};
struct StratArgs { IndexView ix; const uint8_t *bases; const int32_t *base_off, *lens; Biv *strat; int32_t *n_strat; };

// Hands reads to idle lanes: a wave reserves `chunk` reads with one atomic and deals them out by ballot rank; the wave then
// copies the bases of every newly taken read into the taking lane's LDS row.  r < 0 marks an idle lane.  Returns false when
// nothing is left and the whole wave is idle.
struct ReadFeeder {
	int pool_next = 0, pool_end = 0; bool exhausted = false; // wave-uniform
	__device__ bool deal(int &r, bool &took, const uint8_t *bases, const int32_t *base_off, const int32_t *lens, int n, int32_t *counter, int chunk, uint8_t *q_lds)
	{
		const int lane = threadIdx.x;
		took = false;
		const unsigned long long idle = __ballot(r < 0);
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
		took = r < 0 && rank < avail;
		if (took) r = pool_next + rank;
		const int need = __builtin_popcountll(idle);
		pool_next += need < avail ? need : avail;
		unsigned long long fresh = __ballot(took);
		while (fresh) { // 128 bases per sweep of the wave
			const int src = __builtin_ctzll(fresh);
			fresh &= fresh - 1;
			const int rs = __shfl(r, src);
			const int len = lens[rs];
			const uint8_t *b = bases + base_off[rs];
			if (len <= MAX_READ_LEN)
				for (int k = 2 * lane; k < len; k += 128) {
					const int lo = b[k], hi = k + 1 < len ? b[k + 1] : 4;
					q_lds[src * SEED_ROW + (k >> 1)] = (uint8_t)(lo | hi << 4);
				}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		return true;
	}
};

static __global__ void __launch_bounds__(64) k_seed_dyn(SeedArgs A, int n, int32_t *counter, int batch, int chunk)
{
	__shared__ uint8_t q_lds[64 * SEED_ROW];
	const int lane = threadIdx.x;
	const int slot = blockIdx.x * 64 + lane;
	SmemScratch sc; sc.v0 = A.scratch + (size_t)slot * 3 * A.list_cap; sc.v1 = sc.v0 + A.list_cap; sc.mem = sc.v1 + A.list_cap;
	SeedLane<QNibbles> ln;
	ln.state = SeedLane<QNibbles>::ST_DONE;
	ReadFeeder feed;
	int r = -1; // read this lane is searching, -1 = idle
	Biv req = Biv();
	int rb = 0, rc = 0;
	bool have_req = false;
	for (;;) {
		// cheap part: a lane whose search is in an extending state gets its next request
		if (r >= 0 && !have_req) have_req = ln.advance(A.ix, &req, &rb, &rc, false);
		// The rest (finishing a read, taking the next one, the bookkeeping between two searches) is rare per lane but long, and
		// a wavefront pays for a divergent path whenever ONE lane is in it.  Lanes therefore wait in front of it until `batch`
		// of them do (or nobody can extend): the wave then runs that code once for all of them.
		const int waiting = __builtin_popcountll(__ballot(!have_req));
		if (waiting >= batch || waiting == 64) {
			if (r >= 0 && !have_req && ln.done()) { // a finished read: its interval count (the third pass and the sort follow in their own kernels)
				A.n_intv[r] = ln.n;
				if (ln.overflow) atomicOr(A.err, ERR_INTV_OVERFLOW);
				r = -1;
			}
			bool took;
			if (!feed.deal(r, took, A.bases, A.base_off, A.lens, n, counter, chunk, q_lds)) break;
			if (took) {
				int len = A.lens[r];
				if (len > MAX_READ_LEN) { atomicOr(A.err, ERR_READ_TOO_LONG); len = 0; }
				if (len >= OPT_MIN_SEED_LEN) ln.start(sc, len, QNibbles{q_lds + lane * SEED_ROW}, A.intv + (size_t)r * CAP_INTV, CAP_INTV);
				else { A.n_intv[r] = 0; r = -1; } // nothing to seed; the lane asks again next time round
			}
			// everything up to the next request (a read that ends here is written out the next time round)
			if (r >= 0 && !have_req) have_req = ln.advance(A.ix, &req, &rb, &rc, true);
		}
		if (have_req) { ln.consume(req, extend1(A.ix, req, rb, rc)); have_req = false; }
	}
}

// third pass: forward extensions only, no lists -- the loop body is little more than extend1()
static __global__ void __launch_bounds__(64) k_strat_dyn(StratArgs A, int n, int32_t *counter, int chunk)
{
	__shared__ uint8_t q_lds[64 * SEED_ROW];
	const int lane = threadIdx.x;
	StratLane<QNibbles> ln;
	ln.finished = true;
	ReadFeeder feed;
	int r = -1;
	for (;;) {
		if (__ballot(r < 0)) {
			bool took;
			if (!feed.deal(r, took, A.bases, A.base_off, A.lens, n, counter, chunk, q_lds)) break;
			if (took) {
				const int len = A.lens[r];
				if (len >= OPT_MIN_SEED_LEN && len <= MAX_READ_LEN) ln.start(len, QNibbles{q_lds + lane * SEED_ROW}, A.strat + (size_t)r * CAP_STRAT);
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
		if (need) ln.consume(extend1(A.ix, req, 0, rc));
	}
}

} // namespace arx
