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

} // namespace arx
