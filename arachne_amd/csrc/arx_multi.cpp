// arx_multi.cpp -- several GPUs behind ONE handle of the C ABI (the facade arx_open(prefix, n_devices, ...)): what a Go caller
// needs to drive a node without a scheduler of its own.  Host code on top of the single-device entry points of include/arachne_amd.h.
//
// Barcode groups are independent (DoRFAForOneBarcode touches only its WorkUnit, src/aligner/aligner.go:440-501) and the index is replicated
// (one arx_open per device), so a super-batch of whole barcodes is cut by pair count -- greedy longest-processing-time, the rule of
// arachne_amd/shard.py: barcodes by decreasing size to the least-loaded device, ties to the lower index -- every device's share runs on a host
// thread of its own (arx_batch_reset / run / rfa / fetch on that device's context), and the result slabs are renumbered into the order of the
// read set: exactly what ONE batch over everything returns (tests/test_multi.py).  Inside one process the reads reach every GPU from host
// memory, so no GPU-to-GPU transfer exists on this path; the RCCL form of the same dataflow, for reads that arrive on one GPU, is
// arachne_amd/shard.py (step_device).
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <numeric>
#include <string>
#include <thread>
#include <vector>
#include "../../include/arachne_amd.h"

namespace {

struct Share { // one device's part of a super-batch
	std::vector<int32_t> barcodes;      // indices into the super-batch, increasing
	std::vector<uint8_t> bases, flags;
	std::vector<int32_t> lens;
	std::vector<int64_t> pair_off;
	std::vector<int32_t> reg_off, cand_off;
	std::vector<arx_reg> regs; std::vector<arx_aln> alns; std::vector<uint32_t> cigars; std::vector<arx_cand> cands;
	int64_t counts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, n_cands = 0;
	std::string error;
};

struct Multi {
	std::vector<arx_ctx *> ctx;
	std::vector<arx_batch *> batch;
	std::vector<Share> share;
	std::string error;
	// merged result, read-set order
	std::vector<int32_t> reg_off, cand_off, dev_of_bc;
	std::vector<arx_reg> regs; std::vector<arx_aln> alns; std::vector<uint32_t> cigars; std::vector<arx_cand> cands;
};

void run_share(Multi *m, int d, double penalty, const int64_t *cs, const int64_t *ce)
{
	Share &s = m->share[d];
	s.error.clear();
	if (s.barcodes.empty()) return;
	arx_ctx *c = m->ctx[d];
	const int32_t nr = (int32_t)s.lens.size();
	int rc = m->batch[d] ? arx_batch_reset(c, m->batch[d], nr, s.bases.data(), s.lens.data()) : arx_batch_create(c, nr, s.bases.data(), s.lens.data(), &m->batch[d]);
	if (rc == ARX_OK) rc = arx_batch_run(c, m->batch[d], ARX_STAGE_ALN);
	if (rc == ARX_OK) rc = arx_batch_rfa(c, m->batch[d], (int32_t)s.barcodes.size(), s.pair_off.data(), s.flags.data(), penalty, cs, ce, &s.n_cands);
	if (rc == ARX_OK) rc = arx_batch_counts(c, m->batch[d], s.counts);
	if (rc == ARX_OK) {
		s.reg_off.resize((size_t)nr + 1); s.cand_off.resize((size_t)nr + 1);
		s.regs.resize((size_t)s.counts[1] + 1); s.alns.resize((size_t)s.counts[1] + 1); s.cigars.resize((size_t)s.counts[2] + 1); s.cands.resize((size_t)s.n_cands + 1);
		rc = arx_batch_fetch(c, m->batch[d], s.reg_off.data(), s.regs.data(), s.alns.data(), s.cigars.data());
	}
	if (rc == ARX_OK) rc = arx_batch_rfa_fetch(c, m->batch[d], s.cand_off.data(), s.cands.data());
	if (rc != ARX_OK) { const char *e = arx_last_error(c); s.error = std::string("device ") + std::to_string(d) + ": " + (e ? e : "error"); }
}

} // namespace

extern "C" {

int arx_multi_open(const char *prefix, int32_t n_devices, const int32_t *devices, arx_multi **out, char *msg, int32_t msg_cap)
{
	*out = nullptr;
	if (n_devices <= 0) { if (msg && msg_cap > 0) snprintf(msg, (size_t)msg_cap, "n_devices must be positive"); return ARX_E_ARG; }
	Multi *m = new Multi();
	m->ctx.assign((size_t)n_devices, nullptr); m->batch.assign((size_t)n_devices, nullptr); m->share.resize((size_t)n_devices);
	std::vector<int> rcs((size_t)n_devices, ARX_OK);
	std::vector<std::string> errs((size_t)n_devices);
	std::vector<std::thread> th; // the replicas load side by side
	for (int d = 0; d < n_devices; ++d)
		th.emplace_back([&, d]() { rcs[d] = arx_open(prefix, devices ? devices[d] : d, &m->ctx[d]); if (rcs[d] != ARX_OK) { const char *e = arx_last_error(nullptr); errs[d] = e ? e : "arx_open failed"; } });
	for (auto &t : th) t.join();
	for (int d = 0; d < n_devices; ++d)
		if (rcs[d] != ARX_OK) {
			if (msg && msg_cap > 0) snprintf(msg, (size_t)msg_cap, "device %d: %s", devices ? devices[d] : d, errs[d].c_str());
			for (arx_ctx *c : m->ctx) if (c) arx_close(c);
			delete m;
			return rcs[d];
		}
	*out = (arx_multi *)m;
	return ARX_OK;
}

int arx_multi_run(arx_multi *h, int32_t n_reads, const uint8_t *bases, const int32_t *lens, int32_t n_barcodes, const int64_t *bc_pair_off, const uint8_t *do_rfa,
                  double penalty, const int64_t *cen_start, const int64_t *cen_end, arx_multi_result *out)
{
	Multi *m = (Multi *)h;
	const int D = (int)m->ctx.size();
	if (n_barcodes <= 0 || bc_pair_off[0] != 0 || 2 * bc_pair_off[n_barcodes] != n_reads) { m->error = "barcode offsets must cover the reads"; return ARX_E_ARG; }
	// LPT assignment by pair count (the rule of arachne_amd/shard.py: lpt_assign)
	std::vector<int32_t> order((size_t)n_barcodes);
	std::iota(order.begin(), order.end(), 0);
	std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return bc_pair_off[a + 1] - bc_pair_off[a] > bc_pair_off[b + 1] - bc_pair_off[b]; });
	std::vector<int64_t> load((size_t)D, 0);
	m->dev_of_bc.assign((size_t)n_barcodes, 0);
	for (Share &s : m->share) s.barcodes.clear();
	for (int32_t b : order) {
		int best = 0;
		for (int d = 1; d < D; ++d) if (load[d] < load[best]) best = d;
		m->share[best].barcodes.push_back(b); load[best] += bc_pair_off[b + 1] - bc_pair_off[b]; m->dev_of_bc[b] = best;
	}
	std::vector<int64_t> base_off((size_t)n_reads + 1, 0);
	for (int32_t r = 0; r < n_reads; ++r) base_off[(size_t)r + 1] = base_off[r] + lens[r];
	std::vector<int64_t> local_read0((size_t)n_barcodes, 0); // first read of a barcode inside its device's batch
	for (int d = 0; d < D; ++d) {
		Share &s = m->share[d];
		std::sort(s.barcodes.begin(), s.barcodes.end());
		s.bases.clear(); s.lens.clear(); s.flags.clear(); s.pair_off.assign(1, 0);
		for (int32_t b : s.barcodes) {
			const int64_t r0 = 2 * bc_pair_off[b], r1 = 2 * bc_pair_off[b + 1];
			local_read0[b] = (int64_t)s.lens.size();
			s.bases.insert(s.bases.end(), bases + base_off[r0], bases + base_off[r1]);
			s.lens.insert(s.lens.end(), lens + r0, lens + r1);
			s.flags.push_back(do_rfa[b]);
			s.pair_off.push_back(s.pair_off.back() + (bc_pair_off[b + 1] - bc_pair_off[b]));
		}
	}
	{
		std::vector<std::thread> th;
		for (int d = 0; d < D; ++d) th.emplace_back(run_share, m, d, penalty, cen_start, cen_end);
		for (auto &t : th) t.join();
	}
	for (int d = 0; d < D; ++d) if (!m->share[d].error.empty()) { m->error = m->share[d].error; return ARX_E_DEVICE; }
	// merge in read-set order: offsets first, then the slabs of every barcode
	int64_t n_reg = 0, n_cig = 0, n_cand = 0;
	std::vector<int64_t> reg0((size_t)n_barcodes), cig0((size_t)n_barcodes), cand0((size_t)n_barcodes);
	for (int32_t b = 0; b < n_barcodes; ++b) {
		const Share &s = m->share[m->dev_of_bc[b]];
		const int64_t lr0 = local_read0[b], nr = 2 * (bc_pair_off[b + 1] - bc_pair_off[b]);
		const int64_t r0 = s.reg_off[lr0], r1 = s.reg_off[lr0 + nr], c0 = s.cand_off[lr0], c1 = s.cand_off[lr0 + nr];
		const int64_t w0 = r1 > r0 ? s.alns[r0].cigar_off : 0, w1 = r1 > r0 ? s.alns[r1 - 1].cigar_off + s.alns[r1 - 1].n_cigar : 0;
		reg0[b] = n_reg; cig0[b] = n_cig; cand0[b] = n_cand;
		n_reg += r1 - r0; n_cig += w1 - w0; n_cand += c1 - c0;
	}
	if (n_reg >= ((int64_t)1 << 31) || n_cig >= ((int64_t)1 << 31) || n_cand >= ((int64_t)1 << 31)) { m->error = "the merged result exceeds 32-bit offsets: hand over fewer barcodes per call"; return ARX_E_TOO_LARGE; }
	m->reg_off.assign((size_t)n_reads + 1, 0); m->cand_off.assign((size_t)n_reads + 1, 0);
	m->regs.resize((size_t)n_reg + 1); m->alns.resize((size_t)n_reg + 1); m->cigars.resize((size_t)n_cig + 1); m->cands.resize((size_t)n_cand + 1);
	auto merge_range = [&](int32_t b_lo, int32_t b_hi) {
		for (int32_t b = b_lo; b < b_hi; ++b) {
			const Share &s = m->share[m->dev_of_bc[b]];
			const int64_t lr0 = local_read0[b], g0 = 2 * bc_pair_off[b], nr = 2 * (bc_pair_off[b + 1] - bc_pair_off[b]);
			const int64_t r0 = s.reg_off[lr0], r1 = s.reg_off[lr0 + nr], c0 = s.cand_off[lr0], c1 = s.cand_off[lr0 + nr];
			const int64_t w0 = r1 > r0 ? s.alns[r0].cigar_off : 0, w1 = r1 > r0 ? s.alns[r1 - 1].cigar_off + s.alns[r1 - 1].n_cigar : 0;
			for (int64_t k = 0; k < nr; ++k) { m->reg_off[g0 + k] = (int32_t)(s.reg_off[lr0 + k] - r0 + reg0[b]); m->cand_off[g0 + k] = (int32_t)(s.cand_off[lr0 + k] - c0 + cand0[b]); }
			if (r1 > r0) memcpy(&m->regs[reg0[b]], &s.regs[r0], sizeof(arx_reg) * (size_t)(r1 - r0));
			for (int64_t k = r0; k < r1; ++k) { arx_aln a = s.alns[k]; a.cigar_off = (int32_t)(a.cigar_off - w0 + cig0[b]); m->alns[reg0[b] + (k - r0)] = a; }
			if (w1 > w0) memcpy(&m->cigars[cig0[b]], &s.cigars[w0], 4 * (size_t)(w1 - w0));
			for (int64_t k = c0; k < c1; ++k) {
				arx_cand c = s.cands[k];
				if (c.reg >= 0) c.reg = (int32_t)(c.reg - r0 + reg0[b]);
				c.read = (int32_t)(c.read - lr0 + g0);
				m->cands[cand0[b] + (k - c0)] = c;
			}
		}
	};
	{
		const int T = 8;
		std::vector<std::thread> th;
		for (int t = 0; t < T; ++t) th.emplace_back(merge_range, (int32_t)((int64_t)n_barcodes * t / T), (int32_t)((int64_t)n_barcodes * (t + 1) / T));
		for (auto &t : th) t.join();
	}
	m->reg_off[(size_t)n_reads] = (int32_t)n_reg; m->cand_off[(size_t)n_reads] = (int32_t)n_cand;
	out->n_reads = n_reads; out->n_regs = n_reg; out->n_cigar = n_cig; out->n_cands = n_cand;
	out->reg_off = m->reg_off.data(); out->regs = m->regs.data(); out->alns = m->alns.data(); out->cigars = m->cigars.data();
	out->cand_off = m->cand_off.data(); out->cands = m->cands.data(); out->device_of_barcode = m->dev_of_bc.data();
	return ARX_OK;
}

const char *arx_multi_error(arx_multi *h) { return ((Multi *)h)->error.c_str(); }

int arx_multi_contigs(arx_multi *h, int32_t *n, const char *const **names, const int64_t **offsets, const int32_t **lens, const int32_t **is_alt, int64_t *l_pac)
{
	return arx_contigs(((Multi *)h)->ctx[0], n, names, offsets, lens, is_alt, l_pac);
}

void arx_multi_close(arx_multi *h)
{
	Multi *m = (Multi *)h;
	if (!m) return;
	for (size_t d = 0; d < m->ctx.size(); ++d) if (m->ctx[d]) arx_close(m->ctx[d]); // frees the device's batch as well
	delete m;
}

}
