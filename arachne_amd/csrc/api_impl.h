// api_impl.h -- implementation of the C-ABI declared in include/arachne_amd.h on top of Pipeline<RT>.
// The product build (arx_api.hip) instantiates it with HipRT.
//
// Concurrency: a context owns the index in HBM; every batch owns its own runtime (HIP stream, scan scratch, event
// timers), so several host threads can drive several batches at once and their kernels overlap on the device -- the
// step/DP round trips of one batch hide behind the kernels of the others.
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include "../../include/arachne_amd.h"
#include "index_io.h"
#include "index_build.h"
#include "pipeline.h"
#include "pipeline_rfa.h"
#include "pipeline_post.h"

namespace arx {

template <class RT> struct Batch;

// SA[i * d] for every i: the suffix-array sample the locate kernel walks to.  The files sample every 32nd row (bwt.c:62,
// bwtindex.c:309); bwt_sa's LF walk from a row to the next sampled one (bwt.c:86-96) is what locating a seed costs, 15.5 steps on
// average.  A sample every `d`-th row, computed once per arx_open on the device with that same walk, gives the same values in
// (d - 1) / 2 steps.
struct KSaDense {
	IndexView ix; uint64_t *out; int d;
	ARX_DEV void operator()(int i, int) const { out[i] = sa_lookup(ix, (uint64_t)i * (uint64_t)d); }
};

// The whole suffix array and its inverse (IndexView::sa40 / isa40), from the rows the files sample: one thread per sampled row walks bwt_sa's LF
// steps (bwt.c:86-96) to the next sampled row; the rows it passes are those whose own walk would end where it started, so every row and every
// text position is visited by exactly one thread, which knows both its row and its position.  Row 0 (the empty suffix, sa[0] = -1) stands at
// position seq_len and gets no entry.
struct KSaWalk {
	IndexView ix; uint8_t *sa40, *isa40;
	ARX_DEV void operator()(int j, int) const
	{
		const uint64_t mask = (uint64_t)ix.sa_intv - 1;
		uint64_t r = (uint64_t)j * (uint64_t)ix.sa_intv, p = j ? ix.sa[j] : ix.seq_len;
		if (j == 0) p40_store(sa40, 0, 0xffffffffffull);
		for (;;) {
			if (r) { p40_store(sa40, r, p); p40_store(isa40, p, r); }
			r = lf_step(ix, r); --p;
			if ((r & mask) == 0) break;
		}
	}
};

// k-mer table (dev_fm.h: ktab_*): level d + 1 from level d, one thread per parent; the child of base b is the forward extension by b
struct KKmerLevel {
	IndexView ix; const uint64_t *in; uint64_t *out; int d; // in: 4^d entries (null: d == 0, the parents are the four single bases)
	ARX_DEV void operator()(int p, int) const
	{
		if (d == 0) { uint64_t w[2]; ktab_pack(set_intv(ix, p), w); out[2 * (size_t)p] = w[0]; out[2 * (size_t)p + 1] = w[1]; return; }
		const Biv ik = ktab_unpack(in[2 * (size_t)p], in[2 * (size_t)p + 1]);
		for (int b = 0; b < 4; ++b) {
			const Biv ok = extend1(ix, ik, 0, 3 - b);
			uint64_t w[2]; ktab_pack(ok, w);
			const size_t c = (size_t)p | (size_t)b << (2 * d);
			out[2 * c] = w[0]; out[2 * c + 1] = w[1];
		}
	}
};

struct KOccRepack { // BWA's block layout -> the checkpointed one (dev_fm.h), one thread per 64-byte block
	uint32_t *bwt;
	ARX_DEV void operator()(int i, int) const { occ_repack_block(bwt + (size_t)i * 16); }
};

template <class RT> struct Context {
	RT rt;                      // index uploads
	int device = 0;
	HostIndex hix;
	IndexView ix;
	std::vector<void *> dev_index;
	std::vector<const char *> name_ptrs;
	std::mutex mu;
	std::string last_error;
	std::set<Batch<RT> *> live;
	std::map<std::string, KernelTimer> tm_done; // timers of batches already freed
	bool timing = false;
	uint64_t index_bytes = 0;   // device memory the index holds (arx_index_info)

	void set_error(const std::string &e) { std::lock_guard<std::mutex> g(mu); last_error = e; }

	std::string open(const std::string &prefix, int dev)
	{
		device = dev;
		std::string e = rt.init(dev);
		if (!e.empty()) return e;
		e = load_index(prefix, hix);
		if (!e.empty()) return e;
		const uint64_t free_at_start = rt.free_bytes();
		auto up = [&](const void *src, size_t bytes) { void *d = rt.template palloc<uint8_t>(bytes + 64); rt.h2d(d, src, bytes); dev_index.push_back(d); return d; };
		{ // the Occ blocks go to HBM in the checkpointed layout of dev_fm.h (the files keep BWA's): uploaded as they are, re-packed in place
			const size_t n_blk = hix.bwt.size() / 16;
			if (n_blk >= 0x7fffffffull) return "index too large: more than 2^31 Occ blocks";
			uint32_t *d = (uint32_t *)up(hix.bwt.data(), hix.bwt.size() * 4);
			std::vector<uint32_t>().swap(hix.bwt);
			KOccRepack kr{d};
			rt.launch_wide("occ_repack", (int)n_blk, kr);
			rt.sync();
			ix.bwt = d;
		}
		ix.sa = (const uint64_t *)up(hix.sa.data(), hix.sa.size() * 8);
		std::vector<uint64_t>().swap(hix.sa);
		{ // the packed forward strand behind 16 spare bytes: text mode reads the 16 bytes that END at a word of the reverse strand's mirror (dev_fm.h)
			uint8_t *d = rt.template palloc<uint8_t>(hix.pac.size() + 16 + 64);
			rt.memset0(d, 16);
			rt.h2d(d + 16, hix.pac.data(), hix.pac.size());
			dev_index.push_back(d);
			ix.pac = d + 16;
		}
		std::vector<uint8_t>().swap(hix.pac);
		ix.ann_off = (const int64_t *)up(hix.ann_off.data(), hix.ann_off.size() * 8);
		ix.ann_len = (const int32_t *)up(hix.ann_len.data(), hix.ann_len.size() * 4);
		ix.ann_alt = (const int32_t *)up(hix.ann_alt.data(), hix.ann_alt.size() * 4);
		ix.primary = hix.primary; ix.seq_len = hix.seq_len;
		for (int i = 0; i < 5; ++i) ix.L2[i] = hix.L2[i];
		ix.l_pac = hix.l_pac; ix.n_seqs = (int)hix.names.size(); ix.sa_intv = hix.sa_intv;
		{ // k-mer table for the third seeding pass (dev_fm.h): K from the genome size -- the largest K with 4^K <= symbols, at most 16 (GRCh38:
		  // 16, 4.3 G entries = 69 GB of the 288 GB; measured at GRCh38 size: the pass takes 9.7 ms per step without a table, 5.6 at K = 12,
		  // 4.2 at 14, 3.3 at 16) -- and never more than a third of the device memory that is free; ARX_KMER_K overrides, 0 = no table
			const char *e = getenv("ARX_KMER_K");
			int K = 0;
			if (e) K = atoi(e);
			else { while (K < 16 && ((uint64_t)1 << (2 * (K + 1))) <= ix.seq_len) ++K; }
			if (K > OPT_MIN_SEED_LEN - 3) K = OPT_MIN_SEED_LEN - 3;
			while (K >= 4 && ((uint64_t)20 << (2 * K)) > rt.free_bytes() / 3) --K; // 16 bytes per entry, plus the level before it while it is built
			ix.ktab = nullptr; ix.ktab_k = 0; ix.klv = nullptr; ix.klv_k = 0;
			if (K >= 4) {
				// levels 1 .. Kf are kept back to back for the forward extensions of the SMEM pass (every depth of a list prefix is needed there:
				// dev_fm.h FwdLane); Kf = min(K, 14): 5.7 GB.  ARX_KMER_FWD=0: none (the forward kernels walk base by base)
				const char *ef = getenv("ARX_KMER_FWD");
				const int Kf = (ef && atoi(ef) == 0) ? 0 : (K < 14 ? K : 14);
				uint64_t *lv = nullptr;
				if (Kf > 0) lv = rt.template palloc<uint64_t>(2 * ((((size_t)1 << (2 * (Kf + 1))) - 4) / 3) + 2);
				uint64_t *prev = nullptr; bool prev_owned = false;
				for (int d = 0; d < K; ++d) { // level d + 1: 4^(d+1) entries
					const size_t n_out = (size_t)1 << (2 * (d + 1));
					const bool in_lv = d + 1 <= Kf;
					uint64_t *out = in_lv ? lv + 2 * ((n_out - 4) / 3) : rt.template palloc<uint64_t>(2 * n_out + 2);
					KKmerLevel kk{ix, prev, out, d};
					rt.launch_wide("kmer_level", d == 0 ? 4 : (int)(n_out >> 2), kk);
					rt.sync();
					if (prev && prev_owned) rt.pfree(prev);
					prev = out; prev_owned = !in_lv;
				}
				if (prev_owned) dev_index.push_back(prev);
				if (lv) dev_index.push_back(lv);
				ix.ktab = prev; ix.ktab_k = K; ix.klv = lv; ix.klv_k = Kf;
			}
		}
		ix.sa40 = nullptr; ix.isa40 = nullptr;
		{ // the whole suffix array and its inverse, 5 bytes per entry each (GRCh38: 2 x 31 GB), when they take no more than half of what is free
		  // after the tables; ARX_TEXT_INDEX=0: never.  With them locating a seed is one load and the first-pass forward extensions compare
		  // text once one occurrence is left (dev_fm.h: text mode)
			const char *e = getenv("ARX_TEXT_INDEX");
			const uint64_t bytes = 5 * (ix.seq_len + 1) + 64;
			const uint64_t n_sa = (ix.seq_len + (uint64_t)ix.sa_intv) / (uint64_t)ix.sa_intv;
			if (!(e && atoi(e) == 0) && (ix.sa_intv & (ix.sa_intv - 1)) == 0 && 2 * bytes <= rt.free_bytes() / 2 && n_sa < 0x7fffffffull) {
				uint8_t *sa40 = rt.template palloc<uint8_t>((size_t)bytes), *isa40 = rt.template palloc<uint8_t>((size_t)bytes);
				KSaWalk kw{ix, sa40, isa40};
				rt.launch_wide("sa_walk", (int)n_sa, kw);
				rt.sync();
				dev_index.push_back(sa40); dev_index.push_back(isa40);
				ix.sa40 = sa40; ix.isa40 = isa40;
			}
		}
		if (!ix.sa40) { // denser suffix-array sample (ARX_SA_DENSE: rows per sample, a power of two; at least the file's interval switches it off)
			const char *e = getenv("ARX_SA_DENSE");
			const int d = e ? atoi(e) : 4; // every 4th row since round 3 (12 GB at GRCh38 size; locate 4.9 -> 2.5 ms per step, arx_open +1.7 s); 8 in round 2
			const uint64_t n2 = (ix.seq_len + (uint64_t)d) / (uint64_t)d;
			if (d >= 1 && d < ix.sa_intv && (d & (d - 1)) == 0 && n2 < 0x7fffffffull) {
				uint64_t *dense = rt.template palloc<uint64_t>((size_t)n2 + 8);
				KSaDense kd{ix, dense, d};
				rt.launch_wide("sa_dense", (int)n2, kd);
				rt.sync();
				void *old = (void *)ix.sa;
				for (auto &p : dev_index) if (p == old) { rt.pfree(p); p = dense; }
				ix.sa = dense; ix.sa_intv = d;
			}
		}
		for (auto &n : hix.names) name_ptrs.push_back(n.c_str());
		{ const uint64_t f = rt.free_bytes(); index_bytes = free_at_start > f ? free_at_start - f : 0; }
		return "";
	}
	~Context() { for (void *p : dev_index) rt.pfree(p); }
};

template <class RT> struct Batch {
	Context<RT> *ctx;
	RT rt;                      // this batch's stream
	Pipeline<RT> pipe;
	typename Pipeline<RT>::DeviceBatch db;
	typename Pipeline<RT>::Work work;
	BatchResult res;
	int done_stage = 0;
	std::vector<int32_t> lens_host;
	RfaResult rfa;
	PostResult post; std::vector<size_t> post_mark;
	std::vector<size_t> rfa_mark; bool rfa_marked = false; // arena state after ARX_STAGE_ALN: a repeated arx_batch_rfa reuses the same memory
	// arx_batch_detach: the dense results copied aside (device memory of their own, outside the work arena) so that the handle can take its
	// next reads while a second host thread takes them home through a stream of its own (arx_batch_fetch_detached)
	struct Detached {
		bool valid = false; int32_t n_reads = 0; int64_t n_regs = 0, n_cig = 0, n_cands = 0;
		size_t o_reg_off = 0, o_regs = 0, o_alns = 0, o_cig = 0, o_cands = 0;
		std::vector<int32_t> cand_off;
	} det;
	char *slab = nullptr; size_t slab_cap = 0;
	RT rt_copy; bool copy_ready = false;
	explicit Batch(Context<RT> *c) : ctx(c), pipe(rt, c->ix)
	{
		std::string e = rt.init(c->device);
		if (!e.empty()) throw std::runtime_error(e);
		std::lock_guard<std::mutex> g(c->mu);
		c->live.insert(this);
	}
	~Batch()
	{
		rt.bind();
		pipe.free_work(work); pipe.release(db);
		if (slab) rt.pfree(slab);
		std::lock_guard<std::mutex> g(ctx->mu);
		for (auto &kv : rt.timers()) { KernelTimer &t = ctx->tm_done[kv.first]; t.ms += kv.second.ms; t.calls += kv.second.calls; t.items += kv.second.items; }
		ctx->live.erase(this);
	}
};

} // namespace arx

// every entry converts C++ exceptions (HIP errors, bad_alloc) into an error code: the library never aborts the caller
#define ARX_TRY(ctxp, ...) try { __VA_ARGS__ } catch (const std::exception &ex_) { if (ctxp) (ctxp)->set_error(ex_.what()); return ARX_E_DEVICE; }

#define ARX_DEFINE_C_API(RT)                                                                                                        \
	using Ctx = arx::Context<RT>;                                                                                                   \
	using Bat = arx::Batch<RT>;                                                                                                     \
	static thread_local std::string g_open_error;                                                                                   \
	extern "C" {                                                                                                                    \
	int arx_index_build(const char *fasta, const char *prefix, char *msg, int32_t msg_cap)                                          \
	{                                                                                                                               \
		std::string e;                                                                                                              \
		try { e = arx::build_index(fasta, prefix, RT::bwt_sa_fn()); } catch (const std::exception &ex) { e = ex.what(); }                            \
		if (msg && msg_cap > 0) snprintf(msg, msg_cap, "%s", e.c_str());                                                            \
		return e.empty() ? ARX_OK : ARX_E_OPEN;                                                                                     \
	}                                                                                                                               \
	int arx_open(const char *prefix, int device, arx_ctx **out)                                                                     \
	{                                                                                                                               \
		*out = 0;                                                                                                                   \
		Ctx *c = 0;                                                                                                                 \
		std::string e;                                                                                                              \
		try { c = new Ctx(); e = c->open(prefix, device); } catch (const std::exception &ex) { e = ex.what(); }                     \
		if (!e.empty()) { g_open_error = e; delete c; return ARX_E_OPEN; }                                                          \
		*out = (arx_ctx *)c;                                                                                                        \
		return ARX_OK;                                                                                                              \
	}                                                                                                                               \
	void arx_close(arx_ctx *h)                                                                                                      \
	{                                                                                                                               \
		Ctx *c = (Ctx *)h;                                                                                                          \
		if (!c) return;                                                                                                             \
		try {                                                                                                                       \
			std::vector<Bat *> left;                                                                                                \
			{ std::lock_guard<std::mutex> g(c->mu); left.assign(c->live.begin(), c->live.end()); }                                  \
			for (Bat *b : left) delete b; /* batches the caller did not free: their handles die with the context */                 \
			delete c;                                                                                                               \
		} catch (...) {}                                                                                                            \
	}                                                                                                                               \
	const char *arx_last_error(arx_ctx *h)                                                                                          \
	{                                                                                                                               \
		if (!h) return g_open_error.c_str();                                                                                        \
		Ctx *c = (Ctx *)h;                                                                                                          \
		std::lock_guard<std::mutex> g(c->mu);                                                                                       \
		g_open_error = c->last_error;                                                                                               \
		return g_open_error.c_str();                                                                                                \
	}                                                                                                                               \
	const char *arx_backend(void) { return RT::name(); }                                                                            \
	int arx_host_register(void *p, int64_t bytes) { return p && bytes > 0 && RT::host_register(p, (size_t)bytes) == 0 ? ARX_OK : ARX_E_ARG; } \
	int arx_host_unregister(void *p) { return p && RT::host_unregister(p) == 0 ? ARX_OK : ARX_E_ARG; }                              \
	int arx_index_info(arx_ctx *h, int64_t *info)                                                                                   \
	{                                                                                                                               \
		Ctx *c = (Ctx *)h;                                                                                                          \
		info[0] = (int64_t)c->ix.seq_len; info[1] = c->ix.ktab ? c->ix.ktab_k : 0; info[2] = c->ix.klv ? c->ix.klv_k : 0;           \
		info[3] = c->ix.sa40 ? 1 : c->ix.sa_intv; info[4] = c->ix.isa40 ? 1 : 0; info[5] = (int64_t)c->index_bytes;                 \
		info[6] = info[7] = 0;                                                                                                      \
		return ARX_OK;                                                                                                              \
	}                                                                                                                               \
	int arx_contigs(arx_ctx *h, int32_t *n, const char *const **names, const int64_t **offsets, const int32_t **lens,               \
	                const int32_t **is_alt, int64_t *l_pac)                                                                         \
	{                                                                                                                               \
		Ctx *c = (Ctx *)h;                                                                                                          \
		*n = (int32_t)c->hix.names.size(); *names = c->name_ptrs.data(); *offsets = c->hix.ann_off.data();                          \
		*lens = c->hix.ann_len.data(); *is_alt = c->hix.ann_alt.data(); *l_pac = c->hix.l_pac;                                      \
		return ARX_OK;                                                                                                              \
	}                                                                                                                               \
	int arx_batch_create(arx_ctx *h, int32_t n_reads, const uint8_t *bases, const int32_t *lens, arx_batch **out)                   \
	{                                                                                                                               \
		Ctx *c = (Ctx *)h;                                                                                                          \
		*out = 0;                                                                                                                   \
		if (n_reads <= 0 || (n_reads & 1)) { c->set_error("n_reads must be positive and even (read 2i/2i+1 are mates)"); return ARX_E_ARG; } \
		{ int64_t tot = 0;                                                                                                          \
		  for (int i = 0; i < n_reads; ++i) { if (lens[i] < 0 || lens[i] > arx::MAX_READ_LEN) { c->set_error("read length outside [0, 255]"); return ARX_E_ARG; } tot += lens[i]; } \
		  if (tot >= ((int64_t)1 << 31) - 64) { c->set_error("batch too large: more than 2^31 bases, split the batch"); return ARX_E_TOO_LARGE; } } \
		ARX_TRY(c, Bat *b = new Bat(c); b->rt.bind(); b->db = b->pipe.upload(bases, lens, n_reads); b->lens_host.assign(lens, lens + n_reads); *out = (arx_batch *)b;) \
		return ARX_OK;                                                                                                              \
	}                                                                                                                               \
	int arx_batch_reset(arx_ctx *h, arx_batch *bh, int32_t n_reads, const uint8_t *bases, const int32_t *lens)                      \
	{                                                                                                                               \
		Ctx *c = (Ctx *)h; Bat *b = (Bat *)bh;                                                                                      \
		if (n_reads <= 0 || (n_reads & 1)) { c->set_error("n_reads must be positive and even (read 2i/2i+1 are mates)"); return ARX_E_ARG; } \
		{ int64_t tot = 0;                                                                                                          \
		  for (int i = 0; i < n_reads; ++i) { if (lens[i] < 0 || lens[i] > arx::MAX_READ_LEN) { c->set_error("read length outside [0, 255]"); return ARX_E_ARG; } tot += lens[i]; } \
		  if (tot >= ((int64_t)1 << 31) - 64) { c->set_error("batch too large: more than 2^31 bases, split the batch"); return ARX_E_TOO_LARGE; } } \
		ARX_TRY(c, b->rt.bind();                                                                                                    \
			b->pipe.free_work(b->work); b->res = arx::BatchResult(); b->done_stage = 0; b->rfa_marked = false;                      \
			b->rfa = arx::RfaResult(); b->post = arx::PostResult();                                                                 \
			b->pipe.upload_into(b->db, bases, lens, n_reads); b->lens_host.assign(lens, lens + n_reads);)                           \
		return ARX_OK;                                                                                                              \
	}                                                                                                                               \
	int arx_batch_reset_device(arx_ctx *h, arx_batch *bh, int32_t n_reads, int64_t n_bases, const uint8_t *d_bases, const int32_t *d_lens) \
	{                                                                                                                               \
		Ctx *c = (Ctx *)h; Bat *b = (Bat *)bh;                                                                                      \
		if (n_reads <= 0 || (n_reads & 1)) { c->set_error("n_reads must be positive and even (read 2i/2i+1 are mates)"); return ARX_E_ARG; } \
		if (n_bases < 0 || n_bases >= ((int64_t)1 << 31) - 64) { c->set_error("batch too large: more than 2^31 bases, split the batch"); return ARX_E_TOO_LARGE; } \
		bool ok = false;                                                                                                            \
		ARX_TRY(c, b->rt.bind();                                                                                                    \
			b->pipe.free_work(b->work); b->res = arx::BatchResult(); b->done_stage = 0; b->rfa_marked = false;                      \
			b->rfa = arx::RfaResult(); b->post = arx::PostResult();                                                                 \
			ok = b->pipe.upload_from_device(b->db, d_bases, d_lens, n_reads, n_bases, 0, b->lens_host);)                            \
		if (!ok) { c->set_error("device batch: the read lengths do not add up to n_bases, or a length is outside [0, 255]"); return ARX_E_ARG; } \
		return ARX_OK;                                                                                                              \
	}                                                                                                                               \
	int arx_batch_device_view(arx_ctx *h, arx_batch *bh, arx_device_view *v)                                                        \
	{                                                                                                                               \
		Ctx *c = (Ctx *)h; Bat *b = (Bat *)bh;                                                                                      \
		if (!b->work.alns) { c->set_error("arx_batch_device_view before arx_batch_run(ARX_STAGE_ALN)"); return ARX_E_ARG; }         \
		ARX_TRY(c, b->rt.bind(); b->rt.sync();)                                                                                     \
		v->n_reads = b->db.n_reads; v->n_regs = b->work.c_n_regs; v->n_cigar = b->work.c_n_cig;                                     \
		v->reg_off = b->work.c_reg_off; v->regs = (const arx_reg *)b->work.c_regs; v->alns = (const arx_aln *)b->work.c_alns; v->cigars = b->work.c_cig; \
		const bool placed = b->rfa_marked && !b->rfa.cand_off.empty();                                                              \
		v->n_cands = placed ? b->rfa.n_cands : 0; v->cand_off = placed ? b->rfa.d_cand_off : nullptr; v->cands = placed ? (const arx_cand *)b->rfa.d_cands : nullptr; \
		return ARX_OK;                                                                                                              \
	}                                                                                                                               \
	int arx_batch_detach(arx_ctx *h, arx_batch *bh, int64_t *sizes)                                                                 \
	{                                                                                                                               \
		Ctx *c = (Ctx *)h; Bat *b = (Bat *)bh;                                                                                      \
		if (!b->work.alns) { c->set_error("arx_batch_detach before arx_batch_run(ARX_STAGE_ALN)"); return ARX_E_ARG; }              \
		ARX_TRY(c, b->rt.bind();                                                                                                    \
			auto &d = b->det; d.valid = false;                                                                                      \
			const bool placed = b->rfa_marked && !b->rfa.cand_off.empty();                                                          \
			d.n_reads = b->db.n_reads; d.n_regs = b->work.c_n_regs; d.n_cig = b->work.c_n_cig; d.n_cands = placed ? b->rfa.n_cands : 0; \
			auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };                                                            \
			size_t at = 0;                                                                                                          \
			d.o_reg_off = at; at += al(4 * ((size_t)d.n_reads + 1));                                                                \
			d.o_regs = at; at += al(sizeof(arx::Reg) * (size_t)d.n_regs);                                                           \
			d.o_alns = at; at += al(sizeof(arx::Aln) * (size_t)d.n_regs);                                                           \
			d.o_cig = at; at += al(4 * (size_t)d.n_cig);                                                                            \
			d.o_cands = at; at += al(sizeof(arx::Cand) * (size_t)d.n_cands);                                                        \
			if (at > b->slab_cap) { b->rt.sync(); if (b->slab) b->rt.pfree(b->slab); b->slab_cap = at + at / 4 + 4096; b->slab = b->rt.template palloc<char>(b->slab_cap); } \
			b->rt.d2d(b->slab + d.o_reg_off, b->work.c_reg_off, 4 * ((size_t)d.n_reads + 1));                                       \
			b->rt.d2d(b->slab + d.o_regs, b->work.c_regs, sizeof(arx::Reg) * (size_t)d.n_regs);                                     \
			b->rt.d2d(b->slab + d.o_alns, b->work.c_alns, sizeof(arx::Aln) * (size_t)d.n_regs);                                     \
			b->rt.d2d(b->slab + d.o_cig, b->work.c_cig, 4 * (size_t)d.n_cig);                                                       \
			if (placed) { b->rt.d2d(b->slab + d.o_cands, b->rfa.d_cands, sizeof(arx::Cand) * (size_t)d.n_cands); d.cand_off = b->rfa.cand_off; } \
			else d.cand_off.clear();                                                                                                \
			b->rt.sync();                                                                                                           \
			d.valid = true;                                                                                                         \
			if (sizes) { sizes[0] = d.n_reads; sizes[1] = d.n_regs; sizes[2] = d.n_cig; sizes[3] = d.n_cands; })                    \
		return ARX_OK;                                                                                                              \
	}                                                                                                                               \
	int arx_batch_fetch_detached(arx_ctx *h, arx_batch *bh, int32_t *reg_off, arx_reg *regs, arx_aln *alns, uint32_t *cigars, int32_t *cand_off, arx_cand *cands) \
	{                                                                                                                               \
		Ctx *c = (Ctx *)h; Bat *b = (Bat *)bh;                                                                                      \
		if (!b->det.valid) { c->set_error("arx_batch_fetch_detached before arx_batch_detach"); return ARX_E_ARG; }                  \
		ARX_TRY(c,                                                                                                                  \
			if (!b->copy_ready) { std::string e = b->rt_copy.init(c->device); if (!e.empty()) throw std::runtime_error(e); b->copy_ready = true; } \
			b->rt_copy.bind();                                                                                                      \
			const auto &d = b->det;                                                                                                 \
			if (reg_off) b->rt_copy.d2h_async(reg_off, b->slab + d.o_reg_off, 4 * ((size_t)d.n_reads + 1));                         \
			if (regs) b->rt_copy.d2h_async(regs, b->slab + d.o_regs, sizeof(arx::Reg) * (size_t)d.n_regs);                          \
			if (alns) b->rt_copy.d2h_async(alns, b->slab + d.o_alns, sizeof(arx::Aln) * (size_t)d.n_regs);                          \
			if (cigars) b->rt_copy.d2h_async(cigars, b->slab + d.o_cig, 4 * (size_t)d.n_cig);                                       \
			if (cands && d.n_cands) b->rt_copy.d2h_async(cands, b->slab + d.o_cands, sizeof(arx::Cand) * (size_t)d.n_cands);        \
			if (cand_off && !d.cand_off.empty()) memcpy(cand_off, d.cand_off.data(), 4 * ((size_t)d.n_reads + 1));                  \
			b->rt_copy.sync();)                                                                                                     \
		return ARX_OK;                                                                                                              \
	}                                                                                                                               \
	int arx_batch_run(arx_ctx *h, arx_batch *bh, int32_t last_stage)                                                                \
	{                                                                                                                               \
		Ctx *c = (Ctx *)h; Bat *b = (Bat *)bh;                                                                                      \
		ARX_TRY(c,                                                                                                                  \
			b->rt.bind();                                                                                                           \
			b->rt.set_timing(c->timing);                                                                                            \
			/* stages already done are kept (run(SEED) then run(ALN) resumes); asking for a stage again restarts the batch */       \
			if (last_stage <= b->done_stage) { b->pipe.free_work(b->work); b->res = arx::BatchResult(); b->done_stage = 0; }        \
			b->rfa_marked = false; b->post = arx::PostResult();                                                                          \
			if (b->done_stage < ARX_STAGE_SEED) {                                                                                   \
				int rc = b->pipe.stage_seed(b->db, b->work);                                                                        \
				if (rc == -2) { c->set_error("batch too large: seed occurrences exceed 2^30, split the batch"); return ARX_E_TOO_LARGE; } \
				b->res.n_occ = b->work.T;                                                                                           \
			}                                                                                                                       \
			if (b->done_stage < ARX_STAGE_CHAIN && last_stage >= ARX_STAGE_CHAIN) b->pipe.stage_chain(b->db, b->work);              \
			if (b->done_stage < ARX_STAGE_EXTEND && last_stage >= ARX_STAGE_EXTEND) b->pipe.stage_extend(b->db, b->work, b->res);   \
			if (b->done_stage < ARX_STAGE_RESCUE && last_stage >= ARX_STAGE_RESCUE) b->pipe.stage_rescue(b->db, b->work, b->res);   \
			uint32_t e = last_stage >= ARX_STAGE_ALN ? (uint32_t)b->pipe.stage_reg2aln(b->db, b->work) : b->pipe.read_err(b->work); \
			b->done_stage = last_stage;                                                                                             \
			b->rt.sync();                                                                                                           \
			if (e) { c->set_error("device stage raised error bits " + std::to_string(e)); return ARX_E_DEVICE; }                    \
		)                                                                                                                           \
		return ARX_OK;                                                                                                              \
	}                                                                                                                               \
	int arx_batch_counts(arx_ctx *h, arx_batch *bh, int64_t *c8)                                                                    \
	{                                                                                                                               \
		Ctx *c = (Ctx *)h; Bat *b = (Bat *)bh;                                                                                      \
		ARX_TRY(c,                                                                                                                  \
			b->rt.bind();                                                                                                           \
		)                                                                                                                           \
		c8[0] = b->db.n_reads; c8[1] = b->work.c_regs ? b->work.c_n_regs : 0; c8[2] = b->work.c_regs ? b->work.c_n_cig : 0; c8[3] = b->res.n_occ; \
		c8[4] = b->res.ext_rounds; c8[5] = b->res.n_ext_tasks; c8[6] = b->res.rescue_rounds; c8[7] = b->res.n_sw_tasks;             \
		return ARX_OK;                                                                                                              \
	}                                                                                                                               \
	int arx_batch_fetch(arx_ctx *h, arx_batch *bh, int32_t *reg_off, arx_reg *regs, arx_aln *alns, uint32_t *cigars)                \
	{                                                                                                                               \
		Ctx *c = (Ctx *)h; Bat *b = (Bat *)bh;                                                                                      \
		if (!b->work.alns) { c->set_error("arx_batch_fetch before arx_batch_run(ARX_STAGE_ALN)"); return ARX_E_ARG; }               \
		static_assert(sizeof(arx_reg) == sizeof(arx::Reg) && sizeof(arx_aln) == sizeof(arx::Aln), "C-ABI structs must mirror the device structs"); \
		ARX_TRY(c, b->rt.bind(); b->pipe.fetch(b->db, b->work, reg_off, (arx::Reg *)regs, (arx::Aln *)alns, cigars);)                \
		return ARX_OK;                                                                                                              \
	}                                                                                                                               \
	int arx_batch_rfa(arx_ctx *h, arx_batch *bh, int32_t n_barcodes, const int64_t *bc_pair_off, const uint8_t *do_rfa, double penalty_f, \
	                  const int64_t *cen_start, const int64_t *cen_end, int64_t *n_cands)                                           \
	{                                                                                                                               \
		Ctx *c = (Ctx *)h; Bat *b = (Bat *)bh;                                                                                      \
		if (!(penalty_f == (double)(int32_t)penalty_f) || penalty_f < -1000000.0 || penalty_f > 1000000.0) {                        \
			c->set_error("improper-pair penalty (-i) must be an integer: placement scores are kept in exact half-units (penalty / 2 enters " \
			             "fastScore, aligner.go:1140-1198), and a non-integer value would make the result depend on Go's float summation order"); \
			return ARX_E_ARG;                                                                                                       \
		}                                                                                                                           \
		const int32_t penalty = (int32_t)penalty_f;                                                                                 \
		if (!b->work.alns) { c->set_error("arx_batch_rfa before arx_batch_run(ARX_STAGE_ALN)"); return ARX_E_ARG; }                 \
		if (n_barcodes <= 0 || bc_pair_off[0] != 0 || 2 * bc_pair_off[n_barcodes] != b->db.n_reads) { c->set_error("barcode offsets must cover the batch"); return ARX_E_ARG; } \
		for (int i = 0; i < n_barcodes; ++i) if (bc_pair_off[i + 1] < bc_pair_off[i]) { c->set_error("barcode offsets must not decrease"); return ARX_E_ARG; } \
		ARX_TRY(c, b->rt.bind(); b->rt.set_timing(c->timing);                                                                       \
			if (b->rfa_marked) b->rt.arena_rewind(b->rfa_mark); else { b->rfa_mark = b->rt.arena_mark(); b->rfa_marked = true; }    \
			arx::RfaStage<RT>::run(b->pipe, b->db, b->work, n_barcodes, bc_pair_off, do_rfa, penalty, cen_start, cen_end, b->lens_host.data(), b->rfa); \
			b->post = arx::PostResult(); b->post_mark = b->rt.arena_mark();                                                         \
			*n_cands = b->rfa.n_cands;)                                                                               \
		return ARX_OK;                                                                                                              \
	}                                                                                                                               \
	int arx_batch_post(arx_ctx *h, arx_batch *bh, int64_t *n_mm)                                                                    \
	{                                                                                                                               \
		Ctx *c = (Ctx *)h; Bat *b = (Bat *)bh;                                                                                      \
		if (b->rfa.cand_off.empty() || !b->rfa_marked) { c->set_error("arx_batch_post before arx_batch_rfa"); return ARX_E_ARG; }   \
		static_assert(sizeof(arx_cand_post) == sizeof(arx::CandPost) && sizeof(arx_split) == sizeof(arx::SplitRec), "C-ABI structs must mirror the device structs"); \
		ARX_TRY(c, b->rt.bind(); b->rt.set_timing(c->timing);                                                                       \
			b->rt.arena_rewind(b->post_mark);                                                                                       \
			arx::PostStage<RT>::run(b->pipe, b->db, b->work, b->rfa, b->post);                                                      \
			b->rt.sync();                                                                                                           \
			*n_mm = b->post.n_mm;)                                                                                                  \
		return ARX_OK;                                                                                                              \
	}                                                                                                                               \
	int arx_batch_post_fetch(arx_ctx *h, arx_batch *bh, arx_cand_post *post, arx_split *split, int32_t *mm_ref, int32_t *mm_read)   \
	{                                                                                                                               \
		Ctx *c = (Ctx *)h; Bat *b = (Bat *)bh;                                                                                      \
		if (!b->post.done || !b->rfa_marked) { c->set_error("arx_batch_post_fetch before arx_batch_post"); return ARX_E_ARG; }      \
		ARX_TRY(c, b->rt.bind(); arx::PostStage<RT>::fetch(b->pipe, b->db, b->rfa, b->post, (arx::CandPost *)post, (arx::SplitRec *)split, mm_ref, mm_read);) \
		return ARX_OK;                                                                                                              \
	}                                                                                                                               \
	int arx_batch_rfa_fetch(arx_ctx *h, arx_batch *bh, int32_t *cand_off, arx_cand *cands)                                          \
	{                                                                                                                               \
		Ctx *c = (Ctx *)h; Bat *b = (Bat *)bh;                                                                                      \
		if (b->rfa.cand_off.empty() || !b->rfa_marked) { c->set_error("arx_batch_rfa_fetch before arx_batch_rfa"); return ARX_E_ARG; } \
		static_assert(sizeof(arx_cand) == sizeof(arx::Cand), "C-ABI structs must mirror the device structs");                       \
		ARX_TRY(c, b->rt.bind(); arx::RfaStage<RT>::fetch(b->pipe, b->db, b->rfa, cand_off, (arx::Cand *)cands);)                    \
		return ARX_OK;                                                                                                              \
	}                                                                                                                               \
	int arx_batch_debug_intv(arx_ctx *h, arx_batch *bh, int32_t *n_intv, uint64_t *intv4)                                           \
	{                                                                                                                               \
		Ctx *c = (Ctx *)h; Bat *b = (Bat *)bh;                                                                                      \
		if (!b->work.intv) return ARX_E_ARG;                                                                                        \
		ARX_TRY(c, b->rt.bind();                                                                                                    \
			b->rt.d2h(n_intv, b->work.n_intv, 4 * (size_t)b->db.n_reads);                                                           \
			b->rt.d2h(intv4, b->work.intv, sizeof(arx::Biv) * (size_t)b->db.n_reads * arx::CAP_INTV);)                              \
		return ARX_OK;                                                                                                              \
	}                                                                                                                               \
	int arx_batch_debug_chains(arx_ctx *h, arx_batch *bh, int32_t *occ_off, int32_t *n_chain, arx_chain *chains, arx_seed *seeds)   \
	{                                                                                                                               \
		Ctx *c = (Ctx *)h; Bat *b = (Bat *)bh;                                                                                      \
		if (!b->work.cout) return ARX_E_ARG;                                                                                        \
		static_assert(sizeof(arx_chain) == sizeof(arx::Chain) && sizeof(arx_seed) == sizeof(arx::Seed), "C-ABI structs must mirror the device structs"); \
		ARX_TRY(c, b->rt.bind();                                                                                                    \
			b->rt.d2h(occ_off, b->work.occ_off, 4 * ((size_t)b->db.n_reads + 1));                                                   \
			b->rt.d2h(n_chain, b->work.n_chain, 4 * (size_t)b->db.n_reads);                                                         \
			b->rt.d2h(chains, b->work.cout, sizeof(arx::Chain) * (size_t)b->work.T);                                                \
			b->rt.d2h(seeds, b->work.sout, sizeof(arx::Seed) * (size_t)b->work.T);)                                                 \
		return ARX_OK;                                                                                                              \
	}                                                                                                                               \
	int arx_batch_debug_core(arx_ctx *h, arx_batch *bh, int32_t *n_core, arx_reg *regs)                                             \
	{                                                                                                                               \
		Ctx *c = (Ctx *)h; Bat *b = (Bat *)bh;                                                                                      \
		if (!b->work.n_core) return ARX_E_ARG;                                                                                      \
		ARX_TRY(c, b->rt.bind();                                                                                                    \
			b->rt.d2h(n_core, b->work.n_core, 4 * (size_t)b->db.n_reads);                                                           \
			b->rt.d2h(regs, b->work.regs, sizeof(arx::Reg) * (size_t)b->work.T);)                                                   \
		return ARX_OK;                                                                                                              \
	}                                                                                                                               \
	void arx_batch_free(arx_ctx *, arx_batch *bh) { try { delete (Bat *)bh; } catch (...) {} }                                      \
	int arx_kernel_times(arx_ctx *h, int32_t cap, char *names, int32_t name_w, double *ms, int64_t *calls, int64_t *items)          \
	{                                                                                                                               \
		Ctx *c = (Ctx *)h;                                                                                                          \
		std::map<std::string, arx::KernelTimer> all;                                                                                \
		try {                                                                                                                       \
			std::lock_guard<std::mutex> g(c->mu);                                                                                   \
			all = c->tm_done;                                                                                                       \
			for (Bat *b : c->live) { b->rt.bind(); for (auto &kv : b->rt.timers()) { arx::KernelTimer &t = all[kv.first]; t.ms += kv.second.ms; t.calls += kv.second.calls; t.items += kv.second.items; } } \
		} catch (...) { return 0; }                                                                                                 \
		int n = 0;                                                                                                                  \
		for (auto &kv : all) {                                                                                                      \
			if (n >= cap) break;                                                                                                    \
			snprintf(names + (size_t)n * name_w, name_w, "%s", kv.first.c_str());                                                   \
			ms[n] = kv.second.ms; calls[n] = kv.second.calls; items[n] = kv.second.items; ++n;                                      \
		}                                                                                                                           \
		return n;                                                                                                                   \
	}                                                                                                                               \
	void arx_kernel_times_reset(arx_ctx *h, int32_t enable)                                                                         \
	{                                                                                                                               \
		Ctx *c = (Ctx *)h;                                                                                                          \
		try {                                                                                                                       \
			std::lock_guard<std::mutex> g(c->mu);                                                                                   \
			c->tm_done.clear(); c->timing = enable != 0;                                                                            \
			for (Bat *b : c->live) { b->rt.bind(); b->rt.timers_reset(enable != 0); }                                               \
		} catch (...) {}                                                                                                            \
	}                                                                                                                               \
	}
