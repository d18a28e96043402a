// arx_bam.cpp -- C ABI of the BAM sink (bam_sink.h); host code only.
#include <stdio.h>
#include "bam_sink.h"
#include "bam_records.h"

extern "C" {

int arx_bam_open(const char *path, int32_t n_contigs, const char *const *names, const int32_t *lens, const char *extra_header, int32_t threads, int32_t level,
                 arx_bam **out, char *msg, int32_t msg_cap)
{
	*out = nullptr;
	arx::BamSink *w = nullptr;
	try {
		w = new arx::BamSink();
		if (!w->open(path, n_contigs, names, lens, extra_header, threads, level)) {
			if (msg && msg_cap > 0) snprintf(msg, (size_t)msg_cap, "%s", w->error.c_str());
			delete w;
			return ARX_E_IO;
		}
	} catch (const std::exception &e) {
		if (msg && msg_cap > 0) snprintf(msg, (size_t)msg_cap, "%s", e.what());
		delete w;
		return ARX_E_IO;
	}
	*out = (arx_bam *)w;
	return ARX_OK;
}

int arx_bam_write(arx_bam *h, const arx_bam_batch *batch)
{
	arx::BamSink *w = (arx::BamSink *)h;
	try {
		return w->write(*batch) ? ARX_OK : ARX_E_IO;
	} catch (const std::exception &e) {
		w->error = e.what();
		return ARX_E_IO;
	}
}

int arx_bam_close(arx_bam *h, int64_t *stats)
{
	arx::BamSink *w = (arx::BamSink *)h;
	bool ok = false;
	try { ok = w->close(); } catch (const std::exception &) {}
	if (stats) { stats[0] = w->n_records; stats[1] = w->n_blocks; stats[2] = w->bytes_in; stats[3] = w->bytes_out + 28; }
	delete w;
	return ok ? ARX_OK : ARX_E_IO;
}

const char *arx_bam_error(arx_bam *h) { return ((arx::BamSink *)h)->error.c_str(); }

struct RecBufHandle { arx::RecBuf buf; std::string error; };
int arx_recbuf_create(arx_recbuf **out)
{
	try { *out = (arx_recbuf *)new RecBufHandle(); } catch (const std::exception &) { *out = nullptr; return ARX_E_IO; }
	return ARX_OK;
}
int arx_recbuf_build(arx_recbuf *h, const arx_super_batch *sb, const int32_t *cand_off, const arx_cand *cands, const arx_aln *alns, const uint32_t *cigars,
                     const arx_cand_post *post, int32_t threads, arx_bam_batch *view)
{
	RecBufHandle *rb = (RecBufHandle *)h;
	if (!sb || !cand_off || !cands || !alns || !cigars || !view) { rb->error = "null argument"; return ARX_E_ARG; }
	try {
		return rb->buf.build(*sb, cand_off, cands, alns, cigars, post, threads, view, rb->error) ? ARX_OK : ARX_E_ARG;
	} catch (const std::exception &e) { rb->error = e.what(); return ARX_E_IO; }
}
const char *arx_recbuf_error(arx_recbuf *h) { return ((RecBufHandle *)h)->error.c_str(); }
void arx_recbuf_free(arx_recbuf *h) { delete (RecBufHandle *)h; }

}
