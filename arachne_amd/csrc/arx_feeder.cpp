// arx_feeder.cpp -- C ABI of the FASTQ feeder (feeder.h); host code only.
#include <stdio.h>
#include "feeder.h"

extern "C" {

int arx_feeder_open(const char *r1_path, const char *r2_path, arx_feeder **out, char *msg, int32_t msg_cap)
{
	*out = nullptr;
	arx::Feeder *f = nullptr;
	try {
		f = new arx::Feeder();
		if (!f->open(r1_path, r2_path)) {
			if (msg && msg_cap > 0) snprintf(msg, (size_t)msg_cap, "%s", f->error.c_str());
			delete f;
			return ARX_E_IO;
		}
	} catch (const std::exception &e) {
		if (msg && msg_cap > 0) snprintf(msg, (size_t)msg_cap, "%s", e.what());
		delete f;
		return ARX_E_IO;
	}
	*out = (arx_feeder *)f;
	return ARX_OK;
}

int arx_feeder_next(arx_feeder *h, int64_t target_pairs, arx_super_batch *out)
{
	try {
		return ((arx::Feeder *)h)->next(target_pairs, out);
	} catch (const std::exception &) {
		return ARX_E_IO;
	}
}

void arx_feeder_close(arx_feeder *h) { delete (arx::Feeder *)h; }

}
