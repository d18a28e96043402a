// arx_index.hip -- the device half of arx_index_build (hip_index_build.h) as its own translation unit of libarachne_amd.so.
#include "hip_index_build.h"
#include "index_build.h"

namespace arx {
// BWT + sampled SA of arx_index_build: in HBM when a device is visible (any genome size), otherwise the host's induced
// sorting (2 * l_pac < 2^31).  ARX_INDEX_HOST=1 forces the host path, ARX_INDEX_DEVICE=<k> picks the device.
std::string product_bwt_sa(const uint8_t *pac, size_t pac_bytes, int64_t l_pac, const uint64_t cnt_fwd[4], const std::string &prefix)
{
	int ndev = 0;
	const bool force_host = getenv("ARX_INDEX_HOST") && atoi(getenv("ARX_INDEX_HOST")) != 0;
	if (force_host || hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return build_bwt_sa_host(pac, pac_bytes, l_pac, cnt_fwd, prefix);
	return gpuidx::build_bwt_sa_device(pac, pac_bytes, l_pac, cnt_fwd, prefix, getenv("ARX_INDEX_DEVICE") ? atoi(getenv("ARX_INDEX_DEVICE")) : -1);
}
} // namespace arx
