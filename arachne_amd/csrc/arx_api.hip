// arx_api.hip -- libarachne_amd.so: the C-ABI of include/arachne_amd.h on the HIP runtime (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC arx_api.hip -o libarachne_amd.so
#include "hip_rt.h"
#include "api_impl.h"

ARX_DEFINE_C_API(arx::HipRT)
