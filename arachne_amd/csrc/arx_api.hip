// arx_api.hip -- libarachne_amd.so: the C-ABI of include/arachne_amd.h on the HIP runtime (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC arx_api.hip -o libarachne_amd.so
#ifdef ARX_RFA_STATS // diagnostics build (-DARX_RFA_STATS): clock ticks (100 MHz) per phase of rfa_barcode, summed over barcodes, printed per launch
#include <hip/hip_runtime.h>
__device__ unsigned long long g_rfa_t[8];
__shared__ unsigned long long rfa_t_prev;
#define ARX_RFA_T(k) do { if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); if ((k) > 0) atomicAdd(&g_rfa_t[k], t_ - rfa_t_prev); rfa_t_prev = t_; } } while (0)
#endif
#include "hip_rt.h"
#include "api_impl.h"

namespace arx { // register budgets of single thread-per-item kernels (hip_rt.h ItemsWpe), where a measurement backs them
#ifdef ARX_WPE_CHAIN
template <> struct ItemsWpe<KChain> { static constexpr int v = ARX_WPE_CHAIN; };
#endif
#ifdef ARX_WPE_EXTSTEP
template <> struct ItemsWpe<KExtStep> { static constexpr int v = ARX_WPE_EXTSTEP; };
#endif
#ifdef ARX_WPE_MAPQ
template <> struct ItemsWpe<KMapq> { static constexpr int v = ARX_WPE_MAPQ; };
#endif
#ifdef ARX_WPE_REG2ALN
template <> struct ItemsWpe<KReg2Aln> { static constexpr int v = ARX_WPE_REG2ALN; };
#endif
}

ARX_DEFINE_C_API(arx::HipRT)
