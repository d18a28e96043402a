// arx_cold.hip -- the two list-bookkeeping kernels (region de-duplication, rescue state machine), compiled at -O1.
// hipcc 7.2 (AMD clang 22) at -O2/-O3 generates a dedup kernel for gfx950 that never terminates although the same
// source is correct at -O1, on the host (clang -O3, gcc -O2) and under ASan/UBSan; these kernels are far from any
// hot spot, so they live in their own translation unit until the miscompile is understood.
#include "hip_rt.h"
#include "pipeline.h"

namespace arx {
template <class F> struct ColdUsesSlots { static const bool value = true; };
template <> struct ColdUsesSlots<KRescueStep> { static const bool value = false; }; // no per-slot scratch: may take one item per lane
template <class F> void HipRT::launch_cold(const char *nm, int n, const F &f) { launch_cold_impl(nm, n, f, !ColdUsesSlots<F>::value); }
template void HipRT::launch_cold<KDedup>(const char *, int, const KDedup &);
template void HipRT::launch_cold<KRescueStep>(const char *, int, const KRescueStep &);
} // namespace arx
