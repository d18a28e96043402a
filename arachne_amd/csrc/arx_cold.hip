// arx_cold.hip -- the two list-bookkeeping kernels (region de-duplication, rescue state machine) as their own translation unit
// so that the library's units compile side by side.  Until round 2 this unit was built at -O1: hipcc 7.2 at -O2/-O3 emitted
// a dedup kernel that never terminated on gfx950.  The cause was bisected to the optimised body of ks_introsort (arx_dev.h has the
// record and the source form that terminates); the unit is -O3 like the rest.
#include "hip_rt.h"
#include "pipeline.h"

namespace arx {
template <class F> struct ColdUsesSlots { static const bool value = true; };
template <> struct ColdUsesSlots<KRescueStep> { static const bool value = false; }; // no per-slot scratch: may take one item per lane
template <class F> void HipRT::launch_cold(const char *nm, int n, const F &f) { launch_cold_impl(nm, n, f, !ColdUsesSlots<F>::value); }
template void HipRT::launch_cold<KDedup>(const char *, int, const KDedup &);
template void HipRT::launch_cold<KRescueStep>(const char *, int, const KRescueStep &);
} // namespace arx
