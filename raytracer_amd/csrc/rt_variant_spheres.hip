// rt_variant_spheres.hip — the trace kernels (rt_kernels.h) instantiated for the "spheres" scene content; one translation unit per
// variant so that `make -j` compiles them side by side.
#include "rt_kernels.h"

namespace rtd {
const KernelSet ks_spheres = make_kernel_set<CfgSpheres, true>("spheres");
} // namespace rtd
