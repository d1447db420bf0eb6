// rt_variant_mesh.hip — the trace kernels (rt_kernels.h) instantiated for the "mesh" scene content; one translation unit per
// variant so that `make -j` compiles them side by side.
#include "rt_kernels.h"

namespace rtd {
const KernelSet ks_mesh = make_kernel_set<CfgMesh, false>("mesh");
} // namespace rtd
