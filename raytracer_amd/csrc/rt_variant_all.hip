// rt_variant_all.hip — the trace kernels (rt_kernels.h) instantiated for the "generic" scene content; one translation unit per
// variant so that `make -j` compiles them side by side.
#include "rt_kernels.h"

namespace rtd {
const KernelSet ks_all = make_kernel_set<CfgAll, false>("generic");
void (*const k_shade_vis)(SceneDev, RenderDev, uint32_t) = k_shade<CfgAll, true>;
void (*const k_shade_validate)(SceneDev, RenderDev, uint32_t) = k_shade<CfgAll, false, true>;
void (*const k_shade_vis_validate)(SceneDev, RenderDev, uint32_t) = k_shade<CfgAll, true, true>;
} // namespace rtd
