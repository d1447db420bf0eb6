// rt_variant_boxes.hip — the trace kernels (rt_kernels.h) instantiated for the "boxes" scene content; one translation unit per
// variant so that `make -j` compiles them side by side.
#include "rt_kernels.h"

namespace rtd {
const KernelSet ks_boxes = make_kernel_set<CfgBoxes, false>("boxes");
} // namespace rtd
