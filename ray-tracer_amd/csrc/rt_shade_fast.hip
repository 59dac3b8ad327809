// The shade stage in the fast arithmetic mode (see rt_device_math.h): the same source as the exact instances in the
// library's main translation unit, compiled here with hardware reciprocal / square root and FMA contraction.
#define RT_FAST_ARITH 1
#pragma clang fp contract(fast)

#include "rt_shade.h"

namespace rtint {

void launch_shade_fast(bool lds_tables, bool probe, uint32_t grid, hipStream_t st, const rtdev::SceneDev& sc, const rtdev::PathDev& pd,
                       const rtdev::ShadeParams& sp) {
    using namespace rtdev;
    const dim3 g(grid), b(kBlock);
    if (probe) {
        if (lds_tables) hipLaunchKernelGGL((k_shade<true, true>), g, b, 0, st, sc, pd, sp);
        else hipLaunchKernelGGL((k_shade<false, true>), g, b, 0, st, sc, pd, sp);
    } else {
        if (lds_tables) hipLaunchKernelGGL((k_shade<true, false>), g, b, 0, st, sc, pd, sp);
        else hipLaunchKernelGGL((k_shade<false, false>), g, b, 0, st, sc, pd, sp);
    }
}

void launch_bsdf_fast(uint32_t grid, uint32_t block, hipStream_t st, const rtdev::BsdfIo& io, uint32_t n, int do_sample) {
    hipLaunchKernelGGL(rtdev::k_bsdf, dim3(grid), dim3(block), 0, st, io, n, do_sample);
}

}  // namespace rtint
