// The shade stage in the fast arithmetic mode (see rt_device_math.h): the same source as the exact instances in the
// library's main translation unit, compiled here with hardware reciprocal / square root and FMA contraction.
// contract(on), not (fast): a product and a sum are fused only where the source writes them in one expression, which the
// front end decides once for every function; under (fast) the back end fuses across statements as it sees fit in each
// kernel, and the two kernels that shade vertices (k_shade, k_path_tail) would round a few operations differently.
#define RT_FAST_ARITH 1
#pragma clang fp contract(on)

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

void launch_tail_fast(bool lds_tables, bool anyhit, uint32_t grid, hipStream_t st, const rtdev::SceneDev& sc, const rtdev::PathDev& pd,
                      const rtdev::TailParams& tp) {
    using namespace rtdev;
    const dim3 g(grid), b(kTailBlock);
    if (lds_tables) {
        if (anyhit) hipLaunchKernelGGL((k_path_tail<true, true>), g, b, 0, st, sc, pd, tp);
        else hipLaunchKernelGGL((k_path_tail<true, false>), g, b, 0, st, sc, pd, tp);
    } else {
        if (anyhit) hipLaunchKernelGGL((k_path_tail<false, true>), g, b, 0, st, sc, pd, tp);
        else hipLaunchKernelGGL((k_path_tail<false, false>), g, b, 0, st, sc, pd, tp);
    }
}

void launch_bsdf_fast(uint32_t grid, uint32_t block, hipStream_t st, const rtdev::BsdfIo& io, uint32_t n, int do_sample) {
    hipLaunchKernelGGL(rtdev::k_bsdf, dim3(grid), dim3(block), 0, st, io, n, do_sample);
}

}  // namespace rtint
