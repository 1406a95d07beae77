// raytracing_amd/csrc/rt_kernels.inl -- launcher bodies, included with RT_REAL defined.
#include "rt_launch.h"

namespace rt {

hipError_t launch_render(const Scene<RT_REAL>& sc, const RenderArgs& a, RT_REAL* accum, unsigned long long* counters, hipStream_t s)
{
    const int tiles_x = (a.width + 15) / 16, tiles_y = (a.y_end - a.y_begin + 15) / 16;
    if (tiles_x <= 0 || tiles_y <= 0) return hipSuccess;
    dim3 grid((unsigned) (tiles_x * tiles_y)), block(256);
    const size_t lds = (size_t) a.lds_ops_bytes + a.lds_prims_bytes;
    const bool count = (a.flags & RT_FLAG_COUNT_TESTS) != 0;
    if (lds > 0) {
        static thread_local size_t configured = 0;
        if (lds > configured) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel<RT_REAL, true, false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel<RT_REAL, true, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e != hipSuccess) return e;
            configured = lds;
        }
        if (count) hipLaunchKernelGGL((render_kernel<RT_REAL, true, true>), grid, block, lds, s, sc, a, accum, counters);
        else hipLaunchKernelGGL((render_kernel<RT_REAL, true, false>), grid, block, lds, s, sc, a, accum, counters);
    }
    else {
        if (count) hipLaunchKernelGGL((render_kernel<RT_REAL, false, true>), grid, block, 0, s, sc, a, accum, counters);
        else hipLaunchKernelGGL((render_kernel<RT_REAL, false, false>), grid, block, 0, s, sc, a, accum, counters);
    }
    return hipGetLastError();
}

#ifdef RT_WITH_FIRST_HIT
hipError_t launch_first_hit(const Scene<double>& sc, const double* rays, int n, int32_t* prim, double* t, int32_t* front,
                            uint32_t* n_box, uint32_t* n_leaf, unsigned long long* hash, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL((first_hit_kernel<double>), dim3((n + 255) / 256), dim3(256), 0, s, sc, rays, n, prim, t, front, n_box, n_leaf, hash);
    return hipGetLastError();
}
#endif

}  // namespace rt
