// raytracing_amd/csrc/rt_kernels.inl -- launcher bodies, included with RT_REAL defined.
#include "rt_launch.h"

namespace rt {

hipError_t launch_render(const Scene<RT_REAL>& sc, const RenderArgs& a_in, RT_REAL* accum, unsigned long long* counters, hipStream_t s)
{
    RenderArgs a = a_in;
    const int tiles_x = (a.width + 15) / 16, tiles_y = (a.y_end - a.y_begin + 15) / 16;
    if (tiles_x <= 0 || tiles_y <= 0) return hipSuccess;
    dim3 grid((unsigned) (tiles_x * tiles_y)), block(256);
    const bool count = (a.flags & RT_FLAG_COUNT_TESTS) != 0;
    // dynamic LDS = [traversal region: fast-mode stack | staged program | nothing] + [fp32 modes: tile sums + item counter]
    const size_t tile_bytes = sizeof(RT_REAL) == sizeof(float) ? 768 * sizeof(float) + 16 : 0;
    size_t region = 0;
    int trav = 0;
#ifdef RT_WITH_FAST_BVH
    if (a.bvh_stack_levels > 0) { trav = 2; region = (size_t) a.bvh_stack_levels * 256 * sizeof(uint32_t); }
#endif
    if (trav == 0 && a.lds_ops_bytes + a.lds_prims_bytes > 0) { trav = 1; region = (size_t) a.lds_ops_bytes + a.lds_prims_bytes; }
    a.lds_tile_offset = (uint32_t) region;
    const size_t lds = region + tile_bytes;
    if (lds > 48 * 1024) {
        static thread_local size_t configured = 0;
        if (lds > configured) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel<RT_REAL, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel<RT_REAL, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
#ifdef RT_WITH_FAST_BVH
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel<RT_REAL, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel<RT_REAL, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
#endif
            if (e != hipSuccess) return e;
            configured = lds;
        }
    }
#ifdef RT_WITH_FAST_BVH
    if (trav == 2) {
        if (count) hipLaunchKernelGGL((render_kernel<RT_REAL, 2, true>), grid, block, lds, s, sc, a, accum, counters);
        else hipLaunchKernelGGL((render_kernel<RT_REAL, 2, false>), grid, block, lds, s, sc, a, accum, counters);
        return hipGetLastError();
    }
#endif
    if (trav == 1) {
        if (count) hipLaunchKernelGGL((render_kernel<RT_REAL, 1, true>), grid, block, lds, s, sc, a, accum, counters);
        else hipLaunchKernelGGL((render_kernel<RT_REAL, 1, false>), grid, block, lds, s, sc, a, accum, counters);
    }
    else {
        if (count) hipLaunchKernelGGL((render_kernel<RT_REAL, 0, true>), grid, block, lds, s, sc, a, accum, counters);
        else hipLaunchKernelGGL((render_kernel<RT_REAL, 0, false>), grid, block, lds, s, sc, a, accum, counters);
    }
    return hipGetLastError();
}

#ifdef RT_WITH_FAST_BVH
hipError_t launch_first_hit_fast(const Scene<float>& sc, const double* rays, int n, int32_t* prim, float* t, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    const size_t stack_bytes = (size_t) (sc.bvh_depth + 2) * 256 * sizeof(uint32_t);
    hipLaunchKernelGGL(first_hit_fast_kernel, dim3((n + 255) / 256), dim3(256), stack_bytes, s, sc, rays, n, prim, t);
    return hipGetLastError();
}
#endif

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
