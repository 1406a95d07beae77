// The fp64 OBJ-loader wave kernels (scene class 3: render_wave_kernel / render_wave_sorted_kernel and their one-wave twins), in a translation
// unit of their own so that they can be SCHEDULED differently from the rest: with LLVM's max-ILP strategy (csrc/Makefile) this variant --
// latency-bound, four waves per SIMD -- runs 2.5 % faster (mesh 4K 128 spp 139.6 -> 136.2 ms), while the sphere-scene variants lose 6-12 %
// under it and the fp32 ones do not care.  Same flags otherwise (-ffp-contract=off: the reference's arithmetic).
#include "rt_launch.h"
#include "rt_wave_launch.h"

namespace rt {

hipError_t launch_wave_obj(bool count, dim3 grid, size_t lds, hipStream_t s, const Scene<double>& sc, const RenderArgs& a, double* accum, unsigned long long* counters, bool one_wave)
{
    if (a.quantised && sc.sorted_visit) return launch_wave_sorted<double, 3>(count, grid, lds, s, sc, a, accum, counters, one_wave);
    if (a.quantised) return launch_wave<double, true, 3>(count, grid, lds, s, sc, a, accum, counters, one_wave);
    return launch_wave<double, false, 3>(count, grid, lds, s, sc, a, accum, counters, one_wave);
}

}  // namespace rt
