// raytracing_amd/csrc/rt_kernels.inl -- launcher bodies, included with RT_REAL defined.
#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdlib>
#include <vector>
#include "rt_launch.h"
#ifdef RT_WITH_WAVE
#include "rt_wave.h"
#include "rt_wave_launch.h"
#include "rt_queue.h"
#endif

namespace rt {

#ifdef RT_WITH_WAVE
// one variant of the decoupled kernel (rt_queue.h)
template <typename R, bool kQuant, int kFull>
hipError_t launch_queue(bool count, dim3 grid, size_t lds, hipStream_t s, const Scene<R>& sc, const RenderArgs& a, R* accum, unsigned long long* counters, void* pools, unsigned int* locks)
{
    if (lds > 48 * 1024) {
        static LdsConfigured done;
        std::atomic<size_t>* configured = done.of_current_device();
        if (!configured) return hipErrorInvalidDevice;
        if (lds > configured->load(std::memory_order_relaxed)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_queue_kernel<R, false, kQuant, kFull>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_queue_kernel<R, true, kQuant, kFull>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e != hipSuccess) return e;
            configured->store(lds, std::memory_order_relaxed);
        }
    }
    if (count) hipLaunchKernelGGL((render_queue_kernel<R, true, kQuant, kFull>), grid, dim3(256), lds, s, sc, a, accum, counters, static_cast<QSlot<R>*>(pools), locks);
    else hipLaunchKernelGGL((render_queue_kernel<R, false, kQuant, kFull>), grid, dim3(256), lds, s, sc, a, accum, counters, static_cast<QSlot<R>*>(pools), locks);
    return hipGetLastError();
}
#endif

#ifdef RT_WITH_WAVE
size_t queue_slot_bytes(RT_REAL) { return sizeof(QSlot<RT_REAL>); }
#endif

hipError_t launch_render(const Scene<RT_REAL>& sc, const RenderArgs& a_in, RT_REAL* accum, unsigned long long* counters, hipStream_t s, void* q_pools, unsigned int* q_locks)
{
    RenderArgs a = a_in;
    const int tiles_x = (a.width + 15) / 16, tiles_y = (a.y_end - a.y_begin + 15) / 16;
    if (tiles_x <= 0 || tiles_y <= 0) return hipSuccess;
    dim3 grid((unsigned) (tiles_x * tiles_y)), block(256);
    const bool count = (a.flags & RT_FLAG_COUNT_TESTS) != 0;
    // dynamic LDS = [traversal region: fast-mode stack | staged program | nothing] + [workgroup-owned tiles: tile sums + item counter]
    size_t region = 0;
    int trav = 0;
#ifdef RT_WITH_WAVE
    if (a.bvh_stack_levels > 0) { trav = 2; region = (size_t) a.bvh_stack_levels * 256 * sizeof(uint32_t); }
    // the OBJ-loader wave kernels keep no tile sums (rt_wave.h wave_global_sums); launches of few samples per pixel take their one-wave-workgroup
    // twins over the quarters of a tile: a quarter of the stack region each, four times the workgroups
    const bool global_sums = trav == 2 && a.kinds == 3 && !(a.flags & RT_FLAG_LEGACY_LOOP) && !(a.q_slots > 0 && q_pools && q_locks);
    int one_wave_spp = 32;
    if (const char* e = std::getenv("RT_ONE_WAVE_SPP")) one_wave_spp = std::atoi(e);
    const bool one_wave = global_sums && a.spp_end - a.spp_begin <= one_wave_spp;
    if (one_wave) { region /= 4; grid = dim3(grid.x * 4u); }
#endif
    if (trav == 0 && a.lds_ops_bytes + a.lds_prims_bytes > 0) { trav = 1; region = (size_t) a.lds_ops_bytes + a.lds_prims_bytes; }
#ifndef RT_WITH_WAVE
    const bool global_sums = false, one_wave = false;
#endif
    const size_t tile_bytes = global_sums ? 16 : (sizeof(RT_REAL) == sizeof(float) || trav == 2) ? 768 * sizeof(RT_REAL) + 16 : 0;
    a.lds_tile_offset = (uint32_t) region;
    const size_t lds = region + tile_bytes;
#ifdef RT_WITH_WAVE
    if (trav == 2 && !(a.flags & RT_FLAG_LEGACY_LOOP)) {
        if (a.xcd_tiles) { const int supers = ((tiles_x + 7) / 8) * ((tiles_y + 7) / 8); grid = dim3((unsigned) ((supers + 7) / 8 * 8 * 64 * (one_wave ? 4 : 1))); }
        if (a.q_slots > 0 && q_pools && q_locks && (a.kinds == 0 || a.kinds == 3)) {          // decoupled kernel (rt_queue.h): two more KiB of LDS for the lists
            const size_t qlds = lds + 16 + 4 * RT_Q_SLOTS_MAX;
            if (a.kinds == 3) {
                if (a.quantised) return launch_queue<RT_REAL, true, 3>(count, grid, qlds, s, sc, a, accum, counters, q_pools, q_locks);
                return launch_queue<RT_REAL, false, 3>(count, grid, qlds, s, sc, a, accum, counters, q_pools, q_locks);
            }
            return launch_queue<RT_REAL, false, 0>(count, grid, qlds, s, sc, a, accum, counters, q_pools, q_locks);
        }
        // variants by scene content (rt_device.h class_*): 2 instances (unquantised nodes), 1 boxes or volumes, 3 an OBJ-loader scene, 5 spheres and rects, 0 the rest
        if (a.kinds == 2) return launch_wave<RT_REAL, false, 2>(count, grid, lds, s, sc, a, accum, counters);
        if (a.kinds == 1) return launch_wave<RT_REAL, false, 1>(count, grid, lds, s, sc, a, accum, counters);
        if (a.kinds == 5) return launch_wave<RT_REAL, false, 5>(count, grid, lds, s, sc, a, accum, counters);          // spheres and rects (rt_device.h class_*)
        if (a.kinds == 6) return launch_wave<RT_REAL, false, 6>(count, grid, lds, s, sc, a, accum, counters);          // ... with boxes or volumes
        if (a.kinds == 7) return launch_wave<RT_REAL, false, 7>(count, grid, lds, s, sc, a, accum, counters);          // ... with instances
        if (a.kinds == 3) {
            if (global_sums && !(a.flags & RT_FLAG_ACCUMULATE)) {      // the kernel only adds: the rows it renders start from zero
                const hipError_t e = hipMemsetAsync(accum + (size_t) a.y_begin * a.width * 3, 0, (size_t) (a.y_end - a.y_begin) * a.width * 3 * sizeof(RT_REAL), s);
                if (e != hipSuccess) return e;
            }
#ifdef RT_OBJ_WAVE_KERNELS_ELSEWHERE      // fp64: those kernels live in rt_kernels_fp64_obj.hip
            return launch_wave_obj(count, grid, lds, s, sc, a, accum, counters, one_wave);
#else
            if (a.quantised && sc.sorted_visit) return launch_wave_sorted<RT_REAL, 3>(count, grid, lds, s, sc, a, accum, counters, one_wave);
            if (a.quantised) return launch_wave<RT_REAL, true, 3>(count, grid, lds, s, sc, a, accum, counters, one_wave);
            return launch_wave<RT_REAL, false, 3>(count, grid, lds, s, sc, a, accum, counters, one_wave);
#endif
        }
        if (a.quantised && sc.sorted_visit) return launch_wave_sorted<RT_REAL, 0>(count, grid, lds, s, sc, a, accum, counters);
        if (a.quantised) return launch_wave<RT_REAL, true, 0>(count, grid, lds, s, sc, a, accum, counters);
        return launch_wave<RT_REAL, false, 0>(count, grid, lds, s, sc, a, accum, counters);
    }
#endif
    if (lds > 48 * 1024) {
        static LdsConfigured done;
        std::atomic<size_t>* configured = done.of_current_device();
        if (!configured) return hipErrorInvalidDevice;
        if (lds > configured->load(std::memory_order_relaxed)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel<RT_REAL, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel<RT_REAL, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
#ifdef RT_WITH_FAST_BVH
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel<RT_REAL, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel<RT_REAL, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
#endif
            if (e != hipSuccess) return e;
            configured->store(lds, std::memory_order_relaxed);
        }
    }
#ifdef RT_WITH_FAST_BVH
    if (trav == 2) {
        if (count) hipLaunchKernelGGL((render_kernel<RT_REAL, 2, true>), grid, block, lds, s, sc, a, accum, counters);
        else hipLaunchKernelGGL((render_kernel<RT_REAL, 2, false>), grid, block, lds, s, sc, a, accum, counters);
        return hipGetLastError();
    }
#endif
    if (trav == 2) return hipErrorInvalidValue;      // fp64 over the fast tree has no lane-synchronous variant (rtcore.hip refuses the flag)
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

#ifdef RT_WITH_WAVE
hipError_t launch_first_hit_tree(const Scene<RT_REAL>& sc, const double* rays, int n, int quantised, int32_t* prim, RT_REAL* t, int32_t* front, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    const size_t stack_bytes = (size_t) ((quantised ? sc.wide_stack : sc.bvh_depth) + 2) * 256 * sizeof(uint32_t);
    if (stack_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&first_hit_tree_kernel<RT_REAL, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) stack_bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&first_hit_tree_kernel<RT_REAL, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) stack_bytes);
        if (e != hipSuccess) return e;
    }
    if (quantised) hipLaunchKernelGGL((first_hit_tree_kernel<RT_REAL, true>), dim3((n + 255) / 256), dim3(256), stack_bytes, s, sc, rays, n, prim, t, front);
    else hipLaunchKernelGGL((first_hit_tree_kernel<RT_REAL, false>), dim3((n + 255) / 256), dim3(256), stack_bytes, s, sc, rays, n, prim, t, front);
    return hipGetLastError();
}
#endif

#ifdef RT_WITH_FAST_BVH
hipError_t wf_extend_grid(int lds_levels, int* grid_out)
{
    int dev = 0, cus = 0, per_cu = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&wf_extend_kernel<false, false>), 256, (size_t) lds_levels * 1024);
    if (e != hipSuccess) return e;
    *grid_out = std::max(1, cus) * std::max(1, per_cu);
    return hipSuccess;
}

hipError_t wf_render(const Scene<float>& sc, const RenderArgs& a_in, const WfBuffers& buf, float* accum, unsigned long long out[8], hipStream_t s)
{
    RenderArgs a = a_in;
    WfPool p = buf.pool;
    const bool count = (a.flags & RT_FLAG_COUNT_TESTS) != 0;
    const uint32_t tiles_x = (uint32_t) (a.width + 7) / 8, tiles_y = (uint32_t) (a.y_end - a.y_begin + 7) / 8;
    const int samples = a_in.spp_end - a_in.spp_begin;
    for (int i = 0; i < 8; i++) out[i] = 0;
    hipError_t e = hipMemsetAsync(p.ctl, 0, sizeof(WfCtl), s);
    if (e != hipSuccess) return e;
    if (!(a.flags & RT_FLAG_ACCUMULATE) && a.y_end > a.y_begin) {
        e = hipMemsetAsync(accum + (size_t) a.y_begin * a.width * 3, 0, (size_t) (a.y_end - a.y_begin) * a.width * 3 * sizeof(float), s);
        if (e != hipSuccess) return e;
    }
    unsigned int* host = static_cast<unsigned int*>(buf.host_pinned);
    std::vector<unsigned long long> partial;
    if (tiles_x && tiles_y && samples > 0 && a.max_depth > 0) {
        p.tiles_x = tiles_x;
        p.n_pix_padded = tiles_x * tiles_y * 64u;
        const int per_pass = (int) std::max<uint64_t>(1, (1ull << 31) / p.n_pix_padded);     // items of one pass fit 31 bits
        const size_t lds = (size_t) p.lds_levels * 1024;
        const unsigned grid_e = p.extend_threads / 256;
        const bool spill = buf.bvh_depth + 2 > (int) p.lds_levels;
        for (int k0 = a_in.spp_begin; k0 < a_in.spp_end; k0 += per_pass) {
            a.spp_begin = k0;
            a.spp_end = std::min(a_in.spp_end, k0 + per_pass);
            p.n_items = (uint32_t) (a.spp_end - a.spp_begin) * p.n_pix_padded;
            p.n_slots = (uint32_t) std::min<uint64_t>(buf.capacity, ((uint64_t) p.n_items + 255) / 256 * 256);
            p.region_slots = ((p.n_slots + RT_WF_CURSORS - 1) / RT_WF_CURSORS + RT_WF_CHUNK - 1) / RT_WF_CHUNK * RT_WF_CHUNK;
            e = hipMemsetAsync(p.ctl, 0, offsetof(WfCtl, n_box), s);                           // flag and cursors; the diagnostic counters run on
            if (e != hipSuccess) return e;
            const unsigned grid_s = p.n_slots / 256;
            hipLaunchKernelGGL(wf_generate_kernel, dim3(grid_s), dim3(256), 0, s, sc, p, a);
            // greedy scheduling bound: every iteration extends every live slot by one segment, a sample has <= max_depth of them
            const uint64_t max_iters = ((uint64_t) p.n_items / p.n_slots + 2) * (uint64_t) a.max_depth + 64;
            // a slot renders n_items / n_slots samples one after the other, each at least one segment long
            uint64_t iters = 0, batch = std::min<uint64_t>(std::max<uint64_t>(4, (uint64_t) p.n_items / p.n_slots), 64);
            for (;;) {
                for (uint64_t b = 0; b < batch; b++) {
                    if (count) {
                        if (spill) hipLaunchKernelGGL((wf_extend_kernel<true, true>), dim3(grid_e), dim3(256), lds, s, sc, p, a);
                        else hipLaunchKernelGGL((wf_extend_kernel<true, false>), dim3(grid_e), dim3(256), lds, s, sc, p, a);
                        hipLaunchKernelGGL((wf_shade_kernel<true>), dim3(grid_s), dim3(256), 0, s, sc, p, a, accum);
                    }
                    else {
                        if (spill) hipLaunchKernelGGL((wf_extend_kernel<false, true>), dim3(grid_e), dim3(256), lds, s, sc, p, a);
                        else hipLaunchKernelGGL((wf_extend_kernel<false, false>), dim3(grid_e), dim3(256), lds, s, sc, p, a);
                        hipLaunchKernelGGL((wf_shade_kernel<false>), dim3(grid_s), dim3(256), 0, s, sc, p, a, accum);
                    }
                }
                if ((e = hipGetLastError()) != hipSuccess) return e;
                if ((e = hipMemcpyAsync(host, p.ctl, sizeof(unsigned int), hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
                if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
                if (host[0] == 0) break;                                                       // no slot holds a path any more
                if ((iters += batch) > max_iters) return hipErrorLaunchFailure;                // cannot happen: the pool failed to drain
                batch = std::min<uint64_t>(batch, 16);
            }
            partial.resize((size_t) grid_s * 2);
            if ((e = hipMemcpy(partial.data(), p.wg_rays, (size_t) grid_s * 8, hipMemcpyDeviceToHost)) != hipSuccess) return e;
            if ((e = hipMemcpy(partial.data() + grid_s, p.wg_rejected, (size_t) grid_s * 8, hipMemcpyDeviceToHost)) != hipSuccess) return e;
            for (unsigned i = 0; i < grid_s; i++) { out[0] += partial[i]; out[1] += partial[grid_s + i]; }
        }
    }
    if (count) {
        if ((e = hipMemcpyAsync(host, p.ctl, sizeof(WfCtl), hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
        const WfCtl* c = static_cast<const WfCtl*>(buf.host_pinned);
        out[2] = c->n_box; out[3] = c->n_leaf; out[4] = c->w_node; out[5] = c->w_leaf; out[6] = c->w_bounce;
    }
    return hipSuccess;
}

hipError_t launch_first_hit_fast(const Scene<float>& sc, const double* rays, int n, int32_t* prim, float* t, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    const size_t stack_bytes = (size_t) (sc.bvh_depth + 2) * 256 * sizeof(uint32_t);
    hipLaunchKernelGGL(first_hit_fast_kernel, dim3((n + 255) / 256), dim3(256), stack_bytes, s, sc, rays, n, prim, t);
    return hipGetLastError();
}
#endif

#ifdef RT_WITH_FIRST_HIT
hipError_t launch_debug_path(const Scene<double>& sc, const RenderArgs& a, int i, int j, int k, double* rows9, int cap, int* n_out, hipStream_t s)
{
    hipLaunchKernelGGL((path_kernel<double>), dim3(1), dim3(64), 0, s, sc, a, i, j, k, rows9, cap, n_out);
    return hipGetLastError();
}
hipError_t launch_debug_libm(const int32_t* op, const double* x, const double* y, int n, double* out, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(libm_kernel, dim3((n + 255) / 256), dim3(256), 0, s, op, x, y, n, out);
    return hipGetLastError();
}

hipError_t launch_first_hit(const Scene<double>& sc, const double* rays, int n, int32_t* prim, double* t, int32_t* front,
                            uint32_t* n_box, uint32_t* n_leaf, unsigned long long* hash, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL((first_hit_kernel<double>), dim3((n + 255) / 256), dim3(256), 0, s, sc, rays, n, prim, t, front, n_box, n_leaf, hash);
    return hipGetLastError();
}
#endif

}  // namespace rt
