// raytracing_amd/csrc/rt_launch.h -- host-callable launchers, one overload set per precision.
// rt_kernels_fp64.hip is compiled with -ffp-contract=off (bit-exact reference arithmetic),
// rt_kernels_fp32.hip with the default contraction.
#pragma once
#include "rt_device.h"
#include "rt_wavefront.h"

namespace rt {
// q_pools / q_locks: path pools and their locks for the decoupled kernel (rt_queue.h), used when a.q_slots > 0; may be null otherwise
hipError_t launch_render(const Scene<double>& sc, const RenderArgs& a, double* accum, unsigned long long* counters, hipStream_t s, void* q_pools = nullptr, unsigned int* q_locks = nullptr);
hipError_t launch_render(const Scene<float>& sc, const RenderArgs& a, float* accum, unsigned long long* counters, hipStream_t s, void* q_pools = nullptr, unsigned int* q_locks = nullptr);
// the fp64 OBJ-loader wave kernels (their own translation unit, rt_kernels_fp64_obj.hip): the variant by node format and visiting order
hipError_t launch_wave_obj(bool count, dim3 grid, size_t lds, hipStream_t s, const Scene<double>& sc, const RenderArgs& a, double* accum, unsigned long long* counters, bool one_wave);
size_t queue_slot_bytes(double);
size_t queue_slot_bytes(float);
// RT_MODE_FAST_FP32: wavefront loop over a path pool (rt_wavefront.h).  Blocks until the frame's samples are done; fills
// out[0..6] = rays, rejected samples, box tests, leaf tests, wave node steps, wave leaf steps, shade waves.
struct WfBuffers { WfPool pool; uint32_t capacity; int bvh_depth; void* host_pinned; };
hipError_t wf_extend_grid(int lds_levels, int* grid_out);
hipError_t wf_render(const Scene<float>& sc, const RenderArgs& a, const WfBuffers& buf, float* accum, unsigned long long out[8], hipStream_t s);
// closest hit of explicit rays through the fast tree with the wave kernel's own culling code (rt_wave.h): R = double is
// RT_MODE_FAST_FP64's traversal, whose primitive, t bits and frontward flag must be the reference walk's
hipError_t launch_first_hit_tree(const Scene<double>& sc, const double* rays, int n, int quantised, int32_t* prim, double* t, int32_t* front, hipStream_t s);
hipError_t launch_first_hit_tree(const Scene<float>& sc, const double* rays, int n, int quantised, int32_t* prim, float* t, int32_t* front, hipStream_t s);
hipError_t launch_first_hit_fast(const Scene<float>& sc, const double* rays, int n, int32_t* prim, float* t, hipStream_t s);
hipError_t launch_first_hit(const Scene<double>& sc, const double* rays, int n, int32_t* prim, double* t, int32_t* front,
                            uint32_t* n_box, uint32_t* n_leaf, unsigned long long* hash, hipStream_t s);
// test tooling (tools/first_divergence.py): one (pixel, sample) path through the reference-order walk, and the device library's results for
// a list of libm calls.  fp64 translation unit only.
hipError_t launch_debug_path(const Scene<double>& sc, const RenderArgs& a, int i, int j, int k, double* rows9, int cap, int* n_out, hipStream_t s);
hipError_t launch_debug_libm(const int32_t* op, const double* x, const double* y, int n, double* out, hipStream_t s);
}  // namespace rt
