// raytracing_amd/csrc/rt_wave_launch.h -- host launchers of the wave kernels (rt_wave.h), shared by the translation units that hold them:
// rt_kernels_fp32.hip, rt_kernels_fp64.hip and rt_kernels_fp64_obj.hip (the fp64 OBJ-loader variants, which are scheduled with another
// strategy: csrc/Makefile).
#pragma once
#include <atomic>
#include <cstddef>
#include "rt_wave.h"

namespace rt {

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of a kernel ON A DEVICE: a process that drives several devices (rt_device_set,
// scene::render(..., devices)) must set it on each of them.  One instance per kernel variant remembers the largest size set so far per
// device (atomics: renders of different scenes may be launched from different host threads; setting the attribute twice is harmless).
struct LdsConfigured {
    static constexpr int kMaxDevices = 64;
    std::atomic<size_t> bytes[kMaxDevices] = {};
    std::atomic<size_t>* of_current_device()
    {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
        return &bytes[dev];
    }
};

// one variant of the wave kernel (rt_wave.h); more than 48 KiB of dynamic LDS needs the attribute set once per kernel.  one_wave: the
// one-wave-workgroup twin (kernels without tile sums; its LDS is a quarter, far below 48 KiB)
template <typename R, bool kQuant, int kFull>
hipError_t launch_wave(bool count, dim3 grid, size_t lds, hipStream_t s, const Scene<R>& sc, const RenderArgs& a, R* accum, unsigned long long* counters, bool one_wave = false)
{
    if constexpr (wave_global_sums<R, kFull>()) {
        if (one_wave) {
            if (lds > 48 * 1024) return hipErrorInvalidValue;
            if (count) hipLaunchKernelGGL((render_wave1_kernel<R, true, kQuant, kFull>), grid, dim3(64), lds, s, sc, a, accum, counters);
            else hipLaunchKernelGGL((render_wave1_kernel<R, false, kQuant, kFull>), grid, dim3(64), lds, s, sc, a, accum, counters);
            return hipGetLastError();
        }
    }
    if (lds > 48 * 1024) {
        static LdsConfigured done;
        std::atomic<size_t>* configured = done.of_current_device();
        if (!configured) return hipErrorInvalidDevice;
        if (lds > configured->load(std::memory_order_relaxed)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_wave_kernel<R, false, kQuant, kFull>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_wave_kernel<R, true, kQuant, kFull>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e != hipSuccess) return e;
            configured->store(lds, std::memory_order_relaxed);
        }
    }
    if (count) hipLaunchKernelGGL((render_wave_kernel<R, true, kQuant, kFull>), grid, dim3(256), lds, s, sc, a, accum, counters);
    else hipLaunchKernelGGL((render_wave_kernel<R, false, kQuant, kFull>), grid, dim3(256), lds, s, sc, a, accum, counters);
    return hipGetLastError();
}

// ... and its twin that visits the wide nodes' hit children by entry distance (Scene::sorted_visit)
template <typename R, int kFull>
hipError_t launch_wave_sorted(bool count, dim3 grid, size_t lds, hipStream_t s, const Scene<R>& sc, const RenderArgs& a, R* accum, unsigned long long* counters, bool one_wave = false)
{
    if constexpr (wave_global_sums<R, kFull>()) {
        if (one_wave) {
            if (lds > 48 * 1024) return hipErrorInvalidValue;
            if (count) hipLaunchKernelGGL((render_wave1_sorted_kernel<R, true, kFull>), grid, dim3(64), lds, s, sc, a, accum, counters);
            else hipLaunchKernelGGL((render_wave1_sorted_kernel<R, false, kFull>), grid, dim3(64), lds, s, sc, a, accum, counters);
            return hipGetLastError();
        }
    }
    if (lds > 48 * 1024) {
        static LdsConfigured done;
        std::atomic<size_t>* configured = done.of_current_device();
        if (!configured) return hipErrorInvalidDevice;
        if (lds > configured->load(std::memory_order_relaxed)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_wave_sorted_kernel<R, false, kFull>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_wave_sorted_kernel<R, true, kFull>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e != hipSuccess) return e;
            configured->store(lds, std::memory_order_relaxed);
        }
    }
    if (count) hipLaunchKernelGGL((render_wave_sorted_kernel<R, true, kFull>), grid, dim3(256), lds, s, sc, a, accum, counters);
    else hipLaunchKernelGGL((render_wave_sorted_kernel<R, false, kFull>), grid, dim3(256), lds, s, sc, a, accum, counters);
    return hipGetLastError();
}

}  // namespace rt
