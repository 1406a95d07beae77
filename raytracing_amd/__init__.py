"""raytracing_amd -- MI355X-native path-tracing core behind the reference's scene interface.

Python here is plumbing (ctypes over the C ABI in include/rtcore.h, used by tests and bench.py);
the product is raytracing_amd/librtcore.so (hand-written HIP for gfx950) and the C++ host mirror in
raytracing_amd/host/.
"""
from .rtcore import (  # noqa: F401
    MODE_FAST_FP32, MODE_REFERENCE_FP64, MODE_FP32_REFERENCE_TREE, MODE_FAST_FP64, FLAG_ACCUMULATE, FLAG_EARLY_EXIT, FLAG_COUNT_TESTS, FLAG_DEBUG_RAY_COUNTS, FLAG_LEGACY_LOOP, FLAG_WAVEFRONT, FLAG_QUEUE, FLAG_COMPAT_FIXED,
    DeviceBuffer, RtError, Scene, checkpoint_load, checkpoint_save, debug_libm, debug_rng, device_count, frame_reduce, frame_reduce_release, load_library, load_scene_cache, resolve, save_scene_cache,
)
