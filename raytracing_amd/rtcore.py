"""ctypes binding of raytracing_amd/librtcore.so (include/rtcore.h).

There is no CPU fallback: if the HIP library is missing or no GPU is present, calls raise.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTCORE_LIB") or os.path.join(HERE, "librtcore.so")   # RTCORE_LIB: A/B builds of the same library (tools/)

MODE_REFERENCE_FP64, MODE_FAST_FP32, MODE_FP32_REFERENCE_TREE, MODE_FAST_FP64 = 0, 1, 2, 3
FLAG_ACCUMULATE, FLAG_EARLY_EXIT, FLAG_COUNT_TESTS, FLAG_DEBUG_RAY_COUNTS, FLAG_LEGACY_LOOP, FLAG_WAVEFRONT, FLAG_QUEUE, FLAG_COMPAT_FIXED = 1, 2, 4, 8, 16, 32, 64, 128
PRIM_STRIDE, MAT_STRIDE, TEX_STRIDE = 24, 16, 8
NOALLOC_MAX_RAYS = 32768

EXPORTS = ["rt_scene_create", "rt_scene_destroy", "rt_render", "rt_render_wait", "rt_render_kernel_name", "rt_device_count", "rt_device_set", "rt_frame_reduce", "rt_frame_reduce_release", "rt_resolve", "rt_first_hit", "rt_first_hit_noalloc", "rt_first_hit_fast", "rt_first_hit_tree", "rt_buffer_alloc", "rt_buffer_free",
           "rt_buffer_download", "rt_buffer_upload", "rt_debug_rng", "rt_debug_path", "rt_debug_libm", "rt_last_error", "rt_version",
           "rt_scene_desc_save", "rt_scene_desc_load", "rt_scene_desc_free", "rt_checkpoint_save", "rt_checkpoint_load"]


class RtError(RuntimeError):
    pass


class SceneDesc(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int32), ("node_bbox", C.POINTER(C.c_double)), ("node_child", C.POINTER(C.c_int32)),
        ("n_roots", C.c_int32), ("roots", C.POINTER(C.c_int32)),
        ("n_prims", C.c_int32), ("prim_kind", C.POINTER(C.c_int32)), ("prim_flags", C.POINTER(C.c_int32)), ("prim_mat", C.POINTER(C.c_int32)),
        ("prim_data", C.POINTER(C.c_double)),
        ("n_mats", C.c_int32), ("mat_kind", C.POINTER(C.c_int32)), ("mat_tex", C.POINTER(C.c_int32)), ("mat_data", C.POINTER(C.c_double)),
        ("n_tex", C.c_int32), ("tex_kind", C.POINTER(C.c_int32)), ("tex_data", C.POINTER(C.c_double)),
        ("n_pixel_bytes", C.c_size_t), ("tex_pixels", C.POINTER(C.c_uint8)),
        ("n_lights", C.c_int32), ("light_kind", C.POINTER(C.c_int32)), ("light_flags", C.POINTER(C.c_int32)), ("light_mat", C.POINTER(C.c_int32)),
        ("light_data", C.POINTER(C.c_double)),
        ("camera", C.c_double * 24),
        ("n_xforms", C.c_int32), ("xform_data", C.POINTER(C.c_double)), ("node_xform", C.POINTER(C.c_int32)), ("prim_xform", C.POINTER(C.c_int32)),
    ]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("y_begin", C.c_int32), ("y_end", C.c_int32),
                ("spp_begin", C.c_int32), ("spp_end", C.c_int32), ("max_depth", C.c_int32), ("mode", C.c_int32), ("flags", C.c_int32),
                ("seed", C.c_uint64), ("stream", C.c_void_p)]


class RenderStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("rejected_samples", C.c_uint64), ("kernel_ms", C.c_double), ("box_tests", C.c_uint64), ("leaf_tests", C.c_uint64),
                ("wave_node_steps", C.c_uint64), ("wave_leaf_steps", C.c_uint64), ("wave_bounces", C.c_uint64), ("phase_ticks", C.c_uint64 * 4), ("zero_direction_rays", C.c_uint64),
                ("untraced_rays", C.c_uint64)]


_lib = None


def load_library():
    """Load librtcore.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RtError(f"{LIB_PATH} is missing: build it with `make -C raytracing_amd/csrc` (or __graft_entry__.build())")
        L = C.CDLL(LIB_PATH)
        L.rt_last_error.restype = C.c_char_p
        L.rt_version.restype = C.c_char_p
        L.rt_scene_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
        L.rt_scene_destroy.argtypes = [C.c_void_p]
        L.rt_render.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.POINTER(RenderStats)]
        L.rt_render_wait.argtypes = [C.c_void_p, C.POINTER(RenderStats)]
        L.rt_render_kernel_name.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
        L.rt_device_count.argtypes = [C.POINTER(C.c_int)]
        L.rt_device_set.argtypes = [C.c_int]
        L.rt_frame_reduce.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_size_t]
        L.rt_resolve.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.rt_first_hit.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                   C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        L.rt_first_hit_noalloc.argtypes = L.rt_first_hit.argtypes
        L.rt_first_hit_tree.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        L.rt_first_hit_fast.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_float)]
        L.rt_buffer_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
        L.rt_buffer_free.argtypes = [C.c_void_p]
        L.rt_buffer_download.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.rt_buffer_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.rt_debug_rng.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]
        L.rt_debug_path.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.c_int32,
                                    C.POINTER(C.c_int32)]
        L.rt_debug_libm.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double)]
        L.rt_scene_desc_save.argtypes = [C.POINTER(SceneDesc), C.c_char_p]
        L.rt_scene_desc_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(SceneDesc))]
        L.rt_scene_desc_free.argtypes = [C.POINTER(SceneDesc)]
        L.rt_checkpoint_save.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_int32, C.c_uint64]
        L.rt_checkpoint_load.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise RtError(f"rtcore error {rc}: {load_library().rt_last_error().decode()}")


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class DeviceBuffer:
    """Raw device allocation through the C ABI (hosts without torch)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _check(load_library().rt_buffer_alloc(self.nbytes, C.byref(p)))
        self.ptr = p.value

    def download(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        _check(load_library().rt_buffer_download(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        _check(load_library().rt_buffer_upload(self.ptr, arr.ctypes.data, arr.nbytes))

    def free(self):
        if self.ptr:
            load_library().rt_buffer_free(self.ptr)
            self.ptr = None


def make_desc(d):
    """dict of numpy arrays (rt_scene_desc field names; see include/rtcore.h) -> (SceneDesc, keepalive list)."""
    f64 = lambda k: np.ascontiguousarray(d[k], dtype=np.float64)
    i32 = lambda k: np.ascontiguousarray(d[k], dtype=np.int32)
    keep = {k: f64(k) for k in ("node_bbox", "prim_data", "mat_data", "tex_data", "light_data")}
    keep.update({k: i32(k) for k in ("node_child", "roots", "prim_kind", "prim_flags", "prim_mat", "mat_kind", "mat_tex", "tex_kind",
                                     "light_kind", "light_flags", "light_mat")})
    keep["tex_pixels"] = np.ascontiguousarray(d.get("tex_pixels", np.zeros(1, np.uint8)), dtype=np.uint8)
    s = SceneDesc()
    if "xform_data" in d and len(np.asarray(d["xform_data"]).ravel()):      # instances (optional)
        keep["xform_data"] = f64("xform_data"); keep["node_xform"] = i32("node_xform"); keep["prim_xform"] = i32("prim_xform")
        s.n_xforms = keep["xform_data"].size // 6
    s.n_nodes = len(keep["node_child"].reshape(-1, 2))
    s.n_roots = keep["roots"].size
    s.n_prims = keep["prim_kind"].size
    s.n_mats = keep["mat_kind"].size
    s.n_tex = keep["tex_kind"].size
    s.n_lights = keep["light_kind"].size
    s.n_pixel_bytes = int(d.get("n_pixel_bytes", 0))
    for k, a in keep.items():
        t = C.c_double if a.dtype == np.float64 else (C.c_int32 if a.dtype == np.int32 else C.c_uint8)
        setattr(s, k, _ptr(a, t))
    cam = f64("camera").ravel()
    for i in range(24):
        s.camera[i] = cam[i]
    return s, keep


class Scene:
    """A scene resident in HBM on the current HIP device."""

    def __init__(self, desc_arrays):
        L = load_library()
        desc, self._keep = make_desc(desc_arrays)
        h = C.c_void_p()
        _check(L.rt_scene_create(C.byref(desc), C.byref(h)))
        self.h = h

    def destroy(self):
        if self.h:
            load_library().rt_scene_destroy(self.h)
            self.h = None

    def first_hit(self, rays, noalloc=False):
        """noalloc: through rt_first_hit_noalloc (staging in the library's static device array, at most NOALLOC_MAX_RAYS rays)"""
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 7)
        n = len(rays)
        out = dict(prim=np.zeros(n, np.int32), t=np.zeros(n), front=np.zeros(n, np.int32), nbox=np.zeros(n, np.uint32),
                   nleaf=np.zeros(n, np.uint32), hash=np.zeros(n, np.uint64))
        fn = load_library().rt_first_hit_noalloc if noalloc else load_library().rt_first_hit
        _check(fn(self.h, _ptr(rays, C.c_double), n, _ptr(out["prim"], C.c_int32), _ptr(out["t"], C.c_double),
                                           _ptr(out["front"], C.c_int32), _ptr(out["nbox"], C.c_uint32), _ptr(out["nleaf"], C.c_uint32),
                                           _ptr(out["hash"], C.c_uint64)))
        return out

    def first_hit_fast(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 7)
        n = len(rays)
        prim, t = np.zeros(n, np.int32), np.zeros(n, np.float32)
        _check(load_library().rt_first_hit_fast(self.h, _ptr(rays, C.c_double), n, _ptr(prim, C.c_int32), _ptr(t, C.c_float)))
        return dict(prim=prim, t=t)

    def first_hit_tree(self, rays, fp64=True):
        """closest hits through the fast tree with the render kernels' culling code (rt_first_hit_tree)"""
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 7)
        n = len(rays)
        out = dict(prim=np.zeros(n, np.int32), t=np.zeros(n), front=np.zeros(n, np.int32))
        _check(load_library().rt_first_hit_tree(self.h, _ptr(rays, C.c_double), n, int(fp64), _ptr(out["prim"], C.c_int32), _ptr(out["t"], C.c_double),
                                                _ptr(out["front"], C.c_int32)))
        return out

    def render(self, accum_ptr, width, height, spp_begin, spp_end, max_depth, mode=MODE_FAST_FP32, seed=0, flags=0,
               y_begin=0, y_end=None, stream=None, want_stats=True):
        """accum_ptr: device pointer (int) to width*height*3 doubles (reference mode) or floats (fast mode)."""
        p = RenderParams(width, height, y_begin, height if y_end is None else y_end, spp_begin, spp_end, max_depth, mode, flags, seed,
                         C.c_void_p(stream) if stream else None)
        st = RenderStats()
        _check(load_library().rt_render(self.h, C.byref(p), C.c_void_p(accum_ptr), C.byref(st) if want_stats else None))
        return dict(rays=int(st.rays), rejected=int(st.rejected_samples), kernel_ms=float(st.kernel_ms), box_tests=int(st.box_tests),
                    leaf_tests=int(st.leaf_tests), wave_node_steps=int(st.wave_node_steps), wave_leaf_steps=int(st.wave_leaf_steps),
                    wave_bounces=int(st.wave_bounces), phase_ticks=[int(x) for x in st.phase_ticks], zero_direction_rays=int(st.zero_direction_rays),
                    untraced_rays=int(st.untraced_rays)) if want_stats else None

    def kernel_name(self, mode, flags=0):
        """name of the kernel a render of this scene launches, as rocprofv3 lists it (rt_render_kernel_name)"""
        buf = C.create_string_buffer(256)
        _check(load_library().rt_render_kernel_name(self.h, mode, flags, buf, 256))
        return buf.value.decode()

    def debug_path(self, width, height, i, j, k, seed, max_depth, flags=0):
        """segments of one (pixel, sample) path in the fp64 reference mode (rt_debug_path): rows of ray origin, direction, time, hit distance, primitive"""
        out = np.zeros((max_depth + 1, 9))
        n = C.c_int32()
        _check(load_library().rt_debug_path(self.h, width, height, i, j, k, seed, max_depth, flags, _ptr(out, C.c_double), len(out), C.byref(n)))
        return out[:min(n.value, len(out))]

    def render_wait(self):
        """collect a render started with want_stats=False (rt_render_wait)"""
        st = RenderStats()
        _check(load_library().rt_render_wait(self.h, C.byref(st)))
        return dict(rays=int(st.rays), rejected=int(st.rejected_samples), kernel_ms=float(st.kernel_ms), untraced_rays=int(st.untraced_rays))

    def render_to_host(self, width, height, spp_begin, spp_end, max_depth, mode=MODE_FAST_FP32, seed=0, flags=0, y_begin=0, y_end=None):
        """Convenience for tests: render into a fresh device buffer, return (sums[h,w,3], rgb[h,w,3], stats).  With a row
        sub-range the buffer starts zeroed, so rows outside it read as zero."""
        dt = np.float64 if mode in (MODE_REFERENCE_FP64, MODE_FAST_FP64) else np.float32
        n = width * height * 3
        acc = DeviceBuffer(n * np.dtype(dt).itemsize)
        rgb = DeviceBuffer(n)
        try:
            if y_begin != 0 or y_end not in (None, height):
                acc.upload(np.zeros(n, dt))
            stats = self.render(acc.ptr, width, height, spp_begin, spp_end, max_depth, mode, seed, flags, y_begin, y_end)
            resolve(acc.ptr, mode in (MODE_REFERENCE_FP64, MODE_FAST_FP64), width * height, spp_end - spp_begin, rgb.ptr)
            return acc.download(dt, (height, width, 3)), rgb.download(np.uint8, (height, width, 3)), stats
        finally:
            acc.free()
            rgb.free()


def desc_to_arrays(desc):
    """SceneDesc (by value or pointer contents) -> dict of numpy copies, the inverse of make_desc (used by the cache round trip)."""
    def arr(ptr, n, dt):
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)
    d = desc
    return dict(
        node_bbox=arr(d.node_bbox, d.n_nodes * 6, np.float64).reshape(-1, 6), node_child=arr(d.node_child, d.n_nodes * 2, np.int32).reshape(-1, 2),
        roots=arr(d.roots, d.n_roots, np.int32), prim_kind=arr(d.prim_kind, d.n_prims, np.int32), prim_flags=arr(d.prim_flags, d.n_prims, np.int32),
        prim_mat=arr(d.prim_mat, d.n_prims, np.int32), prim_data=arr(d.prim_data, d.n_prims * PRIM_STRIDE, np.float64).reshape(-1, PRIM_STRIDE),
        mat_kind=arr(d.mat_kind, d.n_mats, np.int32), mat_tex=arr(d.mat_tex, d.n_mats, np.int32),
        mat_data=arr(d.mat_data, d.n_mats * MAT_STRIDE, np.float64).reshape(-1, MAT_STRIDE),
        tex_kind=arr(d.tex_kind, d.n_tex, np.int32), tex_data=arr(d.tex_data, d.n_tex * TEX_STRIDE, np.float64).reshape(-1, TEX_STRIDE),
        tex_pixels=arr(d.tex_pixels, d.n_pixel_bytes, np.uint8), n_pixel_bytes=int(d.n_pixel_bytes),
        light_kind=arr(d.light_kind, d.n_lights, np.int32), light_flags=arr(d.light_flags, d.n_lights, np.int32),
        light_mat=arr(d.light_mat, d.n_lights, np.int32), light_data=arr(d.light_data, d.n_lights * PRIM_STRIDE, np.float64).reshape(-1, PRIM_STRIDE),
        camera=np.array(list(d.camera), np.float64),
        xform_data=arr(d.xform_data, d.n_xforms * 6, np.float64).reshape(-1, 6),
        node_xform=arr(d.node_xform, d.n_nodes if d.n_xforms else 0, np.int32), prim_xform=arr(d.prim_xform, d.n_prims if d.n_xforms else 0, np.int32))


def save_scene_cache(desc_arrays, path):
    """Flattened scene (dict as for Scene()) -> binary cache file (rt_scene_desc_save)."""
    desc, keep = make_desc(desc_arrays)
    _check(load_library().rt_scene_desc_save(C.byref(desc), os.fsencode(path)))


def load_scene_cache(path):
    """Binary cache file -> dict of arrays (rt_scene_desc_load + copy + rt_scene_desc_free)."""
    L = load_library()
    p = C.POINTER(SceneDesc)()
    _check(L.rt_scene_desc_load(os.fsencode(path), C.byref(p)))
    try:
        return desc_to_arrays(p.contents)
    finally:
        L.rt_scene_desc_free(p)


def checkpoint_save(path, accum_ptr, is_fp64, width, height, spp_done, seed):
    _check(load_library().rt_checkpoint_save(os.fsencode(path), C.c_void_p(accum_ptr), int(is_fp64), width, height, spp_done, seed))


def checkpoint_load(path, accum_ptr, is_fp64, width, height):
    spp, seed = C.c_int32(), C.c_uint64()
    _check(load_library().rt_checkpoint_load(os.fsencode(path), C.c_void_p(accum_ptr), int(is_fp64), width, height, C.byref(spp), C.byref(seed)))
    return spp.value, seed.value


def frame_reduce(dst_ptr, dst_device, src_ptrs, src_devices, is_fp64, n_values):
    """dst += every source buffer (rt_frame_reduce: peer copies + adds; one process driving several devices)"""
    n = len(src_ptrs)
    ptrs = (C.c_void_p * max(n, 1))(*[C.c_void_p(p) for p in src_ptrs])
    devs = (C.c_int * max(n, 1))(*src_devices)
    _check(load_library().rt_frame_reduce(C.c_void_p(dst_ptr), dst_device, ptrs, devs, n, int(is_fp64), n_values))


def frame_reduce_release():
    """free the staging buffers rt_frame_reduce keeps between calls"""
    _check(load_library().rt_frame_reduce_release())


def device_count():
    n = C.c_int()
    _check(load_library().rt_device_count(C.byref(n)))
    return n.value


def resolve(accum_ptr, is_fp64, n_pixels, spp, rgb_ptr, stream=None):
    _check(load_library().rt_resolve(C.c_void_p(accum_ptr), int(is_fp64), n_pixels, spp, C.c_void_p(rgb_ptr), C.c_void_p(stream) if stream else None))


def debug_libm(op, x, y):
    """the device library's results for libm calls (op 1 sin, 2 cos, 3 acos, 4 atan2(x, y), 5 pow(x, y), 6 log; 7 sign of sin(x) as the checker texture takes it, 8 x^5 as the dielectric's reflectance takes it): rt_debug_libm"""
    op = np.ascontiguousarray(op, dtype=np.int32); x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.zeros(len(op))
    _check(load_library().rt_debug_libm(_ptr(op, C.c_int32), _ptr(x, C.c_double), _ptr(y, C.c_double), len(op), _ptr(out, C.c_double)))
    return out


def debug_rng(seed, pixel, sample, n):
    out = np.zeros(n, np.uint32)
    _check(load_library().rt_debug_rng(seed, pixel, sample, n, _ptr(out, C.c_uint32)))
    return out
