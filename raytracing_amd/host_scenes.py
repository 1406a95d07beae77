"""ctypes binding of raytracing_amd/librthost.so: the C++ host mirror's scene builders + flatten.

This is the PRODUCT path for scene construction (bench.py, smoke): reference-compatible C++ classes build the scene
and its BVH, flatten() writes the rt_scene_desc that librtcore.so uploads.
"""
import ctypes as C
import os

import numpy as np

from .rtcore import PRIM_STRIDE, MAT_STRIDE, TEX_STRIDE, RtError, SceneDesc, load_library, _check

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librthost.so")
_lib = None


def load_host_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RtError(f"{LIB_PATH} is missing: build it with `make -C raytracing_amd/host`")
        load_library()   # librthost links librtcore
        L = C.CDLL(LIB_PATH)
        L.rth_last_error.restype = C.c_char_p
        L.rth_build_random_spheres.restype = C.c_void_p
        L.rth_build_random_spheres.argtypes = [C.c_int, C.c_int, C.c_uint]
        for name in ("rth_build_mixed", "rth_build_deep_bounce", "rth_build_textured", "rth_build_cornell", "rth_build_smoke", "rth_build_features", "rth_build_features_lit", "rth_build_instanced", "rth_build_nested"):
            getattr(L, name).restype = C.c_void_p
            getattr(L, name).argtypes = [C.c_int, C.c_int, C.c_uint]
        L.rth_build_obj.restype = C.c_void_p
        L.rth_build_obj.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_uint, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double]
        L.rth_desc.restype = C.POINTER(SceneDesc)
        L.rth_desc.argtypes = [C.c_void_p]
        L.rth_free.argtypes = [C.c_void_p]
        L.rth_load_image.restype = C.c_longlong
        L.rth_load_image.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_ulonglong]
        _lib = L
    return _lib


def load_image(path):
    """The texels an image texture holds after loading `path` (PNG, JPEG or PPM), as the reference's loader hands them over
    (texture.hpp:87): uint8 array of shape (height, width, channels), top row first."""
    L = load_host_library()
    w, h, ch = C.c_int(), C.c_int(), C.c_int()
    n = L.rth_load_image(path.encode(), C.byref(w), C.byref(h), C.byref(ch), None, 0)
    if n < 0:
        raise RtError(L.rth_last_error().decode())
    out = np.zeros(n, np.uint8)
    if L.rth_load_image(path.encode(), C.byref(w), C.byref(h), C.byref(ch), out.ctypes.data_as(C.c_void_p), n) != n:
        raise RtError(L.rth_last_error().decode())
    return out.reshape(h.value, w.value, ch.value)


class HostScene:
    """A scene built and flattened by the C++ host mirror; `.desc` is a pointer to its rt_scene_desc."""

    def __init__(self, handle, width, height):
        if not handle:
            raise RtError("host scene build failed: " + load_host_library().rth_last_error().decode())
        self.h = C.c_void_p(handle)
        self.width, self.height = width, height
        self.desc = load_host_library().rth_desc(self.h)

    @classmethod
    def random_spheres(cls, width, height, seed=1):
        return cls(load_host_library().rth_build_random_spheres(width, height, seed), width, height)

    @classmethod
    def mixed(cls, width, height, seed=1):
        return cls(load_host_library().rth_build_mixed(width, height, seed), width, height)

    @classmethod
    def cornell(cls, width, height, seed=1):
        """the scene of the reference's main(): raytracing.cpp:77-88 over cornell_box() (demo.hpp:63-91)"""
        return cls(load_host_library().rth_build_cornell(width, height, seed), width, height)

    @classmethod
    def features(cls, width, height, seed=1):
        """all_feature_test() (demo.hpp:94-143): boxes, media, a failed image texture, translate(rotatey(bvh(1000 spheres)))"""
        return cls(load_host_library().rth_build_features(width, height, seed), width, height)

    @classmethod
    def features_lit(cls, width, height, seed=1):
        """all_feature_test() under a flipped rect light just below the demo's own (which shines upwards): a lit picture"""
        return cls(load_host_library().rth_build_features_lit(width, height, seed), width, height)

    @classmethod
    def instanced(cls, width, height, seed=1):
        """the Cornell room with instances and no medium: a turned and moved cloud of spheres under its own tree, a moved and a turned sphere"""
        return cls(load_host_library().rth_build_instanced(width, height, seed), width, height)

    @classmethod
    def nested(cls, width, height, seed=1):
        """nested_test() (host/demo.hpp): wrappers nested every way, volumes around boxes, groups used as objects, an instance in an instance"""
        return cls(load_host_library().rth_build_nested(width, height, seed), width, height)

    @classmethod
    def smoke(cls, width, height, seed=1):
        """volume / fog: sphere-bounded media of all_feature_test() (demo.hpp:117-134) over a ground rect"""
        return cls(load_host_library().rth_build_smoke(width, height, seed), width, height)

    @classmethod
    def deep_bounce(cls, width, height, seed=1):
        return cls(load_host_library().rth_build_deep_bounce(width, height, seed), width, height)

    @classmethod
    def textured(cls, width, height, seed=1):
        return cls(load_host_library().rth_build_textured(width, height, seed), width, height)

    @classmethod
    def obj(cls, path, width, height, seed=1, cam_pos=(0, 1.6, 4.2), cam_lookat=(0, 0.6, 0), fov=40.0):
        L = load_host_library()
        return cls(L.rth_build_obj(path.encode(), width, height, seed, (C.c_double * 3)(*cam_pos), (C.c_double * 3)(*cam_lookat), fov), width, height)

    def free(self):
        if self.h:
            load_host_library().rth_free(self.h)
            self.h = None

    def upload(self):
        """rt_scene_create on the current HIP device -> raytracing_amd.Scene"""
        from .rtcore import Scene
        sc = Scene.__new__(Scene)
        sc._keep = self
        h = C.c_void_p()
        _check(load_library().rt_scene_create(self.desc, C.byref(h)))
        sc.h = h
        return sc

    def arrays(self):
        """numpy copies of the flattened arrays (same keys as tests/oracle_lib.Scene.export)."""
        d = self.desc.contents

        def arr(ptr, n, dtype):
            if n == 0:
                return np.zeros(0, dtype)
            return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)

        out = dict(
            node_bbox=arr(d.node_bbox, d.n_nodes * 6, np.float64).reshape(-1, 6), node_child=arr(d.node_child, d.n_nodes * 2, np.int32).reshape(-1, 2),
            roots=arr(d.roots, d.n_roots, np.int32),
            prim_kind=arr(d.prim_kind, d.n_prims, np.int32), prim_flags=arr(d.prim_flags, d.n_prims, np.int32), prim_mat=arr(d.prim_mat, d.n_prims, np.int32),
            prim_data=arr(d.prim_data, d.n_prims * PRIM_STRIDE, np.float64).reshape(-1, PRIM_STRIDE),
            mat_kind=arr(d.mat_kind, d.n_mats, np.int32), mat_tex=arr(d.mat_tex, d.n_mats, np.int32),
            mat_data=arr(d.mat_data, d.n_mats * MAT_STRIDE, np.float64).reshape(-1, MAT_STRIDE),
            tex_kind=arr(d.tex_kind, d.n_tex, np.int32), tex_data=arr(d.tex_data, d.n_tex * TEX_STRIDE, np.float64).reshape(-1, TEX_STRIDE),
            tex_pixels=arr(d.tex_pixels, d.n_pixel_bytes, np.uint8),
            light_kind=arr(d.light_kind, d.n_lights, np.int32), light_flags=arr(d.light_flags, d.n_lights, np.int32),
            light_mat=arr(d.light_mat, d.n_lights, np.int32),
            light_data=arr(d.light_data, d.n_lights * PRIM_STRIDE, np.float64).reshape(-1, PRIM_STRIDE),
            camera=np.array(list(d.camera)), n_pixel_bytes=int(d.n_pixel_bytes), width=self.width, height=self.height,
            xform_data=arr(d.xform_data, d.n_xforms * 6, np.float64).reshape(-1, 6),
            node_xform=arr(d.node_xform, d.n_nodes if d.n_xforms else 0, np.int32), prim_xform=arr(d.prim_xform, d.n_prims if d.n_xforms else 0, np.int32),
        )
        return out
