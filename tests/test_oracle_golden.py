"""CPU tests: the oracle (oracle/rt_oracle.c) against golden fixtures made from the REAL reference.

Every comparison of floating-point data is BIT-EXACT (uint64 views): the oracle restates the reference's
arithmetic in the reference's evaluation order and is fed a clone of glibc's rand() stream.
"""
import os

import numpy as np
import pytest

import oracle_lib as O


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).ravel().view(np.uint64)


def assert_bits(a, b, what):
    a, b = bits(a), bits(b)
    assert a.shape == b.shape, what
    bad = np.flatnonzero(a != b)
    assert bad.size == 0, f"{what}: {bad.size} of {a.size} values differ, first at {bad[:5]}"


def load(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = {k: int(g["meta_" + k]) for k in ("width", "height", "seed", "spp", "depth")}
    return g, meta


SCENES = {
    "c0_seed1": ("c0", 48, lambda gd, m: O.Scene.c0(m["width"], m["height"], m["seed"])),
    "c0_seed7": ("c0", 48, lambda gd, m: O.Scene.c0(m["width"], m["height"], m["seed"])),
    "mixed": ("mixed", 32, lambda gd, m: O.Scene.mixed(m["width"], m["height"], m["seed"])),
    "cornell": ("cornell", 48, lambda gd, m: O.Scene.cornell(m["width"], m["height"], m["seed"])),
    "smoke": ("smoke", 32, lambda gd, m: O.Scene.smoke(m["width"], m["height"], m["seed"])),
    "features": ("features", 32, lambda gd, m: O.Scene.features(m["width"], m["height"], m["seed"])),
    # wrappers nested every way, volumes around boxes, groups used as objects, an instance inside an instance (ref_harness.cpp build_nested)
    "nested": ("nested", 32, lambda gd, m: O.Scene.nested(m["width"], m["height"], m["seed"])),
    "obj": ("obj", 32, lambda gd, m: O.obj_scene(os.path.join(gd, "mesh_scene.obj"), m["width"], m["height"], m["seed"])),
    "ngon": ("obj", 32, lambda gd, m: O.obj_scene(os.path.join(gd, "ngon_scene.obj"), m["width"], m["height"], m["seed"])),
}


def test_glibc_rand_clone(golden_dir):
    """util.hpp:30-33 draws from glibc rand(); scene + BVH topology depend on that exact stream."""
    g = np.load(os.path.join(golden_dir, "rand.npz"))["rand_stream"]
    for i, seed in enumerate((1, 12345, 0)):
        assert np.array_equal(O.glibc_rand(seed, 400), g[400 * i:400 * (i + 1)])
    assert list(O.glibc_rand(1, 3)) == [1804289383, 846930886, 1681692777]


@pytest.mark.parametrize("name", list(SCENES))
def test_scene_bvh_firsthit_image(golden_dir, name):
    tag, side, make = SCENES[name]
    g, m = load(golden_dir, name)
    s = make(golden_dir, m)
    # scene generation + BVH builder (demo.hpp:17-60, bvh.hpp:22-56): pre-order dump incl. node boxes, leaf params, materials
    assert_bits(s.tree_rows(), g[tag + "_tree"], "tree")
    assert_bits(s.light_rows(), g[tag + "_lights"], "lights")
    assert_bits(s.camera_fields(), g[tag + "_camera"], "camera")
    # camera::emit (camera.hpp:62-68): also keeps the rand() stream in step with the harness, which emits these rays first
    assert_bits(s.primary_rays(side), g[tag + "_fh_rays"].reshape(-1, 7)[:side * side], "primary rays")
    # closest hit + traversal order (group.hpp:68-83, bvh.hpp:58-68, aabb.hpp:23-41, leaf intersects); a volume on the way draws
    # its free-flight distance from rand() inside intersect (primitive.hpp:373), in the harness' walk and here alike
    fh = s.first_hit(g[tag + "_fh_rays"])
    for k in ("entry", "nbox", "nleaf", "front", "hash"):
        assert np.array_equal(fh[k], g[f"{tag}_fh_{k}"]), k
    assert_bits(fh["t"], g[tag + "_fh_t"], "t")
    assert_bits(fh["rec"], g[tag + "_fh_rec"], "hit record")
    # prim ids agree with the entry -> prim id table
    ids = s.entry_prim_ids()
    hit = fh["entry"] >= 0
    inst = hit.copy(); inst[hit] = ids[fh["entry"][hit]] < 0     # the hit entry is an instance (translate / rotatey around a bvh): the
    plain = hit & ~inst                                          # harness reports the wrapper, the oracle the sphere inside it
    assert np.array_equal(ids[fh["entry"][plain]], fh["prim"][plain])
    assert np.all(fh["prim"][inst] >= 0)
    if tag + "_fh_prim" in g:      # the fixture names the LEAF behind every hit (by the hit record's material), also inside wrappers, groups, instances
        assert np.array_equal(fh["prim"], g[tag + "_fh_prim"]), "leaf hit"
    # full render: trace (raytracing.cpp:30-70) + pixel loop / NaN rejection / tonemap (raytracing.cpp:96-123)
    total, rgb, counters = s.render(0, m["spp"], m["depth"], O.RNG_GLIBC)
    assert_bits(total * (1 / m["spp"]), g[tag + "_img_radiance"], "radiance")
    assert np.array_equal(rgb.ravel(), g[tag + "_img_u8"])
    assert [int(x) for x in counters[:3]] == [int(x) for x in g[tag + "_img_counters"]], "rays / rejected samples / rand() position"
    s.free()


def test_unit_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "units.npz"))
    assert np.array_equal(O.unit_aabb(g["unit_aabb_in"]), g["unit_aabb_out"])        # aabb.hpp:23-41 incl. inf/NaN slabs
    out, pdfs = O.unit_prim(g["unit_prim_in"])
    assert_bits(out, g["unit_prim_out"], "primitive records")                         # primitive.hpp / polygon.hpp intersects
    assert_bits(pdfs, g["unit_prim_pdf"], "light pdfs")                               # primitive.hpp:258-269, polygon.hpp:140-151
    assert_bits(O.unit_mat(g["unit_mat_in"]), g["unit_mat_out"], "materials")         # material.hpp shading/pdf/emit
    cam, fields = O.unit_cam(64)
    assert_bits(cam, g["unit_cam_out"], "camera emit")
    assert_bits(fields, g["unit_cam_fields"], "camera fields")


def test_counter_rng_is_a_function_of_its_key():
    a = O.counter_rand(42, 7, 3, 16)
    assert np.array_equal(a, O.counter_rand(42, 7, 3, 16))
    assert not np.array_equal(a, O.counter_rand(42, 7, 4, 16))
    assert not np.array_equal(a, O.counter_rand(43, 7, 3, 16))
    u = np.concatenate([O.counter_rand(1, p, 0, 64) for p in range(256)]).astype(np.float64) / 2**32
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005


def test_counter_render_thread_invariant():
    """COUNTER mode keys each (pixel, sample) stream, so threads and sample-range splits cannot change the sums."""
    s = O.Scene.c0(40, 28, 1)
    a, _, ca = s.render(0, 4, 20, O.RNG_COUNTER, seed=9, n_threads=1)
    b, _, cb = s.render(0, 4, 20, O.RNG_COUNTER, seed=9, n_threads=4)
    assert_bits(a, b, "threads")
    c0, _, _ = s.render(0, 2, 20, O.RNG_COUNTER, seed=9, n_threads=2)
    c1, _, _ = s.render(2, 4, 20, O.RNG_COUNTER, seed=9, n_threads=2)
    assert np.allclose(c0 + c1, a, rtol=1e-14, atol=0)
    assert int(ca[0]) == int(cb[0])
    s.free()
