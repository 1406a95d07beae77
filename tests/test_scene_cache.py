"""Binary scene cache (rt_scene_desc_save / _load): CPU round trip and rejection of damaged files.
The GPU side (a cached scene renders like the original; checkpoint resume) is in test_gpu_parity.py."""
import os

import numpy as np
import pytest

import oracle_lib as O
import raytracing_amd as R


def test_cache_round_trip_is_verbatim(tmp_path, golden_dir):
    for make in (lambda: O.Scene.mixed(64, 48, 3), lambda: O.Scene.cornell(56, 56, 1),
                 lambda: O.obj_scene(os.path.join(golden_dir, "mesh_scene.obj"), 64, 48, 5),
                 lambda: O.Scene.features(48, 32, 4)):        # volumes, a data-less image texture, an instanced sub-tree
        osc = make()
        d = osc.export()
        path = str(tmp_path / "scene.rtsc")
        R.save_scene_cache(d, path)
        back = R.load_scene_cache(path)
        for k in ("node_bbox", "prim_data", "mat_data", "tex_data", "light_data", "camera"):
            a, b = np.ascontiguousarray(d[k], np.float64).ravel(), back[k].ravel()
            assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), k
        for k in ("node_child", "roots", "prim_kind", "prim_flags", "prim_mat", "mat_kind", "mat_tex", "tex_kind", "light_kind", "light_flags", "light_mat"):
            assert np.array_equal(np.asarray(d[k], np.int32).ravel(), back[k].ravel()), k
        if "xform_data" in d:
            assert np.array_equal(np.asarray(d["xform_data"]).ravel().view(np.uint64), back["xform_data"].ravel().view(np.uint64))
            assert np.array_equal(d["node_xform"], back["node_xform"]) and np.array_equal(d["prim_xform"], back["prim_xform"])
        else:
            assert back["xform_data"].size == 0
        osc.free()


def test_damaged_caches_are_errors(tmp_path):
    osc = O.Scene.mixed(32, 24, 3)
    path = str(tmp_path / "scene.rtsc")
    R.save_scene_cache(osc.export(), path)
    blob = open(path, "rb").read()
    bad = str(tmp_path / "bad.rtsc")
    for damaged in (blob[:len(blob) // 2],                      # truncated
                    b"NOTACACHE" + blob[9:],                    # wrong magic
                    blob + b"\0" * 8):                          # trailing bytes: size does not match the header
        open(bad, "wb").write(damaged)
        with pytest.raises(R.RtError):
            R.load_scene_cache(bad)
    # a child code pointing outside the node array must be caught by validation, not walked
    arr = bytearray(blob)
    d = osc.export()
    n_nodes = len(np.asarray(d["node_child"]).reshape(-1, 2))
    header = 8 + 8 * 4 + 8 + 24 * 8        # magic, eight int32 counts, n_pixel_bytes, camera (format RTSCENE2)
    off = header + n_nodes * 6 * 8            # first entry of node_child
    arr[off:off + 4] = np.int32(n_nodes + 5).tobytes()
    open(bad, "wb").write(bytes(arr))
    with pytest.raises(R.RtError):
        R.load_scene_cache(bad)
    with pytest.raises(R.RtError):
        R.load_scene_cache(str(tmp_path / "missing.rtsc"))
    osc.free()
