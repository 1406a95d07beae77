#!/usr/bin/env python3
"""Frames and per-pixel ray counts of the library RTCORE_LIB names, saved to an .npz; with two files, compares them.
usage: RTCORE_LIB=a.so tools/lib_frames.py a.npz ; RTCORE_LIB=b.so tools/lib_frames.py b.npz ; tools/lib_frames.py a.npz b.npz"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) == 3:
    a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
    for k in a.files:
        x, y = a[k].astype(np.float64), b[k].astype(np.float64)
        rel = np.abs(x - y) / np.maximum(np.abs(x), 1e-3)
        per = rel.reshape(rel.shape[0], rel.shape[1], -1).max(axis=2)
        print(f"{k}: pixels differing {(per > 1e-12).sum()} of {per.size}, max rel {rel.max():.3e}, sums {x.sum():.6e} {y.sum():.6e}")
    sys.exit(0)

import raytracing_amd as R
from raytracing_amd.host_scenes import HostScene
from raytracing_amd.tools.make_mesh import write_scene
w, h, spp, depth = 200, 136, 6, 50
out = {}
for which in ("spheres", "obj"):
    if which == "obj":
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rt_blob_l5.obj")
        write_scene(path, 5)
        hs = HostScene.obj(path, w, h, 1, cam_pos=(0, 1.8, 5.0), cam_lookat=(0, 1.0, 0), fov=40.0)
    else:
        hs = HostScene.random_spheres(w, h, 1)
    sc = hs.upload()
    for mode, name in ((R.MODE_FAST_FP64, "fp64"), (R.MODE_FAST_FP32, "fp32")):
        for fl, fname in ((R.FLAG_DEBUG_RAY_COUNTS, "raycounts"), (0, "frame")):
            f, _, st = sc.render_to_host(w, h, 0, spp, depth, mode, seed=3, flags=fl)
            out[f"{which}_{name}_{fname}"] = f
            print(which, name, fname, st["rays"], flush=True)
    sc.destroy(); hs.free()
np.savez(sys.argv[1], **out)
