#!/usr/bin/env python3
"""rt_queue.h against rt_wave.h on the same streams: per-pixel ray counts (exact integers) and frames, both precisions."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_amd as R
from raytracing_amd.host_scenes import HostScene

w, h, spp, depth = 200, 136, 6, 50
for which in ("spheres", "obj"):
    if which == "obj":
        from raytracing_amd.tools.make_mesh import write_scene
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rt_blob_l5.obj")
        write_scene(path, 5)
        hs = HostScene.obj(path, w, h, 1, cam_pos=(0, 1.8, 5.0), cam_lookat=(0, 1.0, 0), fov=40.0)
    else:
        hs = HostScene.random_spheres(w, h, 1)
    sc = hs.upload()
    for mode, name in ((R.MODE_FAST_FP64, "fp64"), (R.MODE_FAST_FP32, "fp32")):
        for fl, fname in ((R.FLAG_DEBUG_RAY_COUNTS, "raycounts"), (0, "frame"), (R.FLAG_EARLY_EXIT, "frame-early")):
            a, _, sa = sc.render_to_host(w, h, 0, spp, depth, mode, seed=3, flags=fl)
            b, _, sb = sc.render_to_host(w, h, 0, spp, depth, mode, seed=3, flags=fl | R.FLAG_QUEUE)
            a = a.astype(np.float64); b = b.astype(np.float64)
            rel = np.abs(a - b) / np.maximum(np.abs(a), 1e-3)
            print(f"{which} {name} {fname}: rays {sa['rays']} vs {sb['rays']}, rejected {sa['rejected']} vs {sb['rejected']}, pixels differing {(rel.max(axis=2) > 1e-12).sum()} max rel {rel.max():.3e}", flush=True)
    sc.destroy(); hs.free()
