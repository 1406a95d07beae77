#!/usr/bin/env python3
"""Per-pixel ray counts of the fast mode over different builds of its tree: binary 64-byte vs quantised four-wide nodes, default vs squeezed depth / stack budget.

    python tools/tree_variants_check.py [--tris 300000]

The closest hit does not depend on the tree, so with the same streams every build must trace the same paths; what may differ is the
winner between two triangles met at exactly the same t (shared edges), which the visit order decides.  Scenes: the generated mesh
(level 6, 82 k triangles, coordinates of a few units) and a triangle soup in [-50, 50]^3 (edge 0.2: boxes 2 500 x smaller than
the coordinates, the stress case for the quantised boxes' padding)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as R                                     # noqa: E402
from raytracing_amd.host_scenes import HostScene               # noqa: E402
from raytracing_amd.tools.make_mesh import write_scene, write_soup   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=300000)
args = ap.parse_args()
tmp = os.environ.get("TMPDIR", "/tmp")
w, h, spp, depth = 640, 360, 4, 50
mesh, soup = os.path.join(tmp, "tvc_mesh.obj"), os.path.join(tmp, "tvc_soup.obj")
write_scene(mesh, 6)
write_soup(soup, args.tris, 8, 64)
scenes = {"mesh": dict(path=mesh), "soup": dict(path=soup, cam_pos=(0, 5.0, 75.0), cam_lookat=(0, 0, 0), fov=40.0)}
bad = 0
for name, kw in scenes.items():
    path = kw.pop("path")
    hs = HostScene.obj(path, w, h, 1, **kw)
    base = None
    for label, env in (("64-byte nodes", {"RT_QNODES": "0"}), ("wide nodes", {"RT_QNODES": "1"}), ("wide, depth 20", {"RT_QNODES": "1", "RT_BVH_DEPTH": "20"}), ("wide, stack budget 22", {"RT_QNODES": "1", "RT_WIDE_STACK": "22"}),
                       ("wide, leaf 8", {"RT_QNODES": "1", "RT_BVH_LEAF": "8"})):
        for k in ("RT_QNODES", "RT_BVH_DEPTH", "RT_BVH_LEAF", "RT_WIDE_STACK"):
            os.environ.pop(k, None)
        os.environ.update(env)
        sc = hs.upload()
        cnt, _, st = sc.render_to_host(w, h, 0, spp, depth, R.MODE_FAST_FP32, seed=7, flags=R.FLAG_DEBUG_RAY_COUNTS | R.FLAG_COUNT_TESTS)
        sc.destroy()
        if base is None:
            base = cnt
        diff = int((cnt[..., 0] != base[..., 0]).sum())
        print(f"{name:5s} {label:18s} rays {st['rays']:9d} rejected {st['rejected']:6d} box tests/ray {st['box_tests'] / st['rays']:6.2f} "
              f"leaf tests/ray {st['leaf_tests'] / st['rays']:5.2f} pixels != 64-byte build: {diff} of {w * h}", flush=True)
        bad += diff > 0.002 * w * h
    hs.free()
sys.exit(1 if bad else 0)
