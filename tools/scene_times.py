#!/usr/bin/env python3
"""Kernel time of the fast modes on the bench scenes at several sample counts, for A/B runs of library builds (RTCORE_LIB).
    python tools/scene_times.py [--mode fp64] [--only obj,random_spheres,deep_bounce,cornell,features,mixed,instanced,textured,smoke] [--spp 16,128]   one line per (scene, spp)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as R
from raytracing_amd.host_scenes import HostScene

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="fp64", choices=["fp64", "fp32"])
ap.add_argument("--level", type=int, default=8)
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--only", default="obj,random_spheres,deep_bounce,cornell")
ap.add_argument("--spp", default="16,128")
args = ap.parse_args()
w, h = args.width, args.height
MODE = R.MODE_FAST_FP64 if args.mode == "fp64" else R.MODE_FAST_FP32
from raytracing_amd.tools.make_mesh import write_scene
path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"rt_blob_l{args.level}.obj")
builders = {"obj": lambda: HostScene.obj(path, w, h, 1, cam_pos=(0, 1.8, 5.0), cam_lookat=(0, 1.0, 0), fov=40.0),
            "random_spheres": lambda: HostScene.random_spheres(w, h, 1), "deep_bounce": lambda: HostScene.deep_bounce(w, h, 1),
            "cornell": lambda: HostScene.cornell(w, h),
            # the general scene classes: all_feature_test() with a light, spheres + triangles + rects, the instanced Cornell room, image textures, media
            "features": lambda: HostScene.features_lit(w, h), "mixed": lambda: HostScene.mixed(w, h), "instanced": lambda: HostScene.instanced(w, h),
            "textured": lambda: HostScene.textured(w, h), "smoke": lambda: HostScene.smoke(w, h)}
depths = {"obj": 50, "random_spheres": 50, "deep_bounce": 200, "cornell": 10, "features": 50, "mixed": 50, "instanced": 10, "textured": 50, "smoke": 50}
acc = R.DeviceBuffer(w * h * 3 * 8)
for name in args.only.split(","):
    if name == "obj" and not os.path.exists(path):
        write_scene(path, args.level)
    sc = builders[name]().upload()
    for spp in [int(x) for x in args.spp.split(",")]:
        sc.render(acc.ptr, w, h, 0, min(spp, 16), depths[name], MODE, seed=1)          # warm-up; also the cost pass of the ordered launch
        runs = [sc.render(acc.ptr, w, h, 0, spp, depths[name], MODE, seed=1) for _ in range(2)]
        t = min(r["kernel_ms"] for r in runs)
        rays = runs[0]["rays"] - runs[0]["untraced_rays"]
        print(f"{name} [{sc.kernel_name(MODE)}] {args.mode} {spp} spp: {t:.2f} ms  {rays / t / 1e3:.0f} Mrays/s traced  rays {runs[0]['rays']} rejected {runs[0]['rejected']}", flush=True)
    sc.destroy()
