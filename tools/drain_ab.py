#!/usr/bin/env python3
"""A/B of the cost-ordered launch (RT_TILE_ORDER, rtcore.hip): kernel time of the same frames with the tiles launched in image order
and dearest first, at the sample counts where a launch's tail shows (a rank of an 8-GPU frame renders 128 spp; previews render 16).
    python tools/drain_ab.py [--mode fp64] [--level 8]       prints one line per (scene, spp, order) and a JSON summary"""
import argparse
import json
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
ap.add_argument("--out", default="")
ap.add_argument("--only", default="", help="comma-separated scene names")
args = ap.parse_args()
w, h = args.width, args.height
MODE = R.MODE_FAST_FP64 if args.mode == "fp64" else R.MODE_FAST_FP32
from raytracing_amd.tools.make_mesh import write_scene
path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"rt_blob_l{args.level}.obj")
write_scene(path, args.level)
builders = {"obj": lambda: HostScene.obj(path, w, h, 1, cam_pos=(0, 1.8, 5.0), cam_lookat=(0, 1.0, 0), fov=40.0),
            "random_spheres": lambda: HostScene.random_spheres(w, h, 1), "deep_bounce": lambda: HostScene.deep_bounce(w, h, 1),
            "cornell": lambda: HostScene.cornell(w, h)}
plan = [("obj", 50, (16, 128, 512)), ("random_spheres", 50, (16, 128)), ("deep_bounce", 200, (16, 128)), ("cornell", 10, (16,))]
acc = R.DeviceBuffer(w * h * 3 * 8)
out = {}
for name, depth, spps in plan:
    if args.only and name not in args.only.split(","):
        continue
    hs = builders[name]()
    for order in ("0", "1"):
        os.environ["RT_TILE_ORDER"] = order
        sc = hs.upload()
        for spp in spps:
            sc.render(acc.ptr, w, h, 0, spp, depth, MODE, seed=1)                  # warm-up; with the order on it is the render that counts rays per unit
            runs = [sc.render(acc.ptr, w, h, 0, spp, depth, MODE, seed=1) for _ in range(3)]
            best = min(runs, key=lambda r: r["kernel_ms"])
            traced = best["rays"] - best["untraced_rays"]
            out.setdefault(f"{name}:{spp}spp", {})["ordered" if order == "1" else "image_order"] = dict(kernel_ms=round(best["kernel_ms"], 3), Mrays_per_s=round(traced / best["kernel_ms"] / 1e3, 1))
            print(name, spp, "spp", "ordered" if order == "1" else "image order", f"{best['kernel_ms']:.2f} ms  {traced / best['kernel_ms'] / 1e3:.0f} Mrays/s", flush=True)
        sc.destroy()
    hs.free()
for k, v in out.items():
    v["time_ratio"] = round(v["ordered"]["kernel_ms"] / v["image_order"]["kernel_ms"], 4)
print(json.dumps(out))
if args.out:
    with open(args.out, "w") as f:
        json.dump(dict(mode=args.mode, width=w, height=h, results=out), f, indent=1)
