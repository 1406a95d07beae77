#!/usr/bin/env python3
"""Kernel throughput of the host-built demo scenes at one resolution (the SURVEY 8f rows next to the BASELINE workloads):
    python tools/scene_rates.py --width 3840 --height 2160 --spp 16 [--out profiles/x.json]
Prints one JSON object: per scene rays, kernel ms (best of 3 launches, HIP events inside rt_render), Mrays/s, rejected-sample share,
and box / leaf tests per ray from the counting kernel."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as R
from raytracing_amd.host_scenes import HostScene

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--out", default="")
ap.add_argument("--fixed", action="store_true", help="RT_FLAG_COMPAT_FIXED: delta specular lobes, phong sets wo (no counterpart in the reference)")
ap.add_argument("--mode", default="fp64", choices=["fp64", "fp32"], help="RT_MODE_FAST_FP64 (default) or RT_MODE_FAST_FP32")
args = ap.parse_args()
w, h = args.width, args.height
SCENES = (("random_spheres", 50, "BASELINE configs[1]"), ("deep_bounce", 200, "BASELINE configs[4]"), ("cornell", 10, "SURVEY 8f-1, the reference's main() scene at its max_depth"),
          ("smoke", 50, "SURVEY 8f-4: sphere-bounded media"), ("features", 50, "SURVEY 8f-4: all_feature_test()"))
FIX = R.FLAG_COMPAT_FIXED if args.fixed else 0
MODE = R.MODE_FAST_FP64 if args.mode == "fp64" else R.MODE_FAST_FP32
out = dict(width=w, height=h, spp=args.spp, mode="RT_MODE_FAST_FP64" if args.mode == "fp64" else "RT_MODE_FAST_FP32", scenes={}, compat="fixed" if args.fixed else "reference")
acc = R.DeviceBuffer(w * h * 3 * 8)
for name, depth, what in SCENES:
    hs = HostScene.cornell(w, h) if name == "cornell" else getattr(HostScene, name)(w, h, 1)
    sc = hs.upload()
    runs = [sc.render(acc.ptr, w, h, 0, args.spp, depth, MODE, seed=1, flags=FIX) for _ in range(3)]
    best = min(runs, key=lambda r: r["kernel_ms"])
    cnt = sc.render(acc.ptr, w, h, 0, 1, depth, MODE, seed=1, flags=R.FLAG_COUNT_TESTS | FIX)
    out["scenes"][name] = dict(what=what, max_depth=depth, n_primitives=int(hs.desc.contents.n_prims), rays=best["rays"], kernel_ms=round(best["kernel_ms"], 3),
                               Mrays_per_s=round(best["rays"] / best["kernel_ms"] / 1e3, 1), rejected_sample_share=round(best["rejected"] / (w * h * args.spp), 4),
                               box_tests_per_ray=round(cnt["box_tests"] / max(cnt["rays"], 1), 2), leaf_tests_per_ray=round(cnt["leaf_tests"] / max(cnt["rays"], 1), 2))
    print(name, out["scenes"][name], file=sys.stderr, flush=True)
    sc.destroy()
    hs.free()
print(json.dumps(out))
if args.out:
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
