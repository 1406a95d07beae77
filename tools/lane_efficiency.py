#!/usr/bin/env python3
"""SIMD lane utilisation of the fast render kernel, from its own counters (RT_FLAG_COUNT_TESTS).

    python tools/lane_efficiency.py [spheres|obj|deep] [--spp 4] [--level 8]

Prints, per max_depth, the fraction of issue slots that did useful work in the bounce loop, the node loop and the leaf
loop (lane tests / (64 x wave-level iterations)) and the timing of the non-counting kernel."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as R
from raytracing_amd.host_scenes import HostScene

ap = argparse.ArgumentParser()
ap.add_argument("which", nargs="?", default="spheres")
ap.add_argument("--spp", type=int, default=4)
ap.add_argument("--level", type=int, default=8)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--depths", default="1,2,50")
ap.add_argument("--variant", default="wave", choices=["wave", "legacy", "wavefront"], help="per-wave state machine (default), lane-synchronous loop, or the multi-kernel wavefront pipeline")
args = ap.parse_args()
w, h = args.width, args.height
if args.which == "obj":
    from raytracing_amd.tools.make_mesh import write_scene
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"rt_blob_l{args.level}.obj")
    write_scene(path, args.level)
    hs = HostScene.obj(path, w, h, 1, cam_pos=(0, 1.8, 5.0), cam_lookat=(0, 1.0, 0), fov=40.0)
elif args.which == "deep":
    hs = HostScene.deep_bounce(w, h, 1)
else:
    hs = HostScene.random_spheres(w, h, 1)
sc = hs.upload()
acc = R.DeviceBuffer(w * h * 3 * 4)
fl = {"wave": 0, "legacy": R.FLAG_LEGACY_LOOP, "wavefront": R.FLAG_WAVEFRONT}[args.variant]
for depth in [int(x) for x in args.depths.split(",")]:
    st = sc.render(acc.ptr, w, h, 0, args.spp, depth, R.MODE_FAST_FP32, seed=1, flags=R.FLAG_COUNT_TESTS | fl)
    t = sc.render(acc.ptr, w, h, 0, args.spp, depth, R.MODE_FAST_FP32, seed=1, flags=fl)
    t = sc.render(acc.ptr, w, h, 0, args.spp, depth, R.MODE_FAST_FP32, seed=1, flags=fl)
    r = st["rays"]
    print(f"{args.variant} {args.which} depth {depth}: rays {r}  {t['kernel_ms']:.2f} ms  {r / t['kernel_ms'] / 1e3:.0f} Mrays/s | "
          f"lane-eff bounce {r / (64 * st['wave_bounces']):.2f} node {st['box_tests'] / 2 / (64 * max(st['wave_node_steps'], 1)):.2f} "
          f"leaf {st['leaf_tests'] / (64 * max(st['wave_leaf_steps'], 1)):.2f} | per ray: nodes {st['box_tests'] / 2 / r:.1f} leaf tests {st['leaf_tests'] / r:.2f} "
          f"wave node steps x64 {st['wave_node_steps'] * 64 / r:.1f} wave leaf steps x64 {st['wave_leaf_steps'] * 64 / r:.1f}", flush=True)
