#!/usr/bin/env python3
"""Where the wave kernel's time goes, from its own counters (RT_FLAG_COUNT_TESTS): share of the wave time per phase (regenerate, node
loop, leaf step, shade), lane utilisation of the node loop and the leaf step, tests per traced ray.

    python tools/phase_shares.py [obj|spheres|deep|features|mixed|instanced|textured|smoke|cornell] [--mode fp64|fp32] [--spp 8] [--width 3840 --height 2160]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as R
from raytracing_amd.host_scenes import HostScene

ap = argparse.ArgumentParser()
ap.add_argument("which", nargs="?", default="obj")
ap.add_argument("--mode", default="fp64")
ap.add_argument("--spp", type=int, default=8)
ap.add_argument("--level", type=int, default=8)
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--depth", type=int, default=50)
args = ap.parse_args()
w, h = args.width, args.height
if args.which == "obj":
    from raytracing_amd.tools.make_mesh import write_scene
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"rt_blob_l{args.level}.obj")
    if not os.path.exists(path):
        write_scene(path, args.level)
    hs = HostScene.obj(path, w, h, 1, cam_pos=(0, 1.8, 5.0), cam_lookat=(0, 1.0, 0), fov=40.0)
elif args.which == "deep":
    hs = HostScene.deep_bounce(w, h, 1)
elif args.which in ("features", "mixed", "instanced", "textured", "smoke", "cornell"):
    hs = getattr(HostScene, "features_lit" if args.which == "features" else args.which)(w, h)
else:
    hs = HostScene.random_spheres(w, h, 1)
sc = hs.upload()
mode = R.MODE_FAST_FP64 if args.mode == "fp64" else R.MODE_FAST_FP32
acc = R.DeviceBuffer(w * h * 3 * (8 if args.mode == "fp64" else 4))
depth = 200 if args.which == "deep" else args.depth
st = sc.render(acc.ptr, w, h, 0, args.spp, depth, mode, seed=1, flags=R.FLAG_COUNT_TESTS)
t = sc.render(acc.ptr, w, h, 0, args.spp, depth, mode, seed=1)
t = sc.render(acc.ptr, w, h, 0, args.spp, depth, mode, seed=1)
traced = st["rays"] - st["untraced_rays"]
ticks = st["phase_ticks"]
tot = float(sum(ticks)) or 1.0
print(f"{sc.kernel_name(mode)}  {args.which} {w}x{h} {args.spp} spp depth {depth}: {t['kernel_ms']:.2f} ms, {traced / t['kernel_ms'] / 1e3:.0f} Mrays/s traced "
      f"({st['rays']} trace() calls, {traced} traced)")
print("  wave time: regenerate %.1f %%  node loop %.1f %%  leaf step %.1f %%  shade %.1f %%" % tuple(100 * x / tot for x in ticks))
per_node = 4 if ", true," in sc.kernel_name(mode) else 2
print(f"  per traced ray: box tests {st['box_tests'] / traced:.2f}  leaf tests {st['leaf_tests'] / traced:.2f}  "
      f"wave node steps x64 {64 * st['wave_node_steps'] / traced:.2f}  wave leaf rounds x64 {64 * st['wave_leaf_steps'] / traced:.2f}  shade passes x64 {64 * st['wave_bounces'] / st['rays']:.2f} per trace() call")
print(f"  lane utilisation: node loop (visits = box tests / {per_node} at most) <= {st['box_tests'] / per_node / (64 * max(st['wave_node_steps'], 1)):.2f}   "
      f"leaf rounds {st['leaf_tests'] / (64 * max(st['wave_leaf_steps'], 1)):.2f}   shade {st['rays'] / (64 * max(st['wave_bounces'], 1)):.2f}")
