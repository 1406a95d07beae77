#!/usr/bin/env python3
"""Sweep the wave state machine's phase thresholds (RT_TRAV_KEEP / RT_NODE_KEEP / RT_REGEN_MIN, read at rt_scene_create).
    python tools/sweep_wave.py [spheres|obj|deep] --spp 16 --depth 50 --trav 64,128,192 --node 1,64,128 --regen 1,8"""
import argparse
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as R
from raytracing_amd.host_scenes import HostScene

ap = argparse.ArgumentParser()
ap.add_argument("which", nargs="?", default="spheres")
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--depth", type=int, default=50)
ap.add_argument("--level", type=int, default=8)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--trav", default="0", help="0 = library default")
ap.add_argument("--node", default="0")
ap.add_argument("--regen", default="0")
ap.add_argument("--count", action="store_true")
ap.add_argument("--mode", default="fp32", choices=["fp32", "fp64"], help="RT_MODE_FAST_FP32 or RT_MODE_FAST_FP64")
ap.add_argument("--flags", type=int, default=0, help="RT_FLAG_* bits (16 legacy loop, 32 wavefront pipeline)")
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
acc = R.DeviceBuffer(w * h * 3 * 8)
MODE = R.MODE_FAST_FP64 if args.mode == 'fp64' else R.MODE_FAST_FP32
ints = lambda s: [int(x) for x in s.split(",")]
for tk, nk, rm in itertools.product(ints(args.trav), ints(args.node), ints(args.regen)):
    for name, v in (("RT_TRAV_KEEP", tk), ("RT_NODE_KEEP", nk), ("RT_REGEN_MIN", rm)):
        os.environ.pop(name, None)
        if v:
            os.environ[name] = str(v)
    sc = hs.upload()
    sc.render(acc.ptr, w, h, 0, args.spp, args.depth, MODE, seed=1, flags=args.flags)
    runs = [sc.render(acc.ptr, w, h, 0, args.spp, args.depth, MODE, seed=1, flags=args.flags) for _ in range(2)]
    t = min(r["kernel_ms"] for r in runs)
    msg = f"{args.which} {args.mode} trav_keep {tk} node_keep {nk} regen_min {rm}: {t:.2f} ms"
    if args.count:
        st = sc.render(acc.ptr, w, h, 0, args.spp, args.depth, MODE, seed=1, flags=R.FLAG_COUNT_TESTS | args.flags)
        r = st["rays"]
        msg += (f"  {r / t / 1e3:.0f} Mrays/s | lane-eff shade {r / (64 * st['wave_bounces']):.2f} node {st['box_tests'] / 2 / (64 * max(st['wave_node_steps'], 1)):.2f} "
                f"leaf {st['leaf_tests'] / (64 * max(st['wave_leaf_steps'], 1)):.2f}")
        msg += f" | per ray: box {st['box_tests'] / r:.1f} leaf {st['leaf_tests'] / r:.2f} node steps/wave-ray {64 * st['wave_node_steps'] / r:.1f}"
        pt = st["phase_ticks"]
        if sum(pt):
            msg += " | wave time regen/node/leaf/shade " + "/".join(f"{100 * x / sum(pt):.0f}%" for x in pt)
    print(msg, flush=True)
    sc.destroy()
