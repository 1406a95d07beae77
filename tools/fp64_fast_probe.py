#!/usr/bin/env python3
"""Probe RT_MODE_FAST_FP64 against RT_MODE_REFERENCE_FP64 on the fixture scenes (GPU): first hits and renders."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import raytracing_amd as R
from raytracing_amd.host_scenes import HostScene

gd = os.path.join(ROOT, "tests", "golden")
def load(name):
    g = np.load(os.path.join(gd, name + ".npz"))
    return g, {k: int(g["meta_" + k]) for k in ("width", "height", "seed", "spp", "depth")}
MAKE = {"c0_seed1": ("c0", lambda m: O.Scene.c0(m["width"], m["height"], m["seed"])), "c0_seed7": ("c0", lambda m: O.Scene.c0(m["width"], m["height"], m["seed"])),
        "mixed": ("mixed", lambda m: O.Scene.mixed(m["width"], m["height"], m["seed"])), "cornell": ("cornell", lambda m: O.Scene.cornell(m["width"], m["height"], m["seed"])),
        "obj": ("obj", lambda m: O.obj_scene(os.path.join(gd, "mesh_scene.obj"), m["width"], m["height"], m["seed"])),
        "features": ("features", lambda m: O.Scene.features(m["width"], m["height"], m["seed"])), "smoke": ("smoke", lambda m: O.Scene.smoke(m["width"], m["height"], m["seed"]))}
for q in ("0", "1"):
    os.environ["RT_QNODES"] = q
    for name, (tag, make) in MAKE.items():
        g, m = load(name)
        osc = make(m)
        d = osc.export()
        sc = R.Scene(d)
        rays = g[tag + "_fh_rays"].reshape(-1, 7)
        a = sc.first_hit(rays); b = sc.first_hit_tree(rays, True)
        bad = (a["prim"] != b["prim"]) | (a["t"].view(np.uint64) != b["t"].view(np.uint64)) | (a["front"] != b["front"])
        print(f"q={q} {name}: {len(rays)} rays, {int(bad.sum())} differ", flush=True)
        for i in np.nonzero(bad)[0][:8]:
            print("   ray", i, rays[i], "ref", a["prim"][i], a["t"][i], a["front"][i], "tree", b["prim"][i], b["t"][i], b["front"][i])
        if q == "1" or name in ("c0_seed7",):
            sc.destroy(); osc.free(); continue
        w, h = m["width"], m["height"]
        for early in (0, R.FLAG_EARLY_EXIT):
            t0 = time.time()
            ref, _, s0 = sc.render_to_host(w, h, 0, 8, min(m["depth"], 24), R.MODE_REFERENCE_FP64, seed=11, flags=early)
            got, _, s1 = sc.render_to_host(w, h, 0, 8, min(m["depth"], 24), R.MODE_FAST_FP64, seed=11, flags=early)
            rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)
            print(f"   render early={early}: rays {s0['rays']} vs {s1['rays']}, rejected {s0['rejected']} vs {s1['rejected']}, max rel {rel.max():.3e}, "
                  f"pixels > 1e-12: {int((rel.max(axis=2) > 1e-12).sum())} of {w*h}, ms {s0['kernel_ms']:.2f} vs {s1['kernel_ms']:.2f}", flush=True)
        sc.destroy(); osc.free()
