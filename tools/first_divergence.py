#!/usr/bin/env python3
"""For every pixel of a small fp64 render whose sum differs from the CPU oracle's on the same streams: the first sample and ray at which the two
paths part, and the libm call (host glibc vs device ocml) whose last bits explain it.   python tools/first_divergence.py [c0_seed1|obj|cornell|mixed|smoke]
(GPU box; scenes and sizes as tests/test_gpu_parity.py::test_render_fp64_matches_oracle_same_streams)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import raytracing_amd as R
from divergence import explain_pixel

name = sys.argv[1] if len(sys.argv) > 1 else "c0_seed1"
spp, seed = 8, 11
golden = os.path.join(ROOT, "tests", "golden")
if name == "obj":
    osc, w, h, depth = O.obj_scene(os.path.join(golden, "mesh_scene.obj"), 96, 64, 1), 96, 64, 24
else:
    w, h, depth = 96, 64, 24
    osc = {"c0_seed1": O.Scene.c0, "cornell": O.Scene.cornell, "mixed": O.Scene.mixed, "smoke": O.Scene.smoke}[name](w, h, 1)
want, _, cnt = osc.render(0, spp, depth, O.RNG_COUNTER, seed=seed, n_threads=8, want_rgb=False)
sc = R.Scene(osc.export())
got, _, st = sc.render_to_host(w, h, 0, spp, depth, R.MODE_REFERENCE_FP64, seed=seed)
rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-3)
off = np.argwhere(rel.max(axis=2) > 1e-9)
print(f"{name}: {len(off)} of {w * h} pixels differ beyond 1e-9 relative; rays gpu {st['rays']} / cpu {int(cnt[0])}")
unexplained = 0
for i, j in off:
    e = explain_pixel(R, osc, sc, w, h, int(i), int(j), spp, seed, depth)
    unexplained += not e["explained"]
    head = f"  pixel ({i},{j}) sample {e['sample']}: {e['kind']} parts at ray {e['segment']}"
    if not e["calls"]:
        print(head + "  -- NO differing libm call: unexplained")
    for c in e["calls"][:3]:
        arg = f"{c['x']!r}" + (f", {c['y']!r}" if c["call"] in ("atan2", "pow") else "")
        print(head + f"  {c['call']}({arg}) = {c['cpu']!r} (glibc) vs {c['gpu']!r} (device), {c['ulps']:.0f} ulp")
print(f"{unexplained} unexplained")
sys.exit(1 if unexplained else 0)
