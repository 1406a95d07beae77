#!/usr/bin/env python3
"""Memory-side traffic of the render kernel per ray, from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE cannot share one).

    python tools/hbm_traffic.py <fetch_results.db> <write_results.db> <bench json line file> [--out profiles/hbm_traffic.json]

FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for 16-B-per-lane reads on gfx950.
The bench line gives the workload key and the rays per launch of the profiled command."""
import json
import sqlite3
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else "profiles/hbm_traffic.json"
if "--out" in sys.argv:
    args.remove(out_path)
round_tag = sys.argv[sys.argv.index("--round") + 1] if "--round" in sys.argv else "r03"
if "--round" in sys.argv:
    args.remove(round_tag)


def counter(db, name, kernel_sub):
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, sum(value) from counters_collection where counter_name = ? and kernel_name like ? group by dispatch_id order by dispatch_id",
                     (name, f"%{kernel_sub}%")).fetchall()
    return [v for _, v in rows]


bench = json.loads(open(args[2]).read().strip().splitlines()[-1])
kernel = bench["roofline"]["kernel"].split("<")[0].replace("rt::", "")
fetch = counter(args[0], "FETCH_SIZE", kernel)
write = counter(args[1], "WRITE_SIZE", kernel)
n = min(len(fetch), len(write))
rays = bench["roofline"]["rays_per_launch"]
# launches of the timed loop are the largest ones (the 1-spp probe launch is last and small): take the median of the full-size launches
full = sorted(zip(fetch[:n], write[:n]), key=lambda x: -x[0])[:max(1, bench["steps"])]
f_kb = sorted(x[0] for x in full)[len(full) // 2]
w_kb = sorted(x[1] for x in full)[len(full) // 2]
bytes_per_launch = f_kb * 1024 * 2 + w_kb * 1024
cfg = bench["config"]
mode = {"RT_MODE_FAST_FP64": "fp64", "RT_MODE_FAST_FP32": "fp32", "RT_MODE_FP32_REFERENCE_TREE": "fp32ref", "RT_MODE_REFERENCE_FP64": "fp64ref"}[cfg["mode"]]
workload = "soup" if "soup" in cfg["workload"] else ("obj" if "OBJ" in cfg["workload"] else "random_spheres")
key = f"{workload}:{cfg['n_primitives']}:{mode}:{cfg['width']}x{cfg['height']}:d{cfg['max_depth']}"
try:
    table = json.load(open(out_path))
except (OSError, ValueError):
    table = {}
table[key] = {"bytes_per_ray": bytes_per_launch / rays, "fetch_KB_per_launch": f_kb, "write_KB_per_launch": w_kb, "rays_per_launch": rays,
              "spp_profiled": cfg["spp"], "round": round_tag, "ray_count": "traced" if "reference_rays_per_step" in cfg else "reference",
              "note": "FETCH_SIZE x 2 + WRITE_SIZE; Infinity-Cache hits are counted by these memory-side counters; bytes_per_ray is per ray that walked the tree"}
json.dump(table, open(out_path, "w"), indent=1)
print(key, table[key])
