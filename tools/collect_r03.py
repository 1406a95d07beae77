#!/usr/bin/env python3
"""Summaries of one tools/run_r03_measurements.sh run -> profiles/ (tracked).   python tools/collect_r03.py <gpurun_out tag> <profiles tag, e.g. r03_b>

From the PMC passes of `bench.py --spp 64` (sq: instruction counts and wave cycles, ta: texture-address busy and GRBM cycles, st: wave stall
counters; `_sorted`: the same over the A/B library with round 2's sorted visit) it derives, per render kernel:
  valu_busy, ta_busy            share of cycles the vector ALUs / texture-address units are busy
  waves_per_simd                resident waves (SQ_WAVE_CYCLES counts in units of four cycles; 128 SIMDs per XCD, counters are per XCD)
  valu_insts_per_ray, valu_lane_slots_per_ray   SQ_INSTS_VALU (wave instructions, summed over the 8 XCD records of a dispatch) x 64 / traced rays
  wait_any_share                SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of resident-wave cycles spent waiting on anything
Rays come from the bench line of the same command (config.rays_per_step = rays that walked the tree)."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", sys.argv[1])
tag = sys.argv[2]
dst = os.path.join(ROOT, "profiles")


def summary(name, sub="render_wave"):
    db = os.path.join(src, name, f"{name}_results.db")
    if not os.path.exists(db):
        return None
    return json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), db, sub], text=True))


def bench_line(name):
    p = os.path.join(src, name)
    if not os.path.exists(p):
        return None
    txt = open(p).read().strip()
    return json.loads(txt.splitlines()[-1]) if txt else None


def derive(sq, ta, st, bench, launches):
    """per kernel name -> derived figures; `launches`: full-size launches of the profiled command (warm-up + steps), the 1-spp probe is the counting twin"""
    out = {}
    for k, v in (sq or {"pmc": {}})["pmc"].items():
        t = (ta or {"pmc": {}})["pmc"].get(k, {})
        s = (st or {"pmc": {}})["pmc"].get(k, {})
        d = {}
        if "GRBM_GUI_ACTIVE" in t:
            d.update(valu_busy=v["SQ_ACTIVE_INST_VALU"] / (32 * t["GRBM_GUI_ACTIVE"]), ta_busy=t["TA_TA_BUSY_sum"] / (32 * t["GRBM_GUI_ACTIVE"]),
                     waves_per_simd=4 * v["SQ_WAVE_CYCLES"] / (128 * t["GRBM_GUI_ACTIVE"]))
        d["valu_insts_per_vmem_read"] = v["SQ_INSTS_VALU"] / max(v["SQ_INSTS_VMEM_RD"], 1)
        if s.get("SQ_WAVE_CYCLES"):
            d["wait_any_share"] = s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"]
            d["wait_inst_share"] = s["SQ_WAIT_INST_ANY"] / s["SQ_WAVE_CYCLES"]
            d["issue_valu_share"] = s["SQ_ACTIVE_INST_VALU"] / s["SQ_WAVE_CYCLES"] / 4      # ACTIVE_INST counts cycles, WAVE_CYCLES units of four
        if bench and ", false," in k:
            rays = bench["config"]["rays_per_step"] * launches
            d["traced_rays_profiled"] = rays
            d["valu_insts_per_ray"] = v["SQ_INSTS_VALU"] / rays
            d["valu_lane_slots_per_ray"] = 64 * v["SQ_INSTS_VALU"] / rays
        out[k] = d
    return out


records = {}
for suffix, bench_name, label in (("", "bench_obj_64.json", "obj"), ("_sorted", "bench_obj_64_sorted.json", "obj_sorted_visit")):
    sq, ta, st = summary("sq" + suffix), summary("ta" + suffix), summary("st" + suffix)
    if not sq:
        continue
    b = bench_line(bench_name)
    derived = derive(sq, ta, st, b, launches=(b["steps"] + b["warmup"]) if b else 3)
    name = f"{tag}_pmc_wave_kernel_fp64_{label}_4k_64spp.json"
    json.dump({"command": "rocprofv3 --pmc <one counter set per run> -- python3 bench.py --spp 64 --no-cpu --no-fp32 --image-width 0 (RT_MODE_FAST_FP64, mesh workload 4K)"
                          + (" with RTCORE_LIB=build/librtcore_sorted.so (make EXTRA=-DRT_SORTED_VISIT: round 2's distance-sorted visit)" if suffix else ""),
               "bench_line_kernel_ms": b["roofline"]["kernel_ms_avg"] if b else None, "sq": sq, "ta": ta, "st": st, "derived": derived}, open(os.path.join(dst, name), "w"), indent=1)
    records[label] = {k: v for k, v in derived.items() if ", false," in k}
sq, ta, st = summary("sph_sq"), summary("sph_ta"), summary("sph_st")
if sq:
    b = bench_line("bench_spheres_64.json")
    derived = derive(sq, ta, st, b, launches=(b["steps"] + b["warmup"]) if b else 3)
    json.dump({"command": "rocprofv3 --pmc ... -- python3 bench.py --workload random_spheres --spp 64 --no-cpu --no-fp32 --image-width 0 (RT_MODE_FAST_FP64, 4K)",
               "bench_line_kernel_ms": b["roofline"]["kernel_ms_avg"] if b else None, "sq": sq, "ta": ta, "st": st, "derived": derived},
              open(os.path.join(dst, f"{tag}_pmc_wave_kernel_fp64_random_spheres_4k_64spp.json"), "w"), indent=1)
    records["random_spheres"] = {k: v for k, v in derived.items() if ", false," in k}
for name in ("bench_obj_1024", "bench_obj_64", "bench_spheres_256", "bench_obj_64_sorted", "bench_spheres_64"):
    if os.path.exists(os.path.join(src, name + ".json")) and os.path.getsize(os.path.join(src, name + ".json")):
        shutil.copy(os.path.join(src, name + ".json"), os.path.join(dst, f"{tag}_{name}.json"))
tr = os.path.join(src, "trace", "trace_results.db")
if os.path.exists(tr):
    json.dump(json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), tr, "render_wave"], text=True)),
              open(os.path.join(dst, f"{tag}_kernel_trace_obj_4k_64spp.json"), "w"), indent=1)
if os.path.exists(os.path.join(src, "fetch", "fetch_results.db")) and os.path.exists(os.path.join(src, "bench_obj_64.json")):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "hbm_traffic.py"), os.path.join(src, "fetch", "fetch_results.db"),
                           os.path.join(src, "write", "write_results.db"), os.path.join(src, "bench_obj_64.json"), "--out", os.path.join(dst, "hbm_traffic.json"), "--round", tag])
print(json.dumps(records, indent=1))
