#!/usr/bin/env python3
"""Copy the summaries of one tools/run_round_measurements.sh run from gpurun_out/<tag>/ into profiles/ (tracked) and refresh
profiles/hbm_traffic.json.    python tools/collect_profiles.py r02_d"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
summary = lambda db, sub="render_wave": json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), db, sub], text=True))
for name in ("bench_obj_1024", "bench_obj_64", "bench_spheres_256"):
    shutil.copy(os.path.join(src, name + ".json"), os.path.join(dst, f"{tag}_{name}.json"))
json.dump(summary(os.path.join(src, "trace", "trace_results.db")), open(os.path.join(dst, f"{tag}_kernel_trace_obj_4k_64spp.json"), "w"), indent=1)
def pmc_record(sq_db, ta_db, command, out_name):
    sq, ta = summary(sq_db), summary(ta_db)
    derived = {}
    for k, v in sq["pmc"].items():
        t = ta["pmc"].get(k, {})
        if "GRBM_GUI_ACTIVE" in t:
            derived[k] = {"valu_busy": v["SQ_ACTIVE_INST_VALU"] / (32 * t["GRBM_GUI_ACTIVE"]), "ta_busy": t["TA_TA_BUSY_sum"] / (32 * t["GRBM_GUI_ACTIVE"]),
                          "valu_insts_per_vmem_read": v["SQ_INSTS_VALU"] / max(v["SQ_INSTS_VMEM_RD"], 1), "wait_share_of_wave_cycles": v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"],
                          "waves_per_simd": 4 * v["SQ_WAVE_CYCLES"] / (128 * t["GRBM_GUI_ACTIVE"])}
    json.dump({"command": command, "sq": sq, "ta": ta, "derived": derived}, open(os.path.join(dst, out_name), "w"), indent=1)
    return derived


if os.path.exists(os.path.join(src, "sph_sq", "sq_results.db")):
    pmc_record(os.path.join(src, "sph_sq", "sq_results.db"), os.path.join(src, "sph_ta", "ta_results.db"),
               "rocprofv3 --pmc ... -- python3 bench.py --workload random_spheres --spp 16 --steps 1 --warmup 0 --no-cpu --no-fp32 (RT_MODE_FAST_FP64, 4K)",
               f"{tag}_pmc_wave_kernel_fp64_random_spheres_4k_16spp.json")
sq, ta = summary(os.path.join(src, "pmc_sq", "sq_results.db")), summary(os.path.join(src, "pmc_ta", "ta_results.db"))
derived = {}
for k, v in sq["pmc"].items():
    t = ta["pmc"].get(k, {})
    if "GRBM_GUI_ACTIVE" in t:
        # the SQ / TA counters of a dispatch are sums over the 32 CUs of one XCD; GRBM_GUI_ACTIVE is that XCD's busy cycles
        derived[k] = {"valu_busy": v["SQ_ACTIVE_INST_VALU"] / (32 * t["GRBM_GUI_ACTIVE"]), "ta_busy": t["TA_TA_BUSY_sum"] / (32 * t["GRBM_GUI_ACTIVE"]),
                      "valu_insts_per_vmem_read": v["SQ_INSTS_VALU"] / max(v["SQ_INSTS_VMEM_RD"], 1), "wait_share_of_wave_cycles": v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"],
                      # SQ_WAVE_CYCLES counts resident waves in units of four cycles; 128 SIMDs per XCD
                      "waves_per_simd": 4 * v["SQ_WAVE_CYCLES"] / (128 * t["GRBM_GUI_ACTIVE"])}
json.dump({"command": "rocprofv3 --pmc ... -- python3 bench.py --spp 16 --steps 1 --warmup 0 --no-cpu --no-fp32 (RT_MODE_FAST_FP64, obj workload 4K)", "sq": sq, "ta": ta,
           "derived": derived}, open(os.path.join(dst, f"{tag}_pmc_wave_kernel_fp64_obj_4k_16spp.json"), "w"), indent=1)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "hbm_traffic.py"), os.path.join(src, "pmc_fetch", "fetch_results.db"),
                       os.path.join(src, "pmc_write", "write_results.db"), os.path.join(src, "bench_obj_64.json"), "--out", os.path.join(dst, "hbm_traffic.json")])
print(json.dumps(derived, indent=1))
