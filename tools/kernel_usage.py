#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stderr of a compile): one line per kernel with
VGPRs, AGPRs, scratch bytes per lane (spills), occupancy and LDS.  Usage: tools/kernel_usage.py usage.txt [filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
for b in blocks:
    name = b.split("\n", 1)[0].split(" [")[0]
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    if flt and flt not in name:
        continue
    def get(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    short = re.sub(r"\(rt::Scene.*", "", name)
    v, ag, sc, oc, lds, sg = (get("VGPRs"), get("AGPRs"), get(r"ScratchSize \[bytes/lane\]"), get(r"Occupancy \[waves/SIMD\]"),
                              get(r"LDS Size \[bytes/block\]"), get("SGPRs"))
    print(f"{short:70s} vgpr {v:4d} agpr {ag:4d} scratch {sc:5d} occ {oc:2d} lds {lds:6d} sgpr {sg:4d}")
