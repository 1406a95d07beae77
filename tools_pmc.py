#!/usr/bin/env python3
"""Summarise a rocprofv3 results .db: per-kernel durations (kernel trace) and PMC sums.  usage: tools_pmc.py <results.db> [substr]"""
import json
import sqlite3
import sys

db, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
c = sqlite3.connect(db)
out = {"kernels": [], "pmc": {}}
try:
    for name, calls, total, avg, pct in c.execute("select * from top_kernels"):
        if sub in name:
            out["kernels"].append({"name": name, "calls": calls, "total_us": total, "avg_us": avg, "pct": pct})
except sqlite3.Error:
    pass
try:
    for k, n, v, cnt in c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
        if sub in k:
            out["pmc"].setdefault(k, {})[n] = v
            out["pmc"][k]["_dispatches"] = cnt
except sqlite3.Error:
    pass
print(json.dumps(out, indent=1))
