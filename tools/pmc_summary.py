#!/usr/bin/env python3
"""Summarise a rocprofv3 results .db: per-kernel durations (kernel trace) and PMC sums.

    tools/pmc_summary.py <results.db> [kernel-name substring] [--per-dispatch]

--per-dispatch lists the counters of every matching dispatch in launch order instead of summing them per kernel."""
import json
import sqlite3
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
per_dispatch = "--per-dispatch" in sys.argv
db, sub = args[0], (args[1] if len(args) > 1 else "")
c = sqlite3.connect(db)
out = {"kernels": [], "pmc": {}}
try:
    for name, calls, total, avg, pct in c.execute("select * from top_kernels"):
        if sub in name:
            out["kernels"].append({"name": name, "calls": calls, "total_us": total, "avg_us": avg, "pct": pct})
except sqlite3.Error:
    pass
try:
    if per_dispatch:
        rows = {}
        for did, k, n, v, dur in c.execute("select dispatch_id, kernel_name, counter_name, sum(value), max(duration) from counters_collection "
                                           "group by dispatch_id, kernel_name, counter_name order by dispatch_id"):
            if sub in k:
                rows.setdefault(did, {"kernel": k, "duration_ns": dur})[n] = v
        out["pmc"] = [dict(dispatch=d, **r) for d, r in sorted(rows.items())]
    else:
        for k, n, v, cnt in c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
            if sub in k:
                out["pmc"].setdefault(k, {})[n] = v
                out["pmc"][k]["_dispatches"] = cnt
except sqlite3.Error:
    pass
print(json.dumps(out, indent=1))
