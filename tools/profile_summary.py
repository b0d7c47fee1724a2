#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs of tools/profile.sh into summary.json / summary.txt (per kernel:
calls, avg ns, and every PMC counter averaged per launch)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
summary = {"kernels": {}, "counters": {}}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Name"].split("(")[0]
        summary["kernels"][name] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]),
                                    "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"]),
                                    "pct": float(row["Percentage"])}
for f in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for extra in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size"):
            if extra in row and row[extra] != "":
                acc[name][extra] = [float(row[extra])]
    for name, cs in acc.items():
        d = summary["counters"].setdefault(name, {})
        for c, vals in cs.items():
            d[c] = sum(vals) / len(vals)
try:
    summary["bench_line"] = json.loads(open(os.path.join(out, "bench_line.json")).read().strip().splitlines()[-1])
except Exception:
    pass
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
with open(os.path.join(out, "summary.txt"), "w") as f:
    for k, v in sorted(summary["kernels"].items(), key=lambda kv: -kv[1]["pct"]):
        f.write(f"{k:28s} calls={v['calls']:4d} avg={v['avg_ns'] / 1e6:10.4f} ms  {v['pct']:6.2f}%\n")
    for k, cs in summary["counters"].items():
        if "sk_scan_grid" not in k:
            continue
        f.write(f"\n[{k}] per-launch averages\n")
        for c, v in sorted(cs.items()):
            f.write(f"  {c:40s} {v:20.1f}\n")
        fs = cs.get("FETCH_SIZE")
        ws = cs.get("WRITE_SIZE")
        if fs is not None:
            f.write(f"  -> FETCH_SIZE KiB x1024 = {fs * 1024 / 1e9:.3f} GB raw (gfx950 tallies the 128-B requests of the streaming read at 64 B: "
                    f"tools/save_profile.py adds half of the record stream's bytes, for the streaming share only)\n")
        if ws is not None:
            f.write(f"  -> WRITE_SIZE = {ws * 1024 / 1e9:.3f} GB\n")
print(open(os.path.join(out, "summary.txt")).read())
