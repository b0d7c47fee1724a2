#!/usr/bin/env python3
"""tools/exp_grid_pmc.py DIR LOG LAUNCHES -- per-variant averages of the PMC counters of the scan kernel from a
`rocprofv3 --kernel-trace --pmc` run of tools/exp_grid.py (launches are assigned to variants in order)."""
import csv
import glob
import json
import sys
from collections import defaultdict

d, log, L = sys.argv[1], sys.argv[2], int(sys.argv[3])
order = [l for l in open(log) if l.startswith("ORDER ")]
names = order[-1].split()[1].split(",") if order else []
rows = defaultdict(dict)                     # dispatch id -> counter -> value
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sk_scan_grid" in r["Kernel_Name"] or "sk_scan_main" in r["Kernel_Name"] or "sk_g" in r["Kernel_Name"]:
            rows[(int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
kernels = sorted({k for _, k in ids})
for kern in kernels:
    kid = [i for i in ids if i[1] == kern]
    per = len(kid) // max(len(names), 1) if names else L
    for vi, name in enumerate(names or ["all"]):
        mine = kid[vi * per:(vi + 1) * per][1:] or kid[vi * per:(vi + 1) * per]     # drop the warm-up launch
        acc = defaultdict(list)
        for i in mine:
            for c, v in rows[i].items():
                acc[c].append(v)
        print(json.dumps({"variant": name, "kernel": kern, **{c: round(sum(v) / len(v) / 1e6, 3) for c, v in sorted(acc.items())}}))
