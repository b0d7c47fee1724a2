#!/usr/bin/env python3
"""tools/exp_grid_pmc.py DIR LOG -- per-variant averages (millions per launch) of the PMC counters of the scan kernel
from a `rocprofv3 --kernel-trace --pmc` run of tools/exp_grid.py: the scan launches, whatever their template
arguments, are assigned to the variants in dispatch order; the first launch of a variant (warm-up) is dropped."""
import csv
import glob
import json
import sys
from collections import defaultdict

d, log = sys.argv[1], sys.argv[2]
names = [l for l in open(log) if l.startswith("ORDER ")][-1].split()[1].split(",")
rows = defaultdict(dict)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sk_scan_grid" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
per = len(ids) // len(names)
for vi, name in enumerate(names):
    acc = defaultdict(list)
    for i in ids[vi * per:(vi + 1) * per][1:]:
        for c, v in rows[i].items():
            acc[c].append(v)
    print(json.dumps({"variant": name, **{c: round(sum(v) / len(v) / 1e6, 3) for c, v in sorted(acc.items())}}))
