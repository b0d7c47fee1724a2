#!/usr/bin/env python3
"""tools/trace_gaps.py KERNEL_TRACE.csv [NAME_SUBSTRING] -- the kernels of a rocprofv3 --kernel-trace in time order with their durations and
the idle gap before each (microseconds), for the last launches that match; says what a launch + collect is made of on the device side."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2] if len(sys.argv) > 2 else "sk_"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = []
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append((r["Kernel_Name"].split("(")[0][:60], (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0, r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
    prev_end = e
tail = [o for o in out if want in o[0]][-int(sys.argv[3]) if len(sys.argv) > 3 else -40:]
first = out.index(tail[0])
for name, dur, gap, grid in out[first:]:
    print(f"{name:60s} {dur:9.1f} us   gap before {gap:9.1f} us   grid {grid}")
