#!/usr/bin/env python3
"""tools/pmc_run.py OUTDIR "<counters>" -- [program args...]   (GPU box)

Collect any list of PMC counters over a program in as many rocprofv3 passes as it takes.  The list is packed into passes by
hardware block (gfx950: SQ 8 slots, TCC 4 with FETCH_SIZE costing 3 and WRITE_SIZE 2, TCP 4, TA 2, TD 2, GRBM 2 --
MI355X_MICROARCH.md "rocprofv3 PMC slots"; the TCP/TA/TD limits are this tool's conservative guess), and a pass that rocprofv3
still refuses ("error code 38: Request exceeds the capabilities of the hardware to collect", which ABORTS the profiler with
signal 6 -- gpurun_out/pmcq_345.log of round 3) is split in two and tried again instead of ending the run.  Every pass is
`rocprofv3 --kernel-trace --pmc ...` only (never combined with other trace domains), bounded by a timeout, in a directory
of its own: OUTDIR/pass<N>.  Prints nothing but a JSON object {kernel name: {counter: average per launch}}; exit status 0
if every counter was collected."""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

LIMITS = {"SQ": 8, "TCC": 4, "TCP": 4, "TA": 2, "TD": 2, "GRBM": 2, "OTHER": 4}
MAX_PER_PASS = 8                       # counters of all blocks together (the passes of tools/profile.sh that are known to work have at most 8)
COST = {"FETCH_SIZE": ("TCC", 3), "WRITE_SIZE": ("TCC", 2)}


def block_of(c):
    if c in COST:
        return COST[c]
    for b in ("SQ", "TCC", "TCP", "TA", "TD", "GRBM"):
        if c.startswith(b + "_"):
            return b, 1
    return "OTHER", 1


def pack(counters):
    passes = []
    for c in counters:
        b, cost = block_of(c)
        for p in passes:
            if p["use"][b] + cost <= LIMITS[b] and len(p["list"]) < MAX_PER_PASS:
                p["use"][b] += cost
                p["list"].append(c)
                break
        else:
            passes.append({"use": defaultdict(int, {b: cost}), "list": [c]})
    return [p["list"] for p in passes]


def main():
    out, counters = sys.argv[1], sys.argv[2].split()
    prog = sys.argv[sys.argv.index("--") + 1:]
    todo, n, failed = pack(counters), 0, []
    result = defaultdict(lambda: defaultdict(list))
    env = dict(os.environ, TMPDIR="/tmp")
    while todo:
        lst = todo.pop(0)
        n += 1
        d = os.path.join(out, f"pass{n}")
        os.makedirs(d, exist_ok=True)
        log = os.path.join(d, "run.log")
        with open(log, "w") as lf:
            lf.write("counters: " + " ".join(lst) + "\n")
            lf.flush()
            try:
                rc = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + lst + ["--output-format", "csv", "-d", d, "--"] + prog,
                                    stdout=lf, stderr=subprocess.STDOUT, env=env, timeout=int(os.environ.get("PMC_PASS_TIMEOUT", "420"))).returncode
            except subprocess.TimeoutExpired:
                rc = -999
        text = open(log, errors="replace").read()
        if rc != 0:
            if "error code 38" in text and len(lst) > 1:              # does not fit one pass after all: split, try again
                h = len(lst) // 2
                todo = [lst[:h], lst[h:]] + todo
                print(f"pmc_run: pass {n} ({' '.join(lst)}) exceeds the hardware's slots: split in two", file=sys.stderr)
                continue
            print(f"pmc_run: pass {n} ({' '.join(lst)}) failed with status {rc}; see {log}", file=sys.stderr)
            failed += lst
            if rc == -999:
                break                                                 # a pass that timed out: no further GPU step in this call
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                result[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    avg = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in result.items()}
    json.dump(avg, sys.stdout, indent=1)
    print()
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
