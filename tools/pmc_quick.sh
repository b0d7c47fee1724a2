#!/bin/bash
# tools/pmc_quick.sh "<counters>" [bench args...] -- one rocprofv3 --pmc pass over bench.py, prints the
# per-launch averages for sk_scan_main.
PMC=$1; shift
OUT=gpurun_out/pmcq_$$
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-host-rate "$@" > $OUT.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if ("sk_scan_grid" in r["Kernel_Name"] or "sk_scan_main" in r["Kernel_Name"]):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v) / len(v) / 1e6, 2) for k, v in acc.items()}, "(millions per launch)")
PY
