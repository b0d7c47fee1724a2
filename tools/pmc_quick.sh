#!/bin/bash
# tools/pmc_quick.sh "<counters>" [bench args...] -- bounded rocprofv3 --pmc passes over bench.py, prints the per-launch averages
# of the scan kernel.  Any counter list is accepted: tools/pmc_run.py packs it into passes by hardware block and splits a pass
# the profiler still refuses (error 38 aborts rocprofv3 with signal 6: round 3 lost a run to that) instead of letting it end the call.
set -o pipefail
PMC=$1; shift
OUT=gpurun_out/pmcq_$(date +%H%M%S)_$$
mkdir -p $OUT
python3 tools/pmc_run.py $OUT "$PMC" -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-sd --no-host-rate --file-reads 0 "$@" > $OUT/averages.json
rc=$?
python3 - $OUT/averages.json <<'PY'
import json, sys
a = json.load(open(sys.argv[1]))
for k, cs in a.items():
    if "sk_scan_grid" in k:
        print(k.split("<")[0], {c: round(v / 1e6, 2) for c, v in cs.items()}, "(millions per launch)")
PY
exit $rc
