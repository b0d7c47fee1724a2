#!/bin/bash
# tools/pmc_quick.sh "<counters>" [bench args...] -- ONE bounded rocprofv3 --pmc pass over bench.py, prints the per-launch
# averages of the scan kernel.  The counter list is checked against the per-pass slots of gfx950 first
# (MI355X_MICROARCH.md "rocprofv3 PMC slots": SQ 8, TCC 4 with FETCH_SIZE costing 3 and WRITE_SIZE 2, GRBM 2): a list that does
# not fit is refused here instead of aborting inside rocprofv3 ("Request exceeds the capabilities of the hardware", as in
# round 1's gpurun_out/pmcq_458.log, which then sat idle for three GPU-minutes).  tools/profile.sh splits its passes the same way.
set -o pipefail
PMC=$1; shift
sq=0; tcc=0; grbm=0
for c in $PMC; do
  case $c in
    FETCH_SIZE) tcc=$((tcc+3));;
    WRITE_SIZE) tcc=$((tcc+2));;
    TCC_*|TCP_*) tcc=$((tcc+1));;
    SQ_*) sq=$((sq+1));;
    GRBM_*) grbm=$((grbm+1));;
  esac
done
if [ $sq -gt 8 ] || [ $tcc -gt 4 ] || [ $grbm -gt 2 ]; then
  echo "pmc_quick: counter list does not fit one pass (SQ $sq/8, TCC+TCP $tcc/4, GRBM $grbm/2): split it" >&2
  exit 2
fi
OUT=gpurun_out/pmcq_$$
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-host-rate --file-reads 0 "$@" > $OUT.log 2>&1
rc=$?
if [ $rc -ne 0 ]; then echo "pmc_quick: rocprofv3 failed (rc $rc)" >&2; tail -5 $OUT.log >&2; exit $rc; fi
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sk_scan_grid" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v) / len(v) / 1e6, 2) for k, v in acc.items()}, "(millions per launch)")
PY
