#!/bin/bash
# tools/exp_grid.sh TAG "<variants>" -- timing run of tools/exp_grid.py, then two bounded --pmc passes (TCC counters).
set -o pipefail
TAG=$1; VAR=$2; READS=${READS:-4000000}
OUT=gpurun_out/exp_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 python3 tools/exp_grid.py --reads $READS --variants "$VAR" > $OUT/timing.log 2>&1 || { echo "timing run failed"; tail -5 $OUT/timing.log; exit 1; }
cat $OUT/timing.log
i=0
for PMC in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 tools/exp_grid.py --preheat-ms 0 --reads $READS --variants "$VAR" > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $OUT/pmc$i.log; exit 1; }
  python3 tools/exp_grid_pmc.py $OUT/pmc$i $OUT/pmc$i.log | tee $OUT/pmc$i.txt
done
