#!/bin/bash
# tools/ab_bench.sh LIB_A LIB_B [bench args] -- bench.py's resident leg alternately with two builds of the library on the same box
# (SK_LIBRARY; build_exp/libsk_NAME.so from tools/exp_variant_build.sh, or "default"): ms_per_step and the scan kernel's ms of each run
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for L in $A $B; do
    if [ $L = default ]; then unset SK_LIBRARY; else export SK_LIBRARY=$PWD/build_exp/libsk_$L.so; fi
    timeout -k 10 300 python3 bench.py --no-cpu --no-sd --no-host-rate --file-reads 0 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$L', 'ms_per_step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline'].get('avg_launch_ms',0),4), 'Tbase/s', round(d['value']/1e12,3))"
  done
done
