#!/bin/bash
# tools/hit_sweep.sh -- device-resident rate of the scan kernel against the fraction of reads drawn from the
# strain (4 M reads), and the build-kernel times; run on the GPU box.
for h in 0 0.02 0.1 0.3 1.0; do python bench.py --no-cpu --no-host-rate --hit-frac $h --reads 4000000 --steps 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('strain read fraction $h:', round(d['value']/1e9), 'Gbase/s,', round(d['ms_per_step'],3), 'ms per 0.6 Gbase pass')"; done
