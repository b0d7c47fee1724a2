#!/bin/bash
# tools/strain_size_sweep.sh -- device-resident scan rate against the size of the strain table (4 M reads of 150 bp,
# 2 % of them from the strain); run on the GPU box.  The filters grow with the table: this shows where they leave the L2.
for bp in 1000000 5000000 20000000 50000000 100000000; do python bench.py --no-cpu --no-host-rate --strain-bp $bp --reads 4000000 --steps 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('strain $bp bp (', d['config']['strain_keys'], 'keys ):', round(d['value']/1e9), 'Gbase/s,', round(d['ms_per_step'],3), 'ms per 0.6 Gbase pass')"; done
