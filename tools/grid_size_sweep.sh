#!/bin/bash
# tools/grid_size_sweep.sh -- scan rate against the size of the grid kernel's level-1 filter for big strains (run on the GPU box)
run() { python bench.py --no-cpu --no-host-rate --strain-bp $1 --grid-kib $2 --reads 4000000 --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('strain $1 bp, level 1 = $2 KiB:', round(d['value']/1e9), 'Gbase/s')"; }
for k in 3072 6144 12288 24576 49152; do run 20000000 $k; done
for k in 3072 16384 65536 131072; do run 100000000 $k; done
for k in 1024 2048 3072 4096 8192; do run 5000000 $k; done
