#!/bin/bash
# inflate rate (1 thread, best of 3) of build variants of sk_gzfast.h on one .gz file -- run on the host that matters
# usage: tools/gz_variants.sh file.gz
f=$1
run() { name=$1; shift; gcc -O2 "$@" -o /tmp/gzv tools/gz_bench.c -lpthread || return; m=0; for i in 1 2 3; do v=$(/tmp/gzv $f 1 | tail -1 | sed 's/.*= \([0-9]*\) MB.*/\1/'); [ "$v" -gt "$m" ] && m=$v; done; echo "$name: $m MB/s of text"; }
run "as built (-O2)"
run "-O3" -O3
run "-mbmi2" -mbmi2
run "-march=native" -march=native
run "-O3 -march=native" -O3 -march=native
run "dist table 9 bits" -DSKZ_DIST_BITS=9
run "dist table 10 bits" -DSKZ_DIST_BITS=10
run "litlen table 10 bits" -DSKZ_LITLEN_BITS=10
run "litlen table 12 bits" -DSKZ_LITLEN_BITS=12
run "dist 10 + bmi2" -DSKZ_DIST_BITS=10 -mbmi2
