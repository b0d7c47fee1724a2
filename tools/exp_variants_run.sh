#!/bin/bash
# tools/exp_variants_run.sh "name name ..." "<exp_grid variants>" -- tools/exp_grid.py once per build_exp/libsk_NAME.so (GPU box)
for n in $1; do
  echo "== $n"
  SK_LIBRARY=$PWD/build_exp/libsk_$n.so timeout -k 10 300 python3 tools/exp_grid.py --reads ${READS:-4000000} --variants "$2" 2>&1 | grep -v "^$" | tail -${TAILN:-8}
done
