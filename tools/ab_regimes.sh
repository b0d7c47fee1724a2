#!/bin/bash
# tools/ab_regimes.sh "LIB LIB ..." [variants] -- tools/exp_grid.py over the strain-read regimes once per build of the library
# (build_exp/libsk_NAME.so from tools/exp_variant_build.sh, or "default"), all on the same box: Gbase/s per (build, regime)
LIBS=$1; VAR=${2:-base_h0,base_h2,base_h30,base_h100,div1_h100,div3_h100}
for L in $LIBS; do
  if [ $L = default ]; then unset SK_LIBRARY; else export SK_LIBRARY=$PWD/build_exp/libsk_$L.so; fi
  timeout -k 10 400 python3 tools/exp_grid.py --reads ${READS:-4000000} --variants "$VAR" 2>/dev/null | python3 -c "
import json,sys
r=[json.loads(l) for l in sys.stdin if l.startswith('{')]
print('$L', ' '.join('%s %.0f' % (x['variant'], x['gbase_s']) for x in r), 'hits', ' '.join(str(x['hits_per_pass']) for x in r))"
done
