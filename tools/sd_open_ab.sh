#!/bin/bash
# A/B of strain_detect's opening (32 strains x 5 Mbp of strainer2_amd/cfg5.py, 200,000 reads): the key set built on the device (default)
# against the host's builder (SK_SD_HOST_KEYSET=1), for several numbers of opening threads.  GPU box: bash tools/sd_open_ab.sh
ROOT=$PWD
WORK=/dev/shm/sk_open_ab
python3 - <<PY
import sys
sys.path.insert(0, "$ROOT")
from strainer2_amd import cfg5
cfg5.write_all("$WORK", procs=16, prefix_reads=200000, only_prefix=True)
PY
cd $WORK
for t in 16 8 4 2; do
  for mode in device host; do
    if [ $mode = host ]; then export SK_SD_HOST_KEYSET=1; else unset SK_SD_HOST_KEYSET; fi
    for i in 1 2 3; do
      SK_THREADS=$t SK_SD_TIMING=1 $ROOT/strainer2_amd/bin/strain_detect -S strains_prefix.txt -b prefix.fa -t SE 2>&1 | grep -E "setup" | sed "s/^/threads $t $mode: /" | cut -c1-75
    done
  done
done
rm -rf $WORK
