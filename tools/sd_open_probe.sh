#!/bin/bash
# Where does the opening of 32 strains go?  HIP API time by function (rocprofv3 --hip-trace --stats) of one
# `strain_detect -S` run on a small metagenome (the inputs of strainer2_amd/cfg5.py, first 200,000 reads).
# usage (GPU box): bash tools/sd_open_probe.sh  -> gpurun_out/sd_open_probe/
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
WORK=/dev/shm/sk_open_probe
OUT=$ROOT/gpurun_out/sd_open_probe
mkdir -p $OUT
trap 'rm -rf $WORK' EXIT                 # (the inputs live in memory: they go whatever happens below)
python3 - <<PY
import sys
sys.path.insert(0, "$ROOT")
from strainer2_amd import cfg5
cfg5.write_all("$WORK", procs=16, prefix_reads=200000, only_prefix=True)
PY
cd $WORK
SK_SD_TIMING=1 $ROOT/strainer2_amd/bin/strain_detect -S strains_prefix.txt -b prefix.fa -t SE 2> $OUT/plain_timing.txt
cat $OUT/plain_timing.txt
export TMPDIR=/tmp
SK_LEAK_AT_EXIT=0 SK_SD_TIMING=1 rocprofv3 --hip-trace --kernel-trace --stats -d $OUT/prof -o sd -- $ROOT/strainer2_amd/bin/strain_detect -S strains_prefix.txt -b prefix.fa -t SE > $OUT/rocprof.log 2>&1 || true
ls $OUT/prof | head
find $OUT/prof -name "*hip_api_stats.csv" -exec head -30 {} \;
