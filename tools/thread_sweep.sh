#!/bin/bash
# decode-thread sweep of kmer_scrub_count on a long .gz list (10 files x 1 M reads listed 35 times = 52.5 Gbase): wall, user, sys per SK_THREADS
set -e
cd $GRAFT_REPO_ROOT
WORK=/tmp/sk_cfg3c QUAL=binned NFILES=10 READS=1000000 KINDS=fq.gz THREADS=16 python tools/e2e_bench.py > gpurun_out/sweep_gen.txt 2>&1
python - <<'PY'
names = open("/tmp/sk_cfg3c/B_fq.gz.txt").read().split()
open("/tmp/sk_cfg3c/B35.txt", "w").write("\n".join(names * 35) + "\n")
PY
for t in ${SWEEP:-8 16 24 32 48}; do
  TIMEFORMAT="threads $t: %R s wall, %U user, %S sys"
  { time SK_THREADS=$t SK_TIMING=1 strainer2_amd/bin/kmer_scrub_count -r /tmp/sk_cfg3c/strain.fa -A /tmp/sk_cfg3c/A.txt -B /tmp/sk_cfg3c/B35.txt > /dev/null ; } 2>> gpurun_out/sweep.txt
done
grep -v "^key set" gpurun_out/sweep.txt
