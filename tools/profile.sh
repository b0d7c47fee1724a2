#!/bin/bash
# tools/profile.sh TAG [--sd] [bench args...] -- run on the GPU box (via gpurun):
#   1. rocprofv3 --kernel-trace --stats of bench.py           -> <run dir>/trace
#   2. separate --pmc passes (never combined with trace domains other than kernel-trace), tools/pmc_run.py
# then tools/profile_summary.py condenses them into <run dir>/summary.{json,txt}.
# EVERY run gets a directory of its own, gpurun_out/prof_TAG_<time>: gpurun merges what a call wrote into the builder's
# gpurun_out/, and round 3's profiles/r03_final_kernel_stats.csv came from an OLDER run of the same tag that was still lying in
# the merged directory.  tools/save_profile.py TAG takes the newest run of the tag and checks that the
# kernel-trace average it copies is the one in summary.json.
#   --sd: profile the strain_detect side leg instead (sk_scan_grid<TALLY,UNION>, 32 resident strains; bench.py --sd-only)
set -o pipefail
TAG=$1; shift
SD=0
if [ "$1" = "--sd" ]; then SD=1; shift; fi
OUT=gpurun_out/prof_${TAG}_$(date +%Y%m%d_%H%M%S)
mkdir -p $OUT
echo $OUT > gpurun_out/prof_${TAG}.latest
export TMPDIR=/tmp
if [ $SD = 1 ]; then
  ARGS="--sd-only --reads 200000 --steps 2 --warmup 1 --no-cpu --no-host-rate --file-reads 0 $@"
else
  ARGS="--steps 5 --warmup 1 --no-cpu --no-sd --no-host-rate --file-reads 0 $@"   # (only the resident launches: per-launch averages must be of one size)
fi
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
grep '^{' $OUT/trace.log > $OUT/bench_line.json
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;      # (one run, one file)
PMC="FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum \
 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU \
 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE \
 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_EA0_RDREQ_sum \
 TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum \
 TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
python3 tools/pmc_run.py $OUT/pmc "$PMC" -- python3 bench.py $ARGS > $OUT/pmc_averages.json || echo "some PMC passes failed (see $OUT/pmc/pass*/run.log)"
python3 tools/profile_summary.py $OUT
