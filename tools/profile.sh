#!/bin/bash
# tools/profile.sh TAG [bench args...] -- run on the GPU box (via gpurun):
#   1. rocprofv3 --kernel-trace --stats of bench.py           -> gpurun_out/prof_TAG/trace
#   2. separate --pmc passes (never combined with trace domains other than kernel-trace)
# then tools/profile_summary.py condenses them into gpurun_out/prof_TAG/summary.{json,txt}
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu --no-sd --no-host-rate --file-reads 0 $@"   # (only the resident launches: per-launch averages must be of one size)
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
grep '^{' $OUT/trace.log > $OUT/bench_line.json
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 bench.py $ARGS > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i ($PMC) failed"; tail -3 $OUT/pmc$i.log; }
done
python3 tools/profile_summary.py $OUT
