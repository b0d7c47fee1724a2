#!/bin/bash
# tools/exp_dense_pmc.sh TAG "<exp_grid variants>" -- where the dense regime (every read a strain read) spends a launch: timing run, then
# bounded --pmc passes (SQ issue/wait, LDS, TCC requests/misses/atomics) over tools/exp_grid.py, per-variant averages.  GPU box.
set -o pipefail
TAG=$1; VAR=$2; READS=${READS:-4000000}
OUT=gpurun_out/exp_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 python3 tools/exp_grid.py --reads $READS --variants "$VAR" > $OUT/timing.log 2>&1 || { echo "timing run failed"; tail -5 $OUT/timing.log; exit 1; }
cat $OUT/timing.log
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 tools/exp_grid.py --preheat-ms 0 --reads $READS --variants "$VAR" > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $OUT/pmc$i.log; continue; }
  python3 tools/exp_grid_pmc.py $OUT/pmc$i $OUT/pmc$i.log | tee $OUT/pmc$i.txt
done
