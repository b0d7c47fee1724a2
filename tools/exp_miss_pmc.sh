# usage (GPU box): bash tools/exp_miss_pmc.sh [NAME]   -- NAME = a variant built by tools/exp_variant_build.sh into build_exp/libsk_NAME.so
#                                                       (default: the library that ships)
export TMPDIR=/tmp
V=base_h0,l2hit_h2,nofilt_h2,base_h2
OUT=gpurun_out/exp_miss
mkdir -p $OUT
if [ -n "$1" ]; then
  export SK_LIBRARY=$PWD/build_exp/libsk_$1.so
  [ -f "$SK_LIBRARY" ] || { echo "no $SK_LIBRARY: run tools/exp_variant_build.sh $1 ... first"; exit 1; }
fi
i=0
for PMC in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 tools/exp_grid.py --preheat-ms 0 --variants "$V" > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $OUT/pmc$i.log; exit 1; }
  python3 tools/exp_grid_pmc.py $OUT/pmc$i $OUT/pmc$i.log | tee $OUT/pmc$i.txt
done
