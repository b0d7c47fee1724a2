#!/bin/bash
# where do the threads of a running kmer_scrub_count sleep?  (kernel wait channel of every thread, sampled a few times)
# usage: tools/wchan_sample.sh <SK_THREADS> <strain.fa> <A.txt> <B.txt>
SK_THREADS=$1 strainer2_amd/bin/kmer_scrub_count -r $2 -A $3 -B $4 > /dev/null &
pid=$!
sleep 2
for i in 1 2 3 4 5 6; do
  for t in /proc/$pid/task/*; do echo "$(cat $t/wchan 2>/dev/null) $(awk '{print $3}' $t/stat 2>/dev/null)"; done
  sleep 0.7
done | sort | uniq -c | sort -rn | head -12
wait $pid
