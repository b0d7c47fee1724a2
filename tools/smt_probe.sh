#!/bin/bash
# does one core have room for a second inflate chain?  one gz_bench pinned to a CPU, then two pinned to that CPU and its
# SMT sibling: the ratio of the aggregate rates says how much an interleaved two-stream decoder could gain per core
f=$1
gcc -O2 -o /tmp/gzv tools/gz_bench.c -lpthread || exit 1
c=4
sib=$(cat /sys/devices/system/cpu/cpu$c/topology/thread_siblings_list)
echo "cpu $c siblings: $sib"
o=$(echo $sib | tr ',-' '  ' | awk '{print $2}')
one() { taskset -c $1 /tmp/gzv $f 1 | tail -1 | sed 's/.*= \([0-9]*\) MB.*/\1/'; }
echo "alone on cpu $c: $(one $c) MB/s"
one $c > /tmp/a.txt & one $o > /tmp/b.txt & wait
echo "two at once on cpus $c and $o: $(cat /tmp/a.txt) + $(cat /tmp/b.txt) MB/s"
one $c > /tmp/a.txt & one $((c+2)) > /tmp/b.txt & wait
echo "two at once on different cores $c and $((c+2)): $(cat /tmp/a.txt) + $(cat /tmp/b.txt) MB/s"
