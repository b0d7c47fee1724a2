#!/bin/bash
# tools/read_len_sweep.sh -- scan rate against the record length at 0.6 Gbase per pass (run on the GPU box): short reads,
# the usual 150, long reads, contigs of a genome list
for spec in "31 19000000" "50 12000000" "100 6000000" "150 4000000" "250 2400000" "1000 600000" "10000 60000"; do   # (records longer than the synthetic strain's 100 kb contigs: not from this generator)
  set -- $spec
  python bench.py --no-cpu --no-host-rate --read-len $1 --reads $2 --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('records of $1 bases:', round(d['value']/1e9), 'Gbase/s,', round(d['ms_per_step'],3), 'ms per pass,', d['config']['hits_per_pass_rank0_or_sum'], 'hits')"
done
