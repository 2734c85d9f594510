#!/bin/bash
# wall time of the workflow loop on <reads> reads for several host thread counts: scripts/threads_sweep.sh <reads> <t1> <t2> ...
n=$1; shift
d=$(mktemp -d)
python scripts/write_reads_db.py $n 60 150 $d/in || exit 1
export CDM_TIMING=1
for th in "$@"; do
  carpedeam_amd/carpedeam ancient_reads_loop $d/in $d/out --ancient-damage $d/in_dhigh --num-iter-reads-only 5 --num-iterations 12 --threads $th 2> $d/log
  echo "threads $th: $(grep 'Time for processing' $d/log)  host merge $(grep 'queues + extension' $d/log | awk '{s += $(NF-1)} END {print s}') s, other merge phases $(grep 'contig merge:' $d/log | grep -v 'queues' | awk '{s += $(NF-1)} END {print s}') s, device stages $(grep 'STEP' $d/log | sed 's/.*device stages: //' | tr -d ',a-z_)' | awk '{for (i = 1; i <= NF; i++) s += $i} END {print s / 1000}') s"
done
rm -rf $d
