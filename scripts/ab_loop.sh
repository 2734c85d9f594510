#!/bin/bash
# A/B of two library builds on the same box: the workflow loop on <reads> reads with the regular library and with the one in
# <variant dir> (LD_LIBRARY_PATH), alternating twice: scripts/ab_loop.sh <reads> <variant dir>
n=${1:-1000000}; var=$2
d=$(mktemp -d)
python scripts/write_reads_db.py $n 60 150 $d/in || exit 1
export CDM_TIMING=1
for rep in 1 2; do
  for which in new old; do
    if [ $which = old ]; then export LD_LIBRARY_PATH=$PWD/$var; else unset LD_LIBRARY_PATH; fi
    carpedeam_amd/carpedeam ancient_reads_loop $d/in $d/out_$which --ancient-damage $d/in_dhigh --num-iter-reads-only 5 --num-iterations 12 --threads 16 2> $d/log_$which
    echo "== $which rep $rep: $(grep 'Time for processing' $d/log_$which)  host merge per iteration: $(grep 'queues + extension' $d/log_$which | awk '{printf "%s ", $(NF-1)}')"
    grep "host threads" $d/log_$which | tail -1
  done
done
cmp $d/out_new $d/out_old && echo "results identical"
rm -rf $d
