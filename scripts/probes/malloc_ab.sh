#!/bin/bash
# the workflow loop on <reads> reads, 16 host threads, with glibc's default malloc thresholds and with mmap/trim thresholds raised
# (large strings then come from the arenas instead of mmap/munmap, whose address-space lock serialises the threads)
n=$1
d=$(mktemp -d)
python scripts/write_reads_db.py $n 60 150 $d/in || exit 1
run() { carpedeam_amd/carpedeam ancient_reads_loop $d/in $d/out --ancient-damage $d/in_dhigh --num-iter-reads-only 5 --num-iterations 12 --threads 16 2> $d/log; echo "$1: $(grep 'Time for processing' $d/log)"; }
for rep in 1 2; do
  unset MALLOC_MMAP_THRESHOLD_ MALLOC_TRIM_THRESHOLD_ MALLOC_TOP_PAD_; run "default malloc"
  export MALLOC_MMAP_THRESHOLD_=1073741824 MALLOC_TRIM_THRESHOLD_=2147483648 MALLOC_TOP_PAD_=268435456; run "no mmap for big blocks"
done
rm -rf $d
