#!/bin/bash
# The workflow loop (5 read + 7 contig iterations) on <reads> mixed-length reads with its laps (CDM_TIMING=1): scripts/loop_laps.sh <reads>
n=${1:-1000000}; R=$PWD; d=$(mktemp -d)
python scripts/write_reads_db.py $n 60 150 $d/in || exit 1
CDM_TIMING=1 $R/carpedeam_amd/carpedeam_mi355x ancient_reads_loop $d/in $d/out --ancient-damage $d/in_dhigh --num-iter-reads-only 5 --num-iterations 12 --threads 16 2>&1 | tail -70
rm -rf $d
