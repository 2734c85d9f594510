#!/bin/bash
# One iteration of the fused reads loop on <reads> 100-letter reads from DB files, with its laps (CDM_TIMING=1): scripts/fused_laps.sh <reads>
n=${1:-10000000}; R=$PWD; d=$(mktemp -d)
python scripts/write_fastq.py $n 100 $d/in.fq > /dev/null && $R/carpedeam_amd/carpedeam_mi355x createdb $d/in.fq $d/in --shuffle 0 --threads 16 2>/dev/null && mv $d/in.fq_dhigh5p.prof $d/in_dhigh5p.prof && mv $d/in.fq_dhigh3p.prof $d/in_dhigh3p.prof || exit 1
for rep in 1 2; do CDM_TIMING=1 $R/carpedeam_amd/carpedeam_mi355x ancient_reads_loop $d/in $d/out --ancient-damage $d/in_dhigh --num-iter-reads-only 1 --threads 16 2>&1 | tail -14; done
rm -rf $d
