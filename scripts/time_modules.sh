#!/bin/bash
# Where the wall time of the four modules goes on DB files (CDM_TIMING=1): scripts/time_modules.sh <reads> [threads]
n=${1:-1000000}; th=${2:-16}
d=$(mktemp -d); bin=carpedeam_amd/carpedeam
python scripts/write_fastq.py $n 100 $d/in.fq && carpedeam_amd/carpedeam_mi355x createdb $d/in.fq $d/in --shuffle 0 --threads $th 2>/dev/null && mv $d/in.fq_dhigh5p.prof $d/in_dhigh5p.prof && mv $d/in.fq_dhigh3p.prof $d/in_dhigh3p.prof || exit 1
export CDM_TIMING=1
K="--kmer-per-seq 200 --kmer-per-seq-scale 0.2 --hash-shift 67 --ignore-multi-kmer 1 --mask 0 --adjust-kmer-len 0 --cov-mode 1 -c 0 --include-only-extendable 0 -k 20"
R="--rescore-mode 3 -e 0.001 --min-seq-id 0.9 --seq-id-mode 0 --sort-results 0 -a 0 --filter-hits 0 --cov-mode 1 -c 0"
A="--rescore-mode 3 --max-seq-len 200000 --min-seq-id 0.9 --ext-random-align 0.85 --excess-penalty 0.0625 --min-ryseq-id-corr-reads 0.99 --likelihood-ratio-threshold 0.5 --unsafe 0 --min-cov-safe 5 --ancient-damage $d/in_dhigh"
for rep in 1 2; do
t() { echo "== $2"; local s=$(date +%s%N); "$@" > /dev/null; echo "wall $(( ($(date +%s%N) - s) / 1000000 )) ms"; }
t $bin kmermatcher $d/in $d/pref $K --threads $th
t $bin rescorediagonal $d/in $d/in $d/pref $d/aln $R --threads $th
t $bin ancient_correction $d/in $d/aln $d/corr $A --threads $th
t $bin ancient_read_assemble $d/corr $d/aln $d/asm $A --threads $th
done
rm -rf $d
