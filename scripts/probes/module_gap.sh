#!/bin/bash
# does a module's device work slow down when it starts right behind the previous module process?  kmermatcher + rescorediagonal at 50 M
# reads, back to back and with a pause in between (scripts/probes/module_gap.sh)
n=${1:-50000000}; th=16
export TMPDIR=/dev/shm
d=$(mktemp -d); bin=carpedeam_amd/carpedeam
python scripts/write_fastq.py $n 100 $d/in.fq && carpedeam_amd/carpedeam_mi355x createdb $d/in.fq $d/in --shuffle 0 --threads $th 2>/dev/null || exit 1
rm -f $d/in.fq
export CDM_TIMING=1
K="--kmer-per-seq 200 --kmer-per-seq-scale 0.2 --hash-shift 67 --ignore-multi-kmer 1 --mask 0 --adjust-kmer-len 0 --cov-mode 1 -c 0 --include-only-extendable 0 -k 20"
R="--rescore-mode 3 -e 0.001 --min-seq-id 0.9 --seq-id-mode 0 --sort-results 0 -a 0 --filter-hits 0 --cov-mode 1 -c 0"
t() { local s=$(date +%s%N); "$@" 2>&1 > /dev/null | grep -E "kernels|Time for|hits up|records down|waited"; echo "   wall $(( ($(date +%s%N) - s) / 1000000 )) ms"; }
for pause in 0 3 0 3; do
  echo "== pause $pause s between the modules"
  t $bin kmermatcher $d/in $d/pref $K --threads $th; sleep $pause
  t $bin rescorediagonal $d/in $d/in $d/pref $d/aln $R --threads $th; sleep $pause
done
rm -rf $d
