#!/bin/bash
# The fused reads loop from a FASTQ file (parse, upload, N iterations of the four stages, download, DB write) with its laps:
#   scripts/fastq_loop.sh <reads> [iterations] [threads]
n=${1:-50000000}; it=${2:-1}; th=${3:-16}
d=$(mktemp -d)
python scripts/write_fastq.py $n 100 $d/in.fq || exit 1
ls -la $d/in.fq
for rep in 1 2; do
  time env CDM_TIMING=1 carpedeam_amd/carpedeam_mi355x ancient_reads_loop $d/in.fq $d/out --ancient-damage $d/in.fq_dhigh --num-iter-reads-only $it --threads $th
  rm -f $d/out*
done
echo "== createdb + convert2fasta on the same file"
time carpedeam_amd/carpedeam_mi355x createdb $d/in.fq $d/db --threads $th
rm -rf $d
