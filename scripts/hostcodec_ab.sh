#!/bin/bash
# A/B of the host text codecs on the GPU box: scripts/hostcodec_ab.sh <reads>   (needs scripts/_hc_old, scripts/_hc_new built beforehand)
n=${1:-10000000}; th=16
d=$(mktemp -d); bin=carpedeam_amd/carpedeam_mi355x
python scripts/write_fastq.py $n 100 $d/in.fq && carpedeam_amd/carpedeam_mi355x createdb $d/in.fq $d/in --shuffle 0 --threads $th 2>/dev/null && mv $d/in.fq_dhigh5p.prof $d/in_dhigh5p.prof && mv $d/in.fq_dhigh3p.prof $d/in_dhigh3p.prof || exit 1
K="--kmer-per-seq 200 --kmer-per-seq-scale 0.2 --hash-shift 67 --ignore-multi-kmer 1 --mask 0 --adjust-kmer-len 0 --cov-mode 1 -c 0 --include-only-extendable 0 -k 20"
R="--rescore-mode 3 -e 0.001 --min-seq-id 0.9 --seq-id-mode 0 --sort-results 0 -a 0 --filter-hits 0 --cov-mode 1 -c 0"
CDM_SINGLE_DATA_FILE=1 $bin kmermatcher $d/in $d/pref $K --threads $th 2>/dev/null
CDM_SINGLE_DATA_FILE=1 $bin rescorediagonal $d/in $d/in $d/pref $d/aln $R --threads $th 2>/dev/null
mkdir -p $d/o
for rep in 1 2; do echo "== mapped writes, 16 threads"; scripts/_hc_new $d/in $d/pref $d/aln $d/o 16 | grep write; echo "== pwrite, 16 threads"; CDM_NO_MMAP_WRITE=1 scripts/_hc_new $d/in $d/pref $d/aln $d/o 16 | grep write; done
echo "== new without MADV_HUGEPAGE, 16 threads"; CDM_NO_HUGEPAGE=1 scripts/_hc_new $d/in $d/pref $d/aln $d/o 16
cmp $d/o/pref $d/pref && cmp $d/o/pref.index $d/pref.index && echo "re-serialised DBs identical"
rm -rf $d
