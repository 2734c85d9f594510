#!/bin/bash
# rocprofv3 kernel statistics of the whole workflow loop (5 read + 7 contig iterations) on <reads> mixed-length reads:
# scripts/profile_loop.sh <reads> <tag>  ->  gpurun_out/<tag>_loopprof/p_kernel_stats.csv
n=${1:-1000000}; tag=${2:-lp}
R=$PWD; d=$(mktemp -d)
python scripts/write_reads_db.py $n 60 150 $d/in || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_loopprof -o p --output-format csv -- $R/carpedeam_amd/carpedeam_mi355x ancient_reads_loop $d/in $d/out --ancient-damage $d/in_dhigh --num-iter-reads-only 5 --num-iterations 12 --threads 16 2>&1 | grep -v "^[EW]2026" | tail -15
rm -f $R/gpurun_out/${tag}_loopprof/p_kernel_trace.csv; rm -rf $d
