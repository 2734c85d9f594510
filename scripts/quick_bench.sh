#!/bin/bash
# quick look at the 50 M-read step: bench line + rocprofv3 kernel statistics (scripts/quick_bench.sh <tag> [bench args])
tag=${1:-q}; shift
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > $O/${tag}_bench.json 2> $O/${tag}_bench.err; echo "bench rc=$?"; cut -c1-400 $O/${tag}_bench.json; tail -n 3 $O/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/${tag}_prof -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/${tag}_prof.log 2>&1; echo "prof rc=$?"
cd $R
rm -f $O/${tag}_prof/*/p_kernel_trace.csv $O/${tag}_prof/p_kernel_trace.csv
python scripts/kstats.py $(find $O/${tag}_prof -name 'p_kernel_stats.csv' | head -1) 4 2>/dev/null | head -40
