#!/bin/bash
# The round's measurements on the GPU box, in one gpurun call:  scripts/measure_round.sh <tag>
# -> gpurun_out/<tag>_*: full -m gpu suite, the default bench line, rocprofv3 kernel statistics of the same command (config 3 and
# config 2), the two PMC passes (FETCH_SIZE, WRITE_SIZE) summarised per kernel.  Copy what is to be judged into profiles/.
tag=${1:-meas}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
step() { name=$1; tmo=$2; shift 2; echo "== $name"; start=$(date +%s); timeout -k 10 $tmo "$@" > $O/${tag}_$name.log 2>&1; rc=$?; echo "== $name rc=$rc in $(( $(date +%s) - start ))s"; tail -n 2 $O/${tag}_$name.log | cut -c1-300; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit $rc; fi; }
step tests 900 python -m pytest tests -m gpu -x -q
step bench 600 python bench.py
cd /tmp && export TMPDIR=/tmp
step prof3 600 rocprofv3 --kernel-trace --stats -d $O/${tag}_prof3 -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline
step prof2 600 rocprofv3 --kernel-trace --stats -d $O/${tag}_prof2 -o p --output-format csv -- python3 $R/bench.py --config 2 --no-cpu-baseline
step pmcf 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${tag}_pmcf -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline
step pmcw 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${tag}_pmcw -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline
cd $R
python scripts/pmc_summary.py $O/${tag}_pmcf/p_counter_collection.csv $O/${tag}_pmcw/p_counter_collection.csv $O/${tag}_pmc.json "python bench.py --steps 1 --warmup 0 --no-cpu-baseline" > $O/${tag}_pmcsum.log 2>&1
rm -f $O/${tag}_pmcf/p_counter_collection.csv $O/${tag}_pmcw/p_counter_collection.csv $O/${tag}_*/p_kernel_trace.csv
ls $O/${tag}_*
