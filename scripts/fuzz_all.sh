#!/bin/bash
# every mode of scripts/fuzz_chain.py, <cases> cases each with seed <seed>, then fuzz_modules.py: scripts/fuzz_all.sh <cases> <seed>
cases=${1:-40}; seed=${2:-1}
for m in small deep repeats contigparams longreads nrich verylong tiling tiny palrepeats letters; do
    echo "== $m"; python -u scripts/fuzz_chain.py $m $cases $seed 2>&1 | grep -a --line-buffered "FAIL\|ERROR\|reads saved\|^mode" | cut -c1-2000 || exit 1
done
echo "== modules"; python -u scripts/fuzz_modules.py $((cases / 2)) $seed 2>&1 | grep -a --line-buffered "FAIL\|ERROR\|cases" | cut -c1-2000      # (its progress lines keep a long campaign from looking hung)
