#!/bin/bash
# Runs the GPU steps given as arguments ("name::timeout_s::command") one after the other on the GPU box; a step that times
# out (or is killed) ends the call - no further GPU step is started after a hang.  Logs go to gpurun_out/<tag>_<name>.log.
# usage: scripts/gpu_steps.sh <tag> "tests::600::python -m pytest tests -m gpu -x -q" ...
tag=$1; shift
mkdir -p gpurun_out
for spec in "$@"; do
    name=${spec%%::*}; rest=${spec#*::}; tmo=${rest%%::*}; cmd=${rest#*::}
    echo "== $name (limit ${tmo}s): $cmd"
    start=$(date +%s)
    timeout -k 10 "$tmo" bash -c "$cmd" > "gpurun_out/${tag}_${name}.log" 2>&1
    rc=$?
    echo "== $name rc=$rc in $(( $(date +%s) - start ))s"
    tail -n 5 "gpurun_out/${tag}_${name}.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out: stopping"; exit $rc; fi
done
exit 0
