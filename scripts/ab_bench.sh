#!/bin/bash
# quick A/B of a kernel change: the kmermatcher tests, then the 50 M-read step with its per-stage kernel times
#   scripts/ab_bench.sh <tag> [pytest -k expression]
tag=${1:-ab}; sel=${2:-}
python -m pytest tests/test_gpu_kmermatch.py -x -q ${sel:+-k "$sel"} > gpurun_out/${tag}_tests.log 2>&1 || { tail -20 gpurun_out/${tag}_tests.log; exit 1; }
tail -1 gpurun_out/${tag}_tests.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
python - <<PY
import json
j = json.load(open("gpurun_out/${tag}_bench.json"))
print("ms_per_step %.1f  value %.3g" % (j["ms_per_step"], j["value"]))
print({k: round(v, 1) for k, v in j["config"]["stage_kernel_ms"].items()})
r = j["roofline"]
print("pass %.2f ms x %d  frac %.3f;" % (r["avg_launch_ms"], r["launches_per_step"], r["frac"]), {k: round(v["ms"], 1) for k, v in r["stage_level"].items()})
PY
