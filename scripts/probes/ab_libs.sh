#!/bin/bash
# A/B of library builds on the GPU box:  scripts/probes/ab_libs.sh <tag> <variant> [<variant> ...]   ("default" = the regular build)
# For every build: bench.py --steps 2 --warmup 1 --no-cpu-baseline (config 3) -> ms per step, the radix pass's average launch, the stage times.
tag=$1; shift
O=gpurun_out; mkdir -p $O
for v in "$@"; do
  if [ "$v" = default ]; then unset CDM_LIB; else export CDM_LIB=$PWD/carpedeam_amd/_variants/libcarpedeam_hip_$v.so; fi
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/${tag}_$v.json 2> $O/${tag}_$v.err || { echo "$v: bench failed"; tail -3 $O/${tag}_$v.err; exit 1; }
  python - "$v" $O/${tag}_$v.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
r = d["roofline"]; s = d["config"]["stage_kernel_ms"]
print("%-10s %.1f ms/step  pass %.2f ms  stages: %s  | %s" % (sys.argv[1], d["ms_per_step"], r["avg_launch_ms"], " ".join("%s %.1f" % (k, v["ms"]) for k, v in r["stage_level"].items()),
      " ".join("%s %.1f" % kv for kv in s.items())))
PY
done
