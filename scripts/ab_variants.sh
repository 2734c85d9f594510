#!/bin/bash
# bench (no tests) over library variants built by scripts/build_variant.py:  scripts/ab_variants.sh <tag> [<tag> ...]   ("base" = the regular build)
for tag in "$@"; do
  if [ "$tag" = base ]; then unset CDM_LIB; else export CDM_LIB=$PWD/carpedeam_amd/_variants/libcarpedeam_hip_$tag.so; fi
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/v_${tag}.json 2> gpurun_out/v_${tag}.err || { echo "$tag FAILED"; tail -3 gpurun_out/v_${tag}.err; continue; }
  python - <<PY
import json
j = json.load(open("gpurun_out/v_${tag}.json")); r = j["roofline"]; c = j["config"]["stage_kernel_ms"]
print("%-10s step %.1f  pass %.2f  sort1 %.1f sort2 %.1f extract %.1f | K %.1f R %.1f C %.1f E %.1f" % ("${tag}", j["ms_per_step"], r["avg_launch_ms"], c["kmer_sort1_call"], c["kmer_sort2_call"], c["kmer_extract"], *[r["stage_level"][s]["ms"] for s in ("kmermatcher", "rescorediagonal", "ancient_correction", "ancient_read_assemble")]))
PY
done
