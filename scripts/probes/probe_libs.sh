#!/bin/bash
# kmermatcher probe under several builds of the library: scripts/probes/probe_libs.sh <reads> <tag|-> ...
n=$1; shift
for tag in "$@"; do
    if [ "$tag" = "-" ]; then unset CDM_LIB; else export CDM_LIB=$PWD/carpedeam_amd/_variants/libcarpedeam_hip_$tag.so; fi
    echo "== lib $tag"
    python scripts/probe_kmer.py $n - || exit 1
done
