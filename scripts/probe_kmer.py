"""kmermatcher alone under several environments: python scripts/probe_kmer.py <reads> <env spec> ...  ('-' = none; k=v;k=v)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carpedeam_amd import capi  # noqa: E402

n = int(sys.argv[1])
ctx = capi.Ctx(0)
db = ctx.synth(n, 100, 100, 1)
for spec in sys.argv[2:]:
    keys = []
    if spec != "-":
        for kv in spec.split(";"):
            k, v = kv.split("=")
            os.environ[k] = v
            keys.append(k)
    best = None
    for rep in range(3):
        hits = ctx.kmermatch(db, capi.KmerParams.reads_default())
        t = [ctx.last_kernel_ms(i) for i in (8, 3, 5, 6)]
        best = t if best is None or t[0] < best[0] else best
        del hits
    print("%-40s kmermatcher %8.2f ms  extract %7.2f  sort1 %7.2f  sort2 %7.2f" % (spec, *best), flush=True)
    for k in keys:
        del os.environ[k]
