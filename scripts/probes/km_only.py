"""kmermatcher alone on synthetic reads (profiling helper): python scripts/probes/km_only.py <reads> [repeats]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from carpedeam_amd import capi

n = int(sys.argv[1])
ctx = capi.Ctx(0)
db = ctx.synth(n, 100, 100, 1)
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    t = time.time(); h = ctx.kmermatch(db); dt = time.time() - t
    print("kmermatch", n, "hits", h.count, "ms", dt * 1e3, [round(ctx.last_kernel_ms(i), 1) for i in range(8)], flush=True)
    del h
