"""First half of the exact multi-GPU kmermatcher (extraction of a k-mer range, sort 1, grouping) for one of W ranges:
python scripts/probes/probe_parts.py <reads> <W> ...   - wall time of cdm_kmermatch_part(part 0 of W) on one device"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from carpedeam_amd import capi  # noqa: E402

n = int(sys.argv[1])
ctx = capi.Ctx(0)
db = ctx.synth(n, 100, 100, 1)
for w in map(int, sys.argv[2:]):
    best = None
    for rep in range(3):
        ctx.sync()
        t0 = time.perf_counter()
        kp = ctx.kmermatch_part(db, w // 2, w)
        ctx.sync()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        info = kp.info()
        del kp
    print("W = %d: phase A of one k-mer range %.1f ms  (extract %.1f, sort-1 call %.1f ms; info %s)" % (w, 1e3 * best, ctx.last_kernel_ms(3), ctx.last_kernel_ms(5), list(info)), flush=True)
