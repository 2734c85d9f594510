#!/usr/bin/env python3
"""What the stages behind kmermatcher cost ONE rank of W on one GPU: the rank's prefilter result holds the hits of the representatives it
owns (ranges of equal hit counts, as cdm_kmermatch_dist cuts them) and only the self hit of every other query.

    python scripts/probe_owned_stages.py [reads]
"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from carpedeam_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ctx = capi.Ctx(0)
with tempfile.TemporaryDirectory() as d:
    synth.write_dhigh_profiles(os.path.join(d, "dhigh"))
    ctx.damage_load(os.path.join(d, "dhigh"))
db = ctx.synth(n, 100, 100, 1)
hits = ctx.kmermatch(db)
off, rec = hits.download()
del hits
per = np.diff(off).astype(np.int64)


def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
    return r, 1e3 * (time.perf_counter() - t0)


for W in (1, 2, 4, 8):
    cum = np.cumsum(per - 1)                       # hits beyond the self hit
    lo = 0 if W == 1 else int(np.searchsorted(cum, cum[-1] * 0 // W))
    hi = n if W == 1 else int(np.searchsorted(cum, cum[-1] // W))       # rank 0's range
    own = np.zeros(n, bool); own[lo:hi] = True
    cnt = np.where(own, per, 1)
    noff = np.zeros(n + 1, np.uint64); noff[1:] = np.cumsum(cnt)
    # the self hit is the first record of a query's list in this corpus? take the record whose target is the query itself
    keep = np.zeros(len(rec), bool)
    starts = off[:-1].astype(np.int64)
    qof = np.repeat(np.arange(n), per)
    keep = own[qof] | (rec["target"] == qof)
    nrec = rec[keep]
    assert len(nrec) == int(noff[-1]), (len(nrec), int(noff[-1]))
    h = ctx.upload_hits(db, noff, nrec)
    for rep in range(2):
        a, t_r = timed(lambda: ctx.rescore(db, h))
        c, t_c = timed(lambda: ctx.correct(db, a))
        e, t_e = timed(lambda: ctx.extend(c, a))
        if rep == 1:
            print("W %d: rank 0 owns queries [%d, %d) = %.1f %% of the ids, %d hits: rescore %.1f ms, correct %.1f ms, extend %.1f ms" % (W, lo, hi, 100.0 * (hi - lo) / n, len(nrec), t_r, t_c, t_e), flush=True)
        del a, c, e
    del h
