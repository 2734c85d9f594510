#!/usr/bin/env python3
"""Stress: short-lived host threads that each create a context, run kmermatcher on a small DB and end - several at a time, for <seconds>.
Every thread reserves, maps, unmaps and frees an arena of device memory at its start and end (csrc/pool.h), concurrently with the others.
CDM_POOL_DRIVER_LOCK=0 lets the threads call the driver's virtual-memory functions at the same time (A/B).

    python scripts/stress_threads.py [seconds] [threads]
"""
import faulthandler
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from carpedeam_amd import capi  # noqa: E402

faulthandler.enable()
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 5
capi.Ctx(0)                      # (the runtime's own start-up, once, before any thread)
t_end = time.time() + seconds
done = [0] * T
errs = []


def body(i):
    try:
        ctx = capi.Ctx(0)
        db = ctx.synth(20_000 + 1000 * i, 60, 150, 3 + i)
        h = ctx.kmermatch(db)
        assert h.count > db.n
        done[i] += 1
    except Exception as e:          # noqa: BLE001
        errs.append(repr(e))


rounds = 0
while time.time() < t_end and not errs:
    ts = [threading.Thread(target=body, args=(i,)) for i in range(T)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    rounds += 1
print("%d rounds of %d threads, %d kmermatcher calls, errors: %s" % (rounds, T, sum(done), errs[:3]), flush=True)
