#!/usr/bin/env python3
"""Probe: what W ranks of the exact multi-GPU iteration cost in kernels, measured with the ranks as threads on ONE device over the
library's RCCL transport and its in-process stand-in (csrc/dist.hip): the device runs the ranks' kernels one after the other, so the
wall time of the iteration is (about) the sum of the ranks' kernel times - divided by W, a rank's share, without the links.

    python scripts/probes/dist_ranks_serial.py <reads> <W> [<W> ...]
"""
import os
import sys
import tempfile
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from carpedeam_amd import capi, synth  # noqa: E402

n = int(sys.argv[1])
worlds = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
d = tempfile.mkdtemp()
synth.write_dhigh_profiles(os.path.join(d, "dhigh"))
for W in worlds:
    group = capi.Comm.standin_group(W)
    bar = threading.Barrier(W)
    times, paths, errs = [[] for _ in range(W)], [None] * W, []

    def body(r):
        try:
            c = capi.Ctx(0)
            c.damage_load(os.path.join(d, "dhigh"))
            comm = capi.Comm.standin(c, group, r, W)
            db = c.synth(n, 100, 100, 1)
            for it in range(3):
                c.sync(); bar.wait()
                t0 = time.perf_counter()
                h, a, co, nx = comm.reads_iteration(db)
                c.sync(); bar.wait()
                times[r].append(time.perf_counter() - t0)
                del h, a, co, nx
            paths[r] = comm.last_path()
        except BaseException as e:  # noqa: BLE001
            errs.append(e); bar.abort()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(W)]
    [t.start() for t in ts]; [t.join() for t in ts]
    if errs:
        print("W = %d: %s" % (W, errs[0]), flush=True)
        continue
    wall = min(max(times[r][it] for r in range(W)) for it in (1, 2))
    print("W = %d (kmermatcher: %s): iteration of all ranks on one device %.1f ms -> %.1f ms of kernels per rank" % (W, paths[0], 1e3 * wall, 1e3 * wall / W), flush=True)
