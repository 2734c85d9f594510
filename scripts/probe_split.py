#!/usr/bin/env python3
"""What one rank's FIRST HALF of kmermatcher costs on one GPU with W ranks (no exchange timed: the ranks run one after the other here):
split by reads - extraction + ordering of rank 0's block (cdm_kmermatch_split_begin), then sort 1 + grouping on what rank 0 receives
(cdm_kmermatch_split_finish; the other ranks' slices are made by running their split_begin too) - against the round-3 scheme, where
every rank extracts every read and keeps its k-mer range (cdm_kmermatch_part).

    python scripts/probe_split.py [reads]
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402  (device buffers for the received tuples)
from carpedeam_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ctx = capi.Ctx(0)
db = ctx.synth(n, 100, 100, 1)


def sync():
    torch.cuda.synchronize()


for W in (1, 2, 4, 8):
    sync(); t0 = time.perf_counter()
    part = ctx.kmermatch_part(db, 0, W)
    sync(); t_part = time.perf_counter() - t0
    del part
    if W == 1:
        print("W 1: first half %.1f ms" % (1e3 * t_part), flush=True)
        continue
    # rank 0's begin, timed; the others' for their slices
    sync(); t0 = time.perf_counter()
    mine = ctx.kmermatch_split_begin(db, 0, W)
    sync(); t_begin = time.perf_counter() - t0
    # all ranks' counts per fine slice -> rank 0's range = the first run of slices with 1/W of the tuples
    offs, ptrs, others = [], [], []
    off0, k0, v0, vb, _, _, _ = mine.outgoing()
    offs.append(off0); ptrs.append((k0, v0))
    for r in range(1, W):
        o = ctx.kmermatch_split_begin(db, r, W)
        off, k, v, _, _, _, _ = o.outgoing()
        offs.append(off); ptrs.append((k, v)); others.append(o)
    fine = sum(np.diff(o).astype(np.int64) for o in offs)
    cut = int(np.searchsorted(np.cumsum(fine), (fine.sum() + W - 1) // W, side="right"))
    cut = max(cut, 1)
    counts = [int(o[cut] - o[0]) for o in offs]
    slices = [(k, v, 0) for (k, v) in ptrs]
    m = sum(counts)
    rk = torch.empty(m, dtype=torch.int64, device="cuda"); rv = torch.empty(m * vb, dtype=torch.uint8, device="cuda")
    at = 0
    for (k, v, o), c in zip(slices, counts):
        if c:
            ctx.dev_copy(rk.data_ptr() + at * 8, k + o * 8, c * 8); ctx.dev_copy(rv.data_ptr() + at * vb, v + o * vb, c * vb)
        at += c
    sync()
    del others
    sync(); t0 = time.perf_counter()
    mine.split_finish(rk.data_ptr(), rv.data_ptr(), m, None, None, 0, False)
    sync(); t_finish = time.perf_counter() - t0
    print("W %d: every rank extracts all reads: %.1f ms; split by reads: begin %.1f + finish %.1f = %.1f ms (rank 0 receives %d tuples, sends %d)"
          % (W, 1e3 * t_part, 1e3 * t_begin, 1e3 * t_finish, 1e3 * (t_begin + t_finish), m, int(off0[-1])), flush=True)
    goff, _ = mine.gather(W)
    share = np.diff(goff).astype(np.float64)
    print("      group keys of rank 0's range by owner of the representative (equal id ranges): %s %%" % " ".join("%.1f" % (100 * x / max(1.0, share.sum())) for x in share), flush=True)
    del mine, rk, rv
