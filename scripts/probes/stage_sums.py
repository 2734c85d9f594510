#!/usr/bin/env python3
"""Probe: a checksum of every stage's output through the workflow loop (to find the first stage whose result depends on the memory
it was given: run under CDM_POOL=blocks, CDM_POOL=arenas, CDM_POOL_POISON=...)."""
import os
import sys
import tempfile
import zlib

import numpy as np

os.environ.setdefault("OMP_NUM_THREADS", "16")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402,F401
from carpedeam_amd import capi, synth  # noqa: E402


def csum(a):
    a = np.ascontiguousarray(a).view(np.uint8)
    pad = (-a.size) % 8
    if pad:
        a = np.concatenate([a, np.zeros(pad, np.uint8)])
    w = a.view(np.uint64)
    return "%016x" % (int(w.sum(dtype=np.uint64)) ^ (int((w * np.arange(1, w.size + 1, dtype=np.uint64)).sum(dtype=np.uint64)) << 1) & 0xFFFFFFFFFFFFFFFF)


def db_sum(db):
    lens, keys, ext = db.meta()
    offs = np.zeros(db.n, np.uint64)
    offs[1:] = np.cumsum(lens[:-1].astype(np.uint64) + 1)
    buf = np.zeros(int(lens.astype(np.uint64).sum() + db.n), np.uint8)
    db.download_into(buf, offs)
    return csum(buf) + "/" + csum(lens) + "/" + csum(ext)


def csr_sum(x):
    off, rec = x.download()
    return csum(off) + "/" + csum(rec)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
last = int(sys.argv[2]) if len(sys.argv) > 2 else 6
# which outputs are checksummed, and from which iteration on ("merged,next" from 6: nothing is downloaded before the contig merge of iteration 6 is done)
what = set((sys.argv[3] if len(sys.argv) > 3 else "hits,alns,corr,merged,next").split(","))
since = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ctx = capi.Ctx(0)
with tempfile.TemporaryDirectory() as d:
    synth.write_dhigh_profiles(os.path.join(d, "dhigh"))
    ctx.damage_load(os.path.join(d, "dhigh"))
capi.lib().cdm_pool_headroom(1.6)
db = ctx.synth(n, 60, 150, 2)
kp = capi.KmerParams.reads_default()
kc = capi.KmerParams.reads_default()
kc.kmer_size, kc.include_only_extendable = 22, 1
par = capi.AncientParams.default()
par.max_seq_len = 200000
for it in range(last + 1):
    hits = ctx.kmermatch(db, kp if it < 5 else kc)
    alns = ctx.rescore(db, hits)
    on = it >= since
    out = []
    if on and "hits" in what:
        out.append("hits " + csr_sum(hits))
    if on and "alns" in what:
        out.append("alns " + csr_sum(alns))
    del hits
    corr = ctx.correct(db, alns, par)
    if on and "corr" in what:
        out.append("corr " + db_sum(corr))
    if it < 5:
        nxt = ctx.extend(corr, alns, par)
    else:
        merged = ctx.contig_merge(corr, alns, par)
        if on and "merged" in what:
            out.append("merged " + db_sum(merged))
        cyc, nxt, _ = ctx.cyclecheck(merged, 200000, True)
        out.append("circular %d" % cyc.n)
        del merged, cyc
    if on and "next" in what:
        out.append("next " + db_sum(nxt))
    print("it %d: %s" % (it, "  ".join(out)), flush=True)
    del corr, alns
    db = nxt
