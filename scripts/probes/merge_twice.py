#!/usr/bin/env python3
"""Probe: ancient_contig_merge twice on the same input at iteration 6 of the 25 M-read loop, a download in between (which changes what
memory the second call gets): which queries come out differently, and what their records look like."""
import os
import sys
import tempfile

import numpy as np

os.environ.setdefault("OMP_NUM_THREADS", "16")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402,F401
from carpedeam_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
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


def letters(d):
    lens, keys, ext = d.meta()
    offs = np.zeros(d.n, np.uint64)
    offs[1:] = np.cumsum(lens[:-1].astype(np.uint64) + 1)
    buf = np.zeros(int(lens.astype(np.uint64).sum() + d.n), np.uint8)
    d.download_into(buf, offs)
    return buf, offs, lens


for it in range(7):
    hits = ctx.kmermatch(db, kp if it < 5 else kc)
    alns = ctx.rescore(db, hits)
    del hits
    corr = ctx.correct(db, alns, par)
    if it < 5:
        nxt = ctx.extend(corr, alns, par)
    elif it == 5:
        merged = ctx.contig_merge(corr, alns, par)
        cyc, nxt, _ = ctx.cyclecheck(merged, 200000, True)
        del merged, cyc
    else:
        m1 = ctx.contig_merge(corr, alns, par)
        l1 = m1.meta()[0]
        cbuf, coffs, clens = letters(corr)
        m2 = ctx.contig_merge(corr, alns, par)
        l2 = m2.meta()[0]
        m3 = ctx.contig_merge(corr, alns, par)
        l3 = m3.meta()[0]
        diff = np.nonzero(l1 != l2)[0]
        print("first call vs second: %d queries differ in length; second vs third: %d" % (len(diff), int((l2 != l3).sum())), flush=True)
        aoff, arec = alns.download()
        b1, o1, _ = letters(m1)
        b2, o2, _ = letters(m2)
        same = 0
        for q in diff[:6]:
            print("query %d: corrected length %d, merged length %d (first) / %d (second)" % (q, clens[q], l1[q], l2[q]))
            print("  first : %s" % bytes(b1[int(o1[q]):int(o1[q]) + int(l1[q])]).decode()[:400])
            print("  second: %s" % bytes(b2[int(o2[q]):int(o2[q]) + int(l2[q])]).decode()[:400])
            for r in arec[int(aoff[q]):int(aoff[q + 1])][:12]:
                print("  record:", r, "target length", clens[int(r[0])] if hasattr(r, "__getitem__") else "")
        print("dtype of a record:", arec.dtype)
        break
    del corr, alns
    db = nxt
