#!/usr/bin/env python3
"""Probe: the circular contigs cyclecheck reports in iteration 6 of the 25 M-read workflow loop - twice on the same input, the flagged
sequences written out for the oracle."""
import os
import sys
import tempfile

os.environ.setdefault("OMP_NUM_THREADS", "16")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402,F401
from carpedeam_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
last = int(sys.argv[2]) if len(sys.argv) > 2 else 6
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
    del hits
    corr = ctx.correct(db, alns, par)
    if it < 5:
        nxt = ctx.extend(corr, alns, par)
    else:
        merged = ctx.contig_merge(corr, alns, par)
        res = []
        for rep in range(3):
            cyc, nxt, _ = ctx.cyclecheck(merged, 200000, True)
            res.append((cyc.n, nxt.n))
            if rep == 0:
                first = cyc
        print("it %d: cyclecheck three times on the same DB: %s" % (it, res), flush=True)
        if it == last and first.n:
            seqs, keys, ext = first.download()[:3] if False else (None, None, None)
            d = first.download()
            out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r04n_cyc.fasta")
            with open(out, "w") as f:
                for i, s in enumerate(d[0][:2000]):
                    f.write(">c%d\n%s\n" % (i, bytes(s).decode()))
            print("wrote", out, len(d[0]))
        del merged, cyc
    del corr, alns
    db = nxt
