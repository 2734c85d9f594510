#!/usr/bin/env python3
"""Where a contig iteration's time goes at size: the workflow loop of bench.py --config 5 (5 read + 7 contig iterations through the C ABI)
with a synchronised lap per stage; CDM_TIMING=1 adds the phases of ancient_contig_merge on stderr.

    python scripts/contig_laps.py [reads] > laps.txt
"""
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "16")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402  (device init as bench.py has it)
from carpedeam_amd import capi, synth  # noqa: E402
import tempfile  # noqa: E402
import ctypes as C  # noqa: E402
import numpy as np  # noqa: E402


def main():
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
    pool_prev = np.zeros(8, np.uint64)
    t_all = time.perf_counter()
    for it in range(12):
        laps = []
        t = [time.perf_counter()]

        def lap(name):
            torch.cuda.synchronize()
            now = time.perf_counter()
            laps.append("%s %.3f" % (name, now - t[0]))
            t[0] = now
        hits = ctx.kmermatch(db, kp if it < 5 else kc); lap("kmermatcher")
        alns = ctx.rescore(db, hits); lap("rescore")
        nh, na = hits.count, alns.count
        del hits
        corr = ctx.correct(db, alns, par); lap("correct")
        if it < 5:
            nxt = ctx.extend(corr, alns, par); lap("extend")
        else:
            sys.stderr.write("-- iteration %d\n" % it); sys.stderr.flush()
            merged = ctx.contig_merge(corr, alns, par); lap("contig_merge")
            cyc, nxt, _ = ctx.cyclecheck(merged, 200000, True); lap("cyclecheck")
            laps.append("circular %d" % cyc.n)
            del merged, cyc
        del corr, alns
        st = np.zeros(8, np.uint64)
        capi.lib().cdm_pool_stats(st.ctypes.data_as(C.c_void_p))
        d = st - pool_prev
        pool_prev[:] = st
        laps.append("| pool: %d requests, %d driver calls, %.1f GB, %.2f s, %d trims; %.1f GB mapped, %.1f GB in use" % (d[0], d[2], d[3] / 1e9, d[4] / 1e9, d[5], st[6] / 1e9, st[7] / 1e9))
        print("it %2d  n %9d  residues %11d  hits %11d  alns %11d  | %s" % (it, db.n, db.residues, nh, na, "  ".join(laps)), flush=True)
        db = nxt
    print("total %.2f s" % (time.perf_counter() - t_all))


if __name__ == "__main__":
    main()
