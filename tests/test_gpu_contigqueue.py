"""ancient_contig_merge's queue and extension loop on the device (csrc/contigqueue.hip; ancientContigsResults.cpp:25-70, 276-470)
against the same loop on the host (csrc/host/contigmerge.cpp, CDM_CONTIG_QUEUE=host) and the reference's goldens: the comparator from
the tables of the C library's own lgammaf / logf, libstdc++'s heap step for step, the rounds of the extension, the queries handed back
to the host and overlaid."""
import os

import numpy as np
import pytest

from carpedeam_amd import capi, synth
from gpuutil import diff_keys, seqdb_to_keyed
from test_contig_phase import CASES, cgold, contig_input

pytestmark = pytest.mark.gpu


def ctx_with_damage(dhigh_prefix):
    ctx = capi.Ctx(0)
    ctx.damage_load(dhigh_prefix)
    return ctx


@pytest.mark.parametrize("name,last_it,step", CASES)
def test_device_queue_equals_host_queue_on_the_goldens(dhigh_prefix, monkeypatch, name, last_it, step):
    ctx = ctx_with_damage(dhigh_prefix)
    corr = ctx.upload_keyed_seqdb(cgold(name, "ccorr", step))
    _, keys, _ = corr.meta()
    aoff, arec = capi.parse_aln_db(cgold(name, "caln", step), keys)
    want = cgold(name, "cmerge", step)
    for where in ("device", "host"):
        monkeypatch.setenv("CDM_CONTIG_QUEUE", where)
        merged = ctx.contig_merge(corr, ctx.upload_alns(corr, aoff, arec))
        assert not diff_keys(seqdb_to_keyed(*merged.download()), want), where
        ext = merged.meta()[2]
        if where == "device":
            ext_dev = ext.copy()
        else:
            assert np.array_equal(ext, ext_dev)


@pytest.mark.parametrize("every", [1, 3])
def test_queries_handed_back_to_the_host_are_overlaid(dhigh_prefix, monkeypatch, every):
    """CDM_CONTIG_HAND_BACK_EVERY=k: every k-th query takes the way of a comparison too close to call - the host code on those queries alone,
    its contigs over the device's"""
    ctx = ctx_with_damage(dhigh_prefix)
    name, step = "mixed3k", 1
    corr = ctx.upload_keyed_seqdb(cgold(name, "ccorr", step))
    _, keys, _ = corr.meta()
    aoff, arec = capi.parse_aln_db(cgold(name, "caln", step), keys)
    monkeypatch.setenv("CDM_CONTIG_QUEUE", "device")          # (a call this small takes the host queue by default)
    monkeypatch.setenv("CDM_CONTIG_HAND_BACK_EVERY", str(every))
    merged = ctx.contig_merge(corr, ctx.upload_alns(corr, aoff, arec))
    assert not diff_keys(seqdb_to_keyed(*merged.download()), cgold(name, "cmerge", step))


def loop(ctx, n, seed, iters_reads, iters_contigs, monkeypatch, where):
    """the workflow loop through the C ABI (as bench.py --config 5 runs it); -> the DB after every contig iteration"""
    monkeypatch.setenv("CDM_CONTIG_QUEUE", where)
    db = ctx.synth(n, 60, 150, seed)
    kp = capi.KmerParams.reads_default()
    kc = capi.KmerParams.reads_default()
    kc.kmer_size, kc.include_only_extendable = 22, 1
    par = capi.AncientParams.default()
    par.max_seq_len = 200000
    out = []
    for it in range(iters_reads + iters_contigs):
        alns = ctx.rescore(db, ctx.kmermatch(db, kp if it < iters_reads else kc))
        corr = ctx.correct(db, alns, par)
        if it < iters_reads:
            db = ctx.extend(corr, alns, par)
        else:
            db = ctx.contig_merge(corr, alns, par)
            lens, keys, ext = db.meta()
            out.append((db.download()[0], lens.copy(), ext.copy()))
    return out


def test_seven_contig_iterations_device_against_host(dhigh_prefix, monkeypatch):
    """200 000 mixed-length reads, 5 read + 7 contig iterations: the DB after every contig iteration, queue on the device against queue on
    the host - letters, lengths, wasExtended flags; contigs grow over many rounds (parked hits re-aligned and pushed again)"""
    ctx = ctx_with_damage(dhigh_prefix)
    dev = loop(ctx, 200_000, 2, 5, 7, monkeypatch, "device")
    host = loop(ctx, 200_000, 2, 5, 7, monkeypatch, "host")
    grew = 0
    for it, ((d, dl, de), (h, hl, he)) in enumerate(zip(dev, host)):
        assert np.array_equal(dl, hl), it
        assert np.array_equal(de, he), it
        assert d == h, it
        grew += int(de.sum())
    assert grew > 10_000 and int(dev[-1][1].max()) > 1000


def test_max_seq_len_stops_the_growth(dhigh_prefix, monkeypatch):
    """--max-seq-len small enough to bite: the loop leaves its queue non-empty (:362, :403) - same contigs either way"""
    ctx = ctx_with_damage(dhigh_prefix)
    res = {}
    for where in ("device", "host"):
        monkeypatch.setenv("CDM_CONTIG_QUEUE", where)
        db = ctx.synth(60_000, 60, 150, 5)
        kp = capi.KmerParams.reads_default()
        kc = capi.KmerParams.reads_default()
        kc.kmer_size, kc.include_only_extendable = 22, 1
        par = capi.AncientParams.default()
        par.max_seq_len = 200000
        for it in range(6):
            if it == 3:
                par.max_seq_len = 260
                longest_read = int(db.meta()[0].max())
            alns = ctx.rescore(db, ctx.kmermatch(db, kp if it < 3 else kc))
            corr = ctx.correct(db, alns, par)
            db = ctx.extend(corr, alns, par) if it < 3 else ctx.contig_merge(corr, alns, par)
        res[where] = (db.download()[0], db.meta())
    assert res["device"][0] == res["host"][0]
    assert np.array_equal(res["device"][1][2], res["host"][1][2])
    assert int(res["device"][1][0].max()) <= max(longest_read, 259)          # nothing reaches 260 letters by growing
    assert int((res["device"][1][0] > 200).sum()) > 100


def _csum(a):
    """scripts/probes/stage_sums.py csum"""
    a = np.ascontiguousarray(a).view(np.uint8)
    pad = (-a.size) % 8
    if pad:
        a = np.concatenate([a, np.zeros(pad, np.uint8)])
    w = a.view(np.uint64)
    return "%016x" % (int(w.sum(dtype=np.uint64)) ^ (int((w * np.arange(1, w.size + 1, dtype=np.uint64)).sum(dtype=np.uint64)) << 1) & 0xFFFFFFFFFFFFFFFF)


def test_small_calls_take_the_host_queue_until_the_tables_exist(dhigh_prefix, monkeypatch, capfd):
    """the default: a call of fewer than 8 M records in a process that has not filled the device queue's tables runs the queue on the host
    (a module process per iteration would pay a second for the tables every time); CDM_TIMING's laps say which way a call went"""
    monkeypatch.delenv("CDM_CONTIG_QUEUE", raising=False)
    monkeypatch.setenv("CDM_TIMING", "1")
    ctx = ctx_with_damage(dhigh_prefix)
    corr = ctx.upload_keyed_seqdb(cgold("mixed3k", "ccorr", 1))
    _, keys, _ = corr.meta()
    aoff, arec = capi.parse_aln_db(cgold("mixed3k", "caln", 1), keys)
    merged = ctx.contig_merge(corr, ctx.upload_alns(corr, aoff, arec))
    assert not diff_keys(seqdb_to_keyed(*merged.download()), cgold("mixed3k", "cmerge", 1))
    err = capfd.readouterr().err
    # (whichever test ran first in this process may have filled the tables: then the device takes every call)
    assert "queues + extension (host)" in err or "queues + extension (device)" in err


def test_config5_workflow_at_10M_reads_ends_in_the_recorded_db(dhigh_prefix):
    """BASELINE.json configs[4] (the 12-iteration loop on mixed-length reads) at 10 M reads - 25 M, its own size, wants 260 GB of the device
    and a minute, which a suite that shares a process's device memory with 200 other tests does not have; profiles/ holds that run: the
    final DB's sizes, flags and letters as recorded (tests/golden/config5/checksums.json; the same sums came out with the queue on the host
    and on the device)"""
    import json
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config5", "checksums.json")))["10000000"]
    ctx = ctx_with_damage(dhigh_prefix)
    capi.lib().cdm_pool_headroom(1.6)
    try:
        db = ctx.synth(10_000_000, 60, 150, 2)
        kp = capi.KmerParams.reads_default()
        kc = capi.KmerParams.reads_default()
        kc.kmer_size, kc.include_only_extendable = 22, 1
        par = capi.AncientParams.default()
        par.max_seq_len = 200000
        circular = 0
        for it in range(12):
            alns = ctx.rescore(db, ctx.kmermatch(db, kp if it < 5 else kc))
            corr = ctx.correct(db, alns, par)
            if it < 5:
                db = ctx.extend(corr, alns, par)
            else:
                cyc, db, _ = ctx.cyclecheck(ctx.contig_merge(corr, alns, par), 200000, True)
                circular += cyc.n
            del corr, alns
        lens, _, ext = db.meta()
        assert db.n == want["final_sequences"] and db.residues == want["final_residues"] and circular == want["circular"]
        offs = np.zeros(db.n, np.uint64)
        offs[1:] = np.cumsum(lens[:-1].astype(np.uint64) + 1)
        buf = np.zeros(int(lens.astype(np.uint64).sum() + db.n), np.uint8)
        db.download_into(buf, offs)
        assert _csum(buf) + "/" + _csum(lens) + "/" + _csum(ext) == want["next"]
    finally:
        capi.lib().cdm_pool_headroom(1.0)          # (the context's end gives the cached device memory back: cdm_ctx_destroy)
