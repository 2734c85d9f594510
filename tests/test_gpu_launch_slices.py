"""Kernels that take a wave per item are launched in SLICES of the items (csrc/common.h cdmSliceItems): a launch of 2^32 threads or more
is not refused by the runtime, it silently runs (blocks x threads) mod 2^32 of them (scripts/probes/big_grid.hip) - at 25 M reads
k_contig_stats (a wave per alignment record, 74 M records) had left 90 % of its statistics unwritten.  The sizes that need more than
one slice do not fit a test; CDM_LAUNCH_SLICE=<items> cuts small inputs into many launches instead, and the results must not change:
ancient_contig_merge and the Hamming rescoring against the reference's goldens, cyclecheck's selection (k_sel_copy) likewise."""
import os

import numpy as np
import pytest

from carpedeam_amd import capi, mmdb
from gpuutil import GOLD, diff_keys, seqdb_to_keyed

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["7", "300"])
def sliced(request, monkeypatch):
    monkeypatch.setenv("CDM_LAUNCH_SLICE", request.param)
    monkeypatch.setenv("CDM_CONTIG_QUEUE", "device")          # (the contig merge's sliced kernels are the device queue's; small calls take the host queue by default)
    yield int(request.param)


def test_contig_merge_in_slices(sliced, dhigh_prefix):
    from test_contig_phase import CASES, cgold
    name, last_it, step = CASES[0]
    ctx = capi.Ctx(0)
    ctx.damage_load(dhigh_prefix)
    corr = ctx.upload_keyed_seqdb(cgold(name, "ccorr", step))
    _, keys, _ = corr.meta()
    aoff, arec = capi.parse_aln_db(cgold(name, "caln", step), keys)
    assert len(arec) > 3 * sliced
    merged = ctx.contig_merge(corr, ctx.upload_alns(corr, aoff, arec))
    assert not diff_keys(seqdb_to_keyed(*merged.download()), cgold(name, "cmerge", step))


def test_hamming_rescoring_in_slices(sliced):
    ctx = capi.Ctx(0)
    g = os.path.join(GOLD, "hamming")
    db = ctx.upload_keyed_seqdb(mmdb.load_keyed(os.path.join(g, "in.keyed.gz")))
    _, keys, _ = db.meta()
    off, rec = capi.parse_pref_db(mmdb.load_keyed(os.path.join(g, "pref.keyed.gz")), keys)
    assert len(rec) > 3 * sliced
    koff, krec = ctx.rescore_hamming(db, ctx.upload_hits(db, off, rec)).download()
    krec = krec.copy()
    krec["diagonal"] = krec["diagonal"].astype(np.int16)
    assert not diff_keys({k: (v, 0) for k, v in capi.hits_to_text(koff, krec, keys).items()}, mmdb.load_keyed(os.path.join(g, "res.keyed.gz")))


def test_cyclecheck_selection_in_slices(sliced):
    from test_cyclecheck import g
    ctx = capi.Ctx(0)
    src = g("cycle", "in")
    db = ctx.upload_keyed_seqdb(src)
    cyc, rest, _ = ctx.cyclecheck(db, 65535, True)
    want = g("cycle", "chop")
    assert not diff_keys(seqdb_to_keyed(*cyc.download()), want)
    assert not diff_keys(seqdb_to_keyed(*rest.download()), {k: v for k, v in src.items() if k not in want})


@pytest.mark.parametrize("batch", ["1", "5000", "200000"])
def test_cyclecheck_in_batches_of_contigs(batch, monkeypatch):
    """cdm_cyclecheck takes the contigs in batches of fewer than 2^31 k-mer positions (32-bit ordinals, the thread limit above, 28 bytes
    of sort buffers per position): CDM_CYC_BATCH=<positions> cuts the golden DB's 151 contigs into many batches - one contig each, a few,
    a handful."""
    from test_cyclecheck import VARIANTS, g
    monkeypatch.setenv("CDM_CYC_BATCH", batch)
    ctx = capi.Ctx(0)
    src = g("cycle", "in")
    db = ctx.upload_keyed_seqdb(src)
    for name, flags in VARIANTS:
        f = dict(zip(flags[::2], flags[1::2]))
        cyc, rest, split = ctx.cyclecheck(db, int(f.get("--max-seq-len", 65535)), f.get("--chop-cycle", "0") == "1")
        want = g("cycle", name)
        assert not diff_keys(seqdb_to_keyed(*cyc.download()), want), (name, batch)
        assert not diff_keys(seqdb_to_keyed(*rest.download()), {k: v for k, v in src.items() if k not in want}), (name, batch)
