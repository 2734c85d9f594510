"""kmermatcher in PASSES over the k-mer space on one device (csrc/kmermatch.hip kmermatchPassesT): what an input takes whose tuples do not
fit the device at once - the tenth iteration of BASELINE config 5 at 25 M reads holds 9.2 G k-mer tuples, 295 GB.  The reference splits
the same way when memory is short (kmermatcher.cpp:634-663, merged :742-784).  CDM_KMER_PASSES=P[,B] selects P passes over B blocks of
the sequences for any DB: the hits must be the single pass's, array for array - narrow and wide group key, reads and contig parameters,
the quirks of tiny DBs against the oracle (first-group strand, the run-past-the-end scan running from one range into the next)."""
import numpy as np
import pytest

from carpedeam_amd import capi, mmdb
from gpuutil import diff_keys, run_oracle
from stageflags import K_FLAGS
from test_gpu_kmermatch import kmermatch_text, strip_ext
from test_gpu_widekey import CONTIGS, READS, databases, hits_under

pytestmark = pytest.mark.gpu

PASSES = ["2", "3,5", "4,1", "1,3", "7,2"]


@pytest.fixture(scope="module")
def ctx():
    return capi.Ctx(0)


@pytest.mark.parametrize("name", ["mixed", "deep", "contigs"])
def test_passes_equal_one_pass(ctx, monkeypatch, name):
    db = ctx.upload_seqs([s.encode() for s in databases()[name]])
    for par in (READS, CONTIGS):
        want = hits_under(ctx, db, par, {}, monkeypatch)
        assert len(want[1]) > len(want[0])
        for passes in PASSES:
            # (CDM_KMER_KEEP=0: the blocks' tuples are not kept between the passes, every range extracts the blocks again;
            #  20000 bytes: kept until the budget is passed, then dropped)
            for extra in ({}, {"CDM_FORCE_WIDE_KEY": "1"}, {"CDM_FORCE_WIDE_KEY": "1", "CDM_UNIT_CAP": "5"}, {"CDM_BUCKET_CAP": "5"}, {"CDM_KMER_KEEP": "0"},
                          {"CDM_KMER_KEEP": "20000", "CDM_FORCE_WIDE_KEY": "1"}):
                got = hits_under(ctx, db, par, dict(extra, CDM_KMER_PASSES=passes), monkeypatch)
                assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (name, par.kmer_size, passes, extra)


def test_passes_on_tiny_databases(ctx, oracle_bin, tmp_path, monkeypatch):
    """120 random databases of 2..14 short reads against the oracle, 3 passes over 2 blocks, narrow and wide key in turn."""
    rng = np.random.default_rng(303)
    letters = np.frombuffer(b"ACGT", np.uint8)
    t = lambda s: str(tmp_path / s)
    monkeypatch.setenv("CDM_KMER_PASSES", "3,2")
    for case in range(120):
        genome = rng.integers(0, 4, 120)
        seqs = []
        for _ in range(int(rng.integers(2, 15))):
            L = int(rng.integers(12, 70)); st = int(rng.integers(0, 120 - L))
            c = genome[st:st + L].copy()
            seqs.append(letters[(3 - c)[::-1] if rng.random() < 0.5 else c].tobytes().decode())
        if rng.random() < 0.3:
            seqs.append(seqs[0])
        ext = int(rng.random() < 0.4)
        if case % 2:
            monkeypatch.setenv("CDM_FORCE_WIDE_KEY", "1")
        else:
            monkeypatch.delenv("CDM_FORCE_WIDE_KEY", raising=False)
        flags = " ".join(K_FLAGS).replace("--include-only-extendable 0", "--include-only-extendable %d" % ext).split()
        mmdb.write_seqdb(t("in"), seqs)
        run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *flags, "--threads", "1")
        bad = diff_keys(kmermatch_text(ctx, mmdb.read_db(t("in")), capi.KmerParams(20, 200, 0.2, 67, 1, ext, 1, 0.0)), strip_ext(mmdb.read_db(t("pref"))))
        assert not bad, (case, ext, seqs, bad)


def test_passes_at_scale(ctx, monkeypatch):
    """1 M reads at 20x coverage: 3 passes over 4 blocks, and 6 over 2 with the wide key."""
    db = ctx.synth(1_000_000, 100, 100, 5)
    want = hits_under(ctx, db, READS, {}, monkeypatch)
    assert len(want[1]) > 3_000_000
    for env in ({"CDM_KMER_PASSES": "3,4"}, {"CDM_KMER_PASSES": "6,2", "CDM_FORCE_WIDE_KEY": "1"}, {"CDM_KMER_PASSES": "2,3", "CDM_KMER_KEEP": "0"}):
        got = hits_under(ctx, db, READS, env, monkeypatch)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), env


def test_passes_at_the_step_size(ctx, monkeypatch):
    """The 50 M-read corpus of the headline step (4.1 G k-mer slots): 2 passes over 3 blocks give the single pass's 188 M hits."""
    db = ctx.synth(50_000_000, 100, 100, 1)
    want = ctx.kmermatch(db).download()
    assert len(want[1]) == 188_090_901
    monkeypatch.setenv("CDM_KMER_PASSES", "2,3")
    got = ctx.kmermatch(db).download()
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
