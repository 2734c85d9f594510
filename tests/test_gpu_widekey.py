"""kmermatcher with the WIDE group key (csrc/runsort.h RunArgs, kmermatch.hip packGroupKey): the representative is not in a member's
key, sort 2 and the vote run on run records + aggregated entries only.  The reference has no such bound (16-byte KmerPosition with a
full u32 id, kmermatcher.h:49-54; `int` positions kmermatcher.cpp:803-808): BASELINE config 5 - 25 M sequences per GPU with contigs of
tens of thousands of letters - needs it.  CDM_FORCE_WIDE_KEY=1 selects the wide form for any DB: its hits must be the narrow form's
(which the other tests pin to the oracle and the goldens) array for array, through every path the wide form has - the record kernels in
one and two passes, the big-bucket grouping, units the table cannot hold and long segments through the two-stage radix sort, the
entry-buffer retry, the 128-bit entry sort word, the look-up-the-length tuple layout - and a DB that is wide by itself must equal the
oracle."""
import numpy as np
import pytest

from carpedeam_amd import capi, mmdb, synth
from gpuutil import diff_keys, run_oracle
from stageflags import K_FLAGS
from test_gpu_kmermatch import kmermatch_text, strip_ext

pytestmark = pytest.mark.gpu

WIDE = {"CDM_FORCE_WIDE_KEY": "1"}
VARIANTS = [{}, {"CDM_AGG_D": "3"}, {"CDM_UNIT_CAP": "5"}, {"CDM_UNIT_CAP": "1", "CDM_AGG_D": "1"}, {"CDM_AGG_CAP": "10"}, {"CDM_RUN_RECORDS": "twopass"},
            {"CDM_RUN_CAP": "10"}, {"CDM_BUCKET_CAP": "5"}, {"CDM_BUCKET_CAP": "3,17", "CDM_AGG_D": "40"}, {"CDM_KMER_LAYOUT": "wide"},
            {"CDM_KMER_LAYOUT": "wide", "CDM_FORCE_HUGE_LAYOUT": "1"}, {"CDM_KMER_LAYOUT": "wide", "CDM_FORCE_HUGE_LAYOUT": "1", "CDM_BUCKET_CAP": "6,100"},
            {"CDM_FORCE_WIDE_WORD": "1"}, {"CDM_FORCE_WIDE_WORD": "1", "CDM_AGG_D": "40", "CDM_UNIT_CAP": "700"}]


@pytest.fixture(scope="module")
def ctx():
    return capi.Ctx(0)


def hits_under(ctx, db, par, env, monkeypatch):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    try:
        return ctx.kmermatch(db, par).download()
    finally:
        for k in env:
            monkeypatch.delenv(k)


READS = capi.KmerParams.reads_default()
CONTIGS = capi.KmerParams(22, 200, 0.2, 67, 1, 1, 1, 0.0)        # -k 22 --include-only-extendable 1: self tuples are dropped run starts


def databases():
    rng = np.random.default_rng(23)
    letters = np.frombuffer(b"ACGT", np.uint8)
    region = letters[rng.integers(0, 4, 400)].tobytes().decode()
    out = {"mixed": synth.generate_strings(6000, seed=11, mixed=(40, 160)) + ["ACGTTGCA" * 12] * 40 + ["AC" * 50, ""]}
    deep = synth.generate_strings(2500, seed=12, mixed=(60, 140))
    deep += [region[0:100]] * 400 + [region[30:130]] * 300 + [region[150:250]] * 700 + [region[160:280]] * 90
    out["deep"] = [deep[i] for i in rng.permutation(len(deep))]
    genome = rng.integers(0, 4, 30000)
    long = []
    for _ in range(40):
        L = int(rng.integers(300, 9000)); st = int(rng.integers(0, len(genome) - L))
        c = genome[st:st + L].copy()
        long.append(letters[(3 - c)[::-1] if rng.random() < 0.5 else c].tobytes().decode())
    out["contigs"] = long + synth.generate_strings(300, seed=5, mixed=(60, 150))
    return out


@pytest.mark.parametrize("name", ["mixed", "deep", "contigs"])
def test_wide_key_equals_narrow_key(ctx, monkeypatch, name):
    db = ctx.upload_seqs([s.encode() for s in databases()[name]])
    for par in (READS, CONTIGS):
        want = hits_under(ctx, db, par, {}, monkeypatch)
        assert len(want[1]) > len(want[0])              # (more than the self hits)
        for env in VARIANTS:
            got = hits_under(ctx, db, par, dict(WIDE, **env), monkeypatch)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (name, par.kmer_size, env)


def test_wide_key_on_tiny_databases(ctx, oracle_bin, tmp_path, monkeypatch):
    """150 random databases of 2..14 short reads, where the reference's quirks decide records (first-group strand, the per-target scan
    running on into the next representative's tuples and into the left-over ones): the wide form against the oracle."""
    rng = np.random.default_rng(101)
    letters = np.frombuffer(b"ACGT", np.uint8)
    t = lambda s: str(tmp_path / s)
    monkeypatch.setenv("CDM_FORCE_WIDE_KEY", "1")
    for case in range(150):
        genome = rng.integers(0, 4, 120)
        seqs = []
        for _ in range(int(rng.integers(2, 15))):
            L = int(rng.integers(12, 70)); st = int(rng.integers(0, 120 - L))
            c = genome[st:st + L].copy()
            seqs.append(letters[(3 - c)[::-1] if rng.random() < 0.5 else c].tobytes().decode())
        if rng.random() < 0.3:
            seqs.append(seqs[0])
        ext = int(rng.random() < 0.4)
        flags = " ".join(K_FLAGS).replace("--include-only-extendable 0", "--include-only-extendable %d" % ext).split()
        mmdb.write_seqdb(t("in"), seqs)
        run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *flags, "--threads", "1")
        bad = diff_keys(kmermatch_text(ctx, mmdb.read_db(t("in")), capi.KmerParams(20, 200, 0.2, 67, 1, ext, 1, 0.0)), strip_ext(mmdb.read_db(t("pref"))))
        assert not bad, (case, ext, seqs, bad)


def test_wide_key_at_scale(ctx, monkeypatch):
    """1 M reads at 20x coverage: identical hit arrays, also with most units / segments pushed through the radix fallback."""
    db = ctx.synth(1_000_000, 100, 100, 5)
    want = hits_under(ctx, db, READS, {}, monkeypatch)
    assert len(want[1]) > 3_000_000
    for env in ({}, {"CDM_AGG_D": "100"}, {"CDM_UNIT_CAP": "700"}, {"CDM_AGG_CAP": "100000"}, {"CDM_FORCE_WIDE_WORD": "1"}):
        got = hits_under(ctx, db, READS, dict(WIDE, **env), monkeypatch)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), env


def test_database_that_needs_the_wide_key(ctx, oracle_bin, tmp_path):
    """1.1 M sequences (21 id bits, twice) and a contig of 600 k letters (a 21-bit diagonal): 64 bits and more for (rep, id, diagonal,
    strand) - refused until round 4.  Reads of both strands on the contig and on two more contigs that overlap it; against the oracle."""
    rng = np.random.default_rng(77)
    letters = np.frombuffer(b"ACGT", np.uint8)
    G = 700_000
    genome = rng.integers(0, 4, G)
    text = lambda c: letters[c].tobytes()
    rc = lambda c: (3 - c)[::-1]
    n_reads = (1 << 20) + 60_000
    starts = rng.integers(0, G - 150, n_reads)
    lens = rng.integers(60, 151, n_reads)
    flip = rng.random(n_reads) < 0.5
    g8 = letters[genome]
    r8 = letters[rc(genome)]
    seqs = [(r8[G - s - l: G - s] if f else g8[s: s + l]).tobytes() for s, l, f in zip(starts.tolist(), lens.tolist(), flip.tolist())]
    seqs[1000] = text(genome[0:600_000]); seqs[500_000] = text(rc(genome[550_000:700_000])); seqs[77] = text(genome[580_000:640_000])
    db = ctx.upload_seqs(seqs)
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in"), [s.decode() for s in seqs])
    run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "8")
    _, keys, _ = db.meta()
    off, rec = ctx.kmermatch(db).download()
    got = {k: (v, 0) for k, v in capi.hits_to_text(off, rec, keys).items()}
    want = strip_ext(mmdb.read_db(t("pref")))
    assert not diff_keys(got, want)
    assert max(abs(int(l.split(b"\t")[1])) for l in want[1000][0].split(b"\n") if l) > 5000        # the contigs' shared k-mers
