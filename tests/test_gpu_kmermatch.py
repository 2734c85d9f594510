"""GPU parity of kmermatcher: prefilter DB text identical to the oracle / the reference's goldens."""
import numpy as np
import pytest

from carpedeam_amd import capi, mmdb, synth
from gpuutil import DATASETS, diff_keys, gold, run_oracle, stage_input
from stageflags import K_FLAGS
from test_oracle_golden import pref_sign_ties

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return capi.Ctx(0)


def kmermatch_text(ctx, seq_keyed, par=None):
    db = ctx.upload_keyed_seqdb(seq_keyed)
    _, keys, _ = db.meta()
    off, rec = ctx.kmermatch(db, par).download()
    return {k: (v, 0) for k, v in capi.hits_to_text(off, rec, keys).items()}


def strip_ext(db):
    return {k: (v[0], 0) for k, v in db.items()}


@pytest.mark.parametrize("name,its", DATASETS)
def test_kmermatch_matches_oracle_and_golden(ctx, oracle_bin, tmp_path, name, its):
    for it in range(its):
        seq_keyed = stage_input(name, it)
        got = kmermatch_text(ctx, seq_keyed)
        t = lambda s: str(tmp_path / s)
        mmdb.write_from_keyed(t("in"), seq_keyed, mmdb.DBTYPE_NUCLEOTIDES)
        run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "4")
        # bit-exact against the oracle (same deterministic tie rule) ...
        assert not diff_keys(got, strip_ext(mmdb.read_db(t("pref")))), (name, it)
        # ... and against the reference's own output up to its run-dependent strand tie (N1)
        ties, bad = pref_sign_ties(mmdb.canon(got), mmdb.canon(strip_ext(gold(name, "pref", it))))
        assert not bad and sum(n for _, n in ties) <= 1, (name, it, bad[:5], ties)


def test_kmermatch_N_repeats_and_contig_params(ctx, oracle_bin, tmp_path):
    """N letters break k-mers; low-complexity reads exercise the repeated-k-mer skip; k=22 / include-only-extendable
    are the contig-phase parameters."""
    from carpedeam_amd import synth
    rng = np.random.default_rng(3)
    seqs = synth.generate_strings(1200, seed=4, mixed=(30, 150))
    seqs = ["".join("N" if rng.random() < 0.01 else c for c in s) for s in seqs]
    seqs += ["ACGTTGCA" * 12, "AC" * 50, "A" * 80, "ACGTACGTAC" * 9 + "GGGTTTAAACCC", "ACGTTGCA" * 12, "TTGCAACG" * 11, "ACG", ""]
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in"), seqs)
    for flags, par in ((K_FLAGS, capi.KmerParams.reads_default()),
                       (" ".join(K_FLAGS).replace("-k 20", "-k 22").replace("--include-only-extendable 0", "--include-only-extendable 1").split(),
                        capi.KmerParams(22, 200, 0.2, 67, 1, 1, 1, 0.0))):
        run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *flags, "--threads", "4")
        got = kmermatch_text(ctx, mmdb.read_db(t("in")), par)
        assert not diff_keys(got, strip_ext(mmdb.read_db(t("pref"))))


def test_kmermatch_long_sequences_bottom_m_selection(ctx, oracle_bin, tmp_path):
    """Sequences of 300..3500 bp: more k-mer positions than the per-sequence budget (199 + 0.2 L), so the hash threshold /
    bottom-m selection of fillKmerPositionArray (kmermatcher.cpp:224-240,277-350) and the block-per-sequence kernel run."""
    rng = np.random.default_rng(17)
    genome = rng.integers(0, 4, 40000)
    letters = np.frombuffer(b"ACGT", np.uint8)
    seqs = []
    for _ in range(400):
        L = int(rng.integers(300, 3500))
        s = int(rng.integers(0, len(genome) - L))
        c = genome[s:s + L].copy()
        if rng.random() < 0.5:
            c = (3 - c)[::-1]
        seqs.append(letters[c].tobytes().decode())
    seqs += ["ACGTTGCAAT" * 150, "AC" * 400]          # repeats longer than the fast path's capacity
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in"), seqs)
    for k, ext in ((20, 0), (22, 1)):
        flags = " ".join(K_FLAGS).replace("-k 20", "-k %d" % k).replace("--include-only-extendable 0", "--include-only-extendable %d" % ext).split()
        run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *flags, "--threads", "4")
        got = kmermatch_text(ctx, mmdb.read_db(t("in")), capi.KmerParams(k, 200, 0.2, 67, 1, ext, 1, 0.0))
        assert not diff_keys(got, strip_ext(mmdb.read_db(t("pref")))), k


def test_kmermatch_tuple_layouts_agree(ctx, oracle_bin, tmp_path, monkeypatch):
    """The packed 12-byte tuples (short sequences) and the wide 16-byte ones are two layouts of the same algorithm."""
    from carpedeam_amd import synth
    seqs = synth.generate_strings(3000, seed=9, mixed=(20, 400)) + ["ACGTTGCA" * 12, "AC" * 50, "ACGTTGCA" * 12, ""]
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in"), seqs)
    run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "4")
    want = strip_ext(mmdb.read_db(t("pref")))
    for layout in ("packed", "wide"):
        monkeypatch.setenv("CDM_KMER_LAYOUT", layout)
        assert not diff_keys(kmermatch_text(ctx, mmdb.read_db(t("in"))), want), layout
    monkeypatch.setenv("CDM_KMER_LAYOUT", "packed")
    with pytest.raises(capi.CdmError):      # 2 * 20 + 2 * 12 bits do not fit
        kmermatch_text(ctx, {0: (b"ACGT" * 700 + b"\n", 0), 1: (b"ACGTTGCA" * 300 + b"\n", 0)})
    monkeypatch.delenv("CDM_KMER_LAYOUT")


def test_kmermatch_bucket_sort_paths_agree(ctx, oracle_bin, tmp_path, monkeypatch):
    """The in-LDS bucket finish of the two sorts, its big-bucket path (forced by a tiny chunk capacity) and the plain
    all-global radix sort give the same prefilter DB."""
    from carpedeam_amd import synth
    seqs = synth.generate_strings(6000, seed=11, mixed=(40, 160)) + ["ACGTTGCA" * 12] * 40 + ["AC" * 50, ""]
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in"), seqs)
    run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "4")
    want = strip_ext(mmdb.read_db(t("pref")))
    for env in ({}, {"CDM_BUCKET_CAP": "64"}, {"CDM_BUCKET_CAP": "5"}, {"CDM_BUCKET_CAP": "3,17"}, {"CDM_BUCKET_CAP": "1"}, {"CDM_BUCKET_CAP": "512,40"},
                {"CDM_KMER_SORT": "lsd"}, {"CDM_KMER_LAYOUT": "wide"}, {"CDM_KMER_LAYOUT": "wide", "CDM_BUCKET_CAP": "6,100"},
                # sort 2: the all-radix variant; both variants compared on the device; the unit sorter limited to segments of 5
                # tuples (the rest through the block-wide network), rocPRIM beyond 8 tuples, every segment through rocPRIM, units with
                # a sub-bucket of more than 3 tuples through the hard list
                {"CDM_KMER_SORT2": "radix"}, {"CDM_KMER_SORT2": "check"}, {"CDM_KMER_SORT2": "check", "CDM_UNIT_CAP": "5"},
                {"CDM_KMER_SORT2": "check", "CDM_UNIT_CAP": "5", "CDM_BLOCK_CAP": "8"}, {"CDM_UNIT_CAP": "1", "CDM_BLOCK_CAP": "0"},
                {"CDM_KMER_SORT2": "check", "CDM_UNIT_SUB": "3"}, {"CDM_RUN_RECORDS": "twopass"}, {"CDM_KMER_SORT2": "check", "CDM_RUN_CAP": "10"},
                {"CDM_KMER_SORT": "lsd", "CDM_KMER_LAYOUT": "wide"}, {"CDM_KMER_LAYOUT": "wide"},
                # the vote: on tuples instead of aggregated entries (aggvote.h); units of more than 3 / 40 distinct triples, segments of
                # more than 5 / 1 tuples and an entry buffer of 10 entries take the aggregation's other paths (tuple sorters + k_rle_segment,
                # the fallback to the tuple vote)
                {"CDM_KMER_VOTE": "tuples"}, {"CDM_AGG_D": "3"}, {"CDM_AGG_D": "40", "CDM_UNIT_CAP": "5"}, {"CDM_UNIT_CAP": "5", "CDM_BLOCK_CAP": "8"},
                {"CDM_UNIT_CAP": "1", "CDM_AGG_D": "1"}, {"CDM_AGG_CAP": "10"}, {"CDM_KMER_LAYOUT": "wide", "CDM_AGG_D": "7"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        assert not diff_keys(kmermatch_text(ctx, mmdb.read_db(t("in"))), want), env
        for k in env:
            monkeypatch.delenv(k)


def test_kmermatch_variants_identical_at_scale(ctx, monkeypatch):
    """1 M reads at 20x coverage (k-mer buckets of a few dozen tuples, as in the bench): the hybrid sorts, the all-global
    radix sort and the two tuple layouts return identical hit arrays."""
    db = ctx.synth(1_000_000, 100, 100, 5)
    ref = None
    for env in ({}, {"CDM_KMER_SORT": "lsd"}, {"CDM_KMER_LAYOUT": "wide"}, {"CDM_KMER_LAYOUT": "wide", "CDM_KMER_SORT": "lsd"}, {"CDM_BUCKET_CAP": "48"},
                {"CDM_KMER_SORT2": "radix"}, {"CDM_KMER_SORT2": "check"}, {"CDM_KMER_SORT2": "check", "CDM_UNIT_CAP": "700"}, {"CDM_KMER_SORT2": "check", "CDM_UNIT_SUB": "12"},
                {"CDM_RUN_RECORDS": "twopass"}, {"CDM_RUN_CAP": "1000"},
                {"CDM_KMER_VOTE": "tuples"}, {"CDM_AGG_D": "100"}, {"CDM_UNIT_CAP": "700"}, {"CDM_AGG_CAP": "100000"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        off, rec = ctx.kmermatch(db).download()
        for k in env:
            monkeypatch.delenv(k)
        if ref is None:
            ref = (off, rec)
            assert len(rec) > 3_000_000
        else:
            assert np.array_equal(off, ref[0]) and np.array_equal(rec, ref[1]), env


def test_kmermatch_high_multiplicity_buckets(ctx, oracle_bin, tmp_path, monkeypatch):
    """Hundreds of identical / overlapping reads: k-mer buckets of 257..512 tuples (the 8-words-per-lane network), buckets
    beyond 512 (gathered and sorted globally) and representatives with tens of thousands of group tuples, all at the default
    capacities."""
    from carpedeam_amd import synth
    rng = np.random.default_rng(23)
    letters = np.frombuffer(b"ACGT", np.uint8)
    region = letters[rng.integers(0, 4, 400)].tobytes().decode()
    seqs = synth.generate_strings(2500, seed=12, mixed=(60, 140))
    seqs += [region[0:100]] * 400 + [region[30:130]] * 300 + [region[150:250]] * 700 + [region[160:280]] * 90
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order]
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in"), seqs)
    run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "4")
    assert not diff_keys(kmermatch_text(ctx, mmdb.read_db(t("in"))), strip_ext(mmdb.read_db(t("pref"))))
    # sort 2 at the default capacities: segments for the wave sorter, for all three block sorters and for rocPRIM
    monkeypatch.setenv("CDM_KMER_SORT2", "check")
    assert not diff_keys(kmermatch_text(ctx, mmdb.read_db(t("in"))), strip_ext(mmdb.read_db(t("pref"))))
    monkeypatch.delenv("CDM_KMER_SORT2")


@pytest.mark.parametrize("seqs", [["ACGTTGCAAGGCTTAACGGATCCGATTACAGGCATCGA"], ["ACG", "TTGCA", "", "ACGTACGTAC"],
                                  ["ACGTTGCAAGGCTTAACGGATCCGATTACAGGCATCGA"] * 3, ["ACGTTGCAAGGCTTAACGGA", "ACGTTGCAAGGCTTAACGGA", "TCCGTTAAGCCTTGCAACGT"]])
def test_kmermatch_tiny_databases(ctx, oracle_bin, tmp_path, seqs):
    """One sequence; only sequences shorter than k (no k-mer tuple at all); identical sequences; exactly k letters."""
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in"), seqs)
    run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "2")
    assert not diff_keys(kmermatch_text(ctx, mmdb.read_db(t("in"))), strip_ext(mmdb.read_db(t("pref"))))


def test_kmermatch_fuzz_small_databases(ctx, oracle_bin, tmp_path):
    """60 random databases of 2..14 short reads cut from a 120 bp genome (both strands, duplicates, reads shorter than k):
    with so few sequences the reference's quirks decide most records - the first-group strand rule, the per-target scan
    running on into the next representative's tuples and past the end of the group tuples into the left-over ones."""
    rng = np.random.default_rng(99)
    letters = np.frombuffer(b"ACGT", np.uint8)
    t = lambda s: str(tmp_path / s)
    for case in range(60):
        genome = rng.integers(0, 4, 120)
        seqs = []
        for _ in range(int(rng.integers(2, 15))):
            L = int(rng.integers(12, 70)); st = int(rng.integers(0, 120 - L))
            c = genome[st:st + L].copy()
            if rng.random() < 0.5:
                c = (3 - c)[::-1]
            seqs.append(letters[c].tobytes().decode())
        if rng.random() < 0.3:
            seqs.append(seqs[0])
        mmdb.write_seqdb(t("in"), seqs)
        run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "1")
        bad = diff_keys(kmermatch_text(ctx, mmdb.read_db(t("in"))), strip_ext(mmdb.read_db(t("pref"))))
        assert not bad, (case, seqs, bad)


def test_kmermatch_strand_ties_follow_std_sort(ctx, oracle_bin, tmp_path):
    """Tandem repeats of reverse-palindromic units put the same canonical k-mer at the same stored position on both strands of
    one sequence.  The reference's per-sequence comparator ignores the strand, so the order of such a pair - and the strand of
    a tuple - is whatever libstdc++'s std::sort leaves; the device emulates that algorithm for these sequences."""
    rng = np.random.default_rng(5)
    units = ["GTACGC", "GTAC", "ACGT", "GATC", "CATG", "GCGC", "AT", "TGCA", "AGCT", "GTACGCGTAC", "ACGTTGCAACGT"]
    t = lambda s: str(tmp_path / s)
    first = ["TATGCGTACGCGTACGCGTACGCGTACGCGTACGCGTACGCGTACGCGTACGCATA", "GTACGCGTACGCGTACGCGTACGCGTACGCGTACGCGTACGCGTACGCGTACGCGTACGCGTACA"]
    for case in range(80):
        seqs = list(first) if case == 0 else []
        letters = "ACGT"
        for _ in range(int(rng.integers(2, 12))):
            u = units[int(rng.integers(0, len(units)))]
            L = int(rng.integers(30, 300)); o = int(rng.integers(0, len(u)))
            body = (u * (L // len(u) + 2))[o:o + L]
            left = "".join(letters[int(x)] for x in rng.integers(0, 4, int(rng.integers(0, 4))))
            right = "".join(letters[int(x)] for x in rng.integers(0, 4, int(rng.integers(0, 4))))
            seqs.append(left + body + right)
        mmdb.write_seqdb(t("in"), seqs)
        run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "1")
        bad = diff_keys(kmermatch_text(ctx, mmdb.read_db(t("in"))), strip_ext(mmdb.read_db(t("pref"))))
        assert not bad, (case, seqs, bad)


def test_kmermatch_without_ignore_multi_kmer(ctx, oracle_bin, tmp_path):
    """--ignore-multi-kmer 0: no per-sequence sort, repeated k-mers stay, the bottom-m selection walks positions in order."""
    rng = np.random.default_rng(77)
    genome = rng.integers(0, 4, 5000)
    letters = np.frombuffer(b"ACGT", np.uint8)
    seqs = []
    for _ in range(150):
        L = int(rng.integers(40, 900)); st = int(rng.integers(0, len(genome) - L))
        c = genome[st:st + L].copy()
        if rng.random() < 0.5:
            c = (3 - c)[::-1]
        seqs.append(letters[c].tobytes().decode())
    seqs += ["ACGTTGCA" * 12, "GTACGC" * 20, "AC" * 50]
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in"), seqs)
    flags = " ".join(K_FLAGS).replace("--ignore-multi-kmer 1", "--ignore-multi-kmer 0").split()
    assert flags != K_FLAGS
    run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *flags, "--threads", "1")
    got = kmermatch_text(ctx, mmdb.read_db(t("in")), capi.KmerParams(20, 200, 0.2, 67, 0, 0, 1, 0.0))
    assert not diff_keys(got, strip_ext(mmdb.read_db(t("pref"))))


def test_kmermatch_sequences_beyond_4096_positions(ctx, oracle_bin, tmp_path):
    """Sequences of 4 100 .. 12 000 letters (contigs late in the reads loop): more k-mer positions than the LDS version of the
    general extraction kernel holds, so its global-scratch variant runs; bottom-m selection picks 199 + 0.2 L of them."""
    rng = np.random.default_rng(41)
    genome = rng.integers(0, 4, 30000)
    letters = np.frombuffer(b"ACGT", np.uint8)
    seqs = []
    for _ in range(24):
        L = int(rng.integers(4100, 12000)); st = int(rng.integers(0, len(genome) - L))
        c = genome[st:st + L].copy()
        if rng.random() < 0.5:
            c = (3 - c)[::-1]
        seqs.append(letters[c].tobytes().decode())
    seqs += ["ACGTTGCAAT" * 600, synth.generate_strings(1, seed=3, mixed=(100, 101))[0]]
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in"), seqs)
    for k, ext in ((20, 0), (22, 1)):
        flags = " ".join(K_FLAGS).replace("-k 20", "-k %d" % k).replace("--include-only-extendable 0", "--include-only-extendable %d" % ext).split()
        run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *flags, "--threads", "1")
        got = kmermatch_text(ctx, mmdb.read_db(t("in")), capi.KmerParams(k, 200, 0.2, 67, 1, ext, 1, 0.0))
        assert not diff_keys(got, strip_ext(mmdb.read_db(t("pref")))), k


def test_kmermatch_long_sequences_int_position_path(ctx, oracle_bin, tmp_path):
    """Sequences of 32 765 letters and more: the reference switches to `int` positions (kmermatcher.cpp:803-808; diagonals beyond
    the short range, written truncated and probed +-65 536 by rescorediagonal).  Contigs of 33 k, 70 k (16-byte tuples with 20-bit
    fields) and 40 k letters that overlap each other by tens of thousands of letters, plus reads of both strands on them; then the
    rescored alignments."""
    rng = np.random.default_rng(31)
    letters = np.frombuffer(b"ACGT", np.uint8)
    genome = rng.integers(0, 4, 120_000)
    text = lambda c: letters[c].tobytes().decode()
    rc = lambda c: (3 - c)[::-1]
    seqs = [text(genome[0:33_000]), text(genome[20_000:90_000]), text(rc(genome[60_000:100_000])), text(genome[85_000:119_000])]
    for _ in range(300):
        L = int(rng.integers(60, 151)); st = int(rng.integers(0, 120_000 - L))
        c = genome[st:st + L]
        seqs.append(text(rc(c) if rng.random() < 0.5 else c))
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order]
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in"), seqs)
    run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "4")
    got = kmermatch_text(ctx, mmdb.read_db(t("in")))
    want = strip_ext(mmdb.read_db(t("pref")))
    assert not diff_keys(got, want)
    assert any(abs(int(l.split(b"\t")[1])) > 3000 for v in want.values() for l in v[0].split(b"\n") if l)      # contig-contig hits with thousands of shared k-mers
    # and through rescorediagonal (the truncated diagonals are probed back)
    from stageflags import R_FLAGS
    run_oracle(oracle_bin, "rescorediagonal", t("in"), t("in"), t("pref"), t("aln"), *R_FLAGS, "--threads", "4")
    db = ctx.upload_keyed_seqdb(mmdb.read_db(t("in")))
    lens, keys, _ = db.meta()
    aoff, arec = ctx.rescore(db, ctx.kmermatch(db)).download()
    assert not diff_keys({k: (v, 0) for k, v in capi.alns_to_text(aoff, arec, keys, lens, db.residues).items()}, mmdb.read_db(t("aln")))
