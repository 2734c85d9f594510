"""The CPU oracle (oracle/cdm_oracle.cpp) against the golden vectors produced by the reference's own
object code (tests/golden/make_golden.py).  This is what pins the oracle; the GPU parity tests then
compare the HIP path with the oracle."""
import gzip
import os
import subprocess

import pytest

from carpedeam_amd import mmdb
from stageflags import A_FLAGS, K_FLAGS, R_FLAGS

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def run(exe, *args, inp=None):
    r = subprocess.run([exe] + list(args), input=inp, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def _lines(path):
    with gzip.open(path, "rt") as f:
        return [l.rstrip("\n").split("\t") for l in f if l.strip()]


@pytest.mark.parametrize("name,prefix_kind", [("damage_dhigh.txt", "dhigh"), ("damage_empty.txt", "empty"), ("damage_template.txt", "tmpl")])
def test_damage_tables(oracle_bin, dhigh_prefix, tmp_path, name, prefix_kind):
    if prefix_kind == "dhigh":
        prefix = dhigh_prefix
    elif prefix_kind == "empty":
        prefix = ""
    else:
        from carpedeam_amd import synth
        prefix = str(tmp_path / "tmpl")
        body = synth.PROF_HEADER + "\n" + ("\t".join(["0.0"] * 12) + "\n") * 5
        open(prefix + "5p.prof", "w").write(body)
        open(prefix + "3p.prof", "w").write(body)
    assert run(oracle_bin, "probe", "damage", prefix) == open(os.path.join(GOLD, "functions", name)).read()


def test_seqerr(oracle_bin):
    got = run(oracle_bin, "probe", "seqerr", "0.01") + run(oracle_bin, "probe", "seqerr", "0.001")
    assert got == open(os.path.join(GOLD, "functions", "seqerr.txt")).read()


def test_most_likeli_base(oracle_bin, dhigh_prefix):
    rows = _lines(os.path.join(GOLD, "functions", "mostlikeli.tsv.gz"))
    out = run(oracle_bin, "probe", "mostlikeli", dhigh_prefix, inp="".join(r[0] + "\n" for r in rows)).split("\n")
    bad = [i for i, r in enumerate(rows) if out[i] != r[1]]
    assert not bad, "first mismatches: %s" % bad[:5]


def test_overlap_likelihood(oracle_bin, dhigh_prefix):
    rows = _lines(os.path.join(GOLD, "functions", "overlap.tsv.gz"))
    out = run(oracle_bin, "probe", "overlap", dhigh_prefix, inp="".join(r[0] + "\n" for r in rows)).split("\n")
    # bit-exact: sLenNorm and sRatio as C99 hex doubles
    bad = [i for i, r in enumerate(rows) if out[i] != r[1]]
    assert not bad, "first mismatches: %s" % [(rows[i][1], out[i]) for i in bad[:3]]


def test_evalue_table(oracle_bin):
    rows = _lines(os.path.join(GOLD, "functions", "evalue.tsv.gz"))
    by_db = {}
    for r in rows:
        db, rest = r[0].split(" ", 1)
        by_db.setdefault(db, []).append((rest, r[1]))
    for db, items in by_db.items():
        out = run(oracle_bin, "probe", "evalue", db, inp="".join(a + "\n" for a, _ in items)).split("\n")
        for (a, exp), got in zip(items, out):
            e, g = exp.split(" "), got.split(" ")
            assert g[2] == e[2], (db, a, exp, got)           # the %.3E text that reaches the alignment DB
            assert g[1] == e[1] and g[3] == e[3], (db, a)     # bit score / raw score: bit-exact
            assert abs(float.fromhex(g[0]) - float.fromhex(e[0])) <= 1e-12 * abs(float.fromhex(e[0])), (db, a)


def pref_sign_ties(got, exp):
    """Keys whose prefilter lists differ ONLY by the strand sign of one hit (same target, |score|, diagonal).

    Reference nondeterminism N1: assignGroup starts with repIsReverse=false whatever the strand of
    the very first k-mer group's representative (M/linclust/kmermatcher.cpp:453-467), so that one group can emit
    tuples whose strand bit contradicts the other tuples of the same (rep, target, diagonal); the comparator of
    the second sort ignores the strand bit (kmermatcher.h:98-114) and the sort (ips4o) is unstable, so which
    strand `writeKmerMatcherResult` sees last - and reports - depends on the run (it changes with --threads).
    At most one k-mer group per run is affected."""
    ties, bad = [], []
    for k in sorted(set(got) | set(exp)):
        if got.get(k) == exp.get(k):
            continue
        if k not in got or k not in exp or got[k][1] != exp[k][1]:
            bad.append(k)
            continue
        a, b = got[k][0].decode().split("\n"), exp[k][0].decode().split("\n")
        diff = [(x.split("\t"), y.split("\t")) for x, y in zip(a, b) if x != y]
        if len(a) == len(b) and all(x[0] == y[0] and x[2] == y[2] and int(x[1]) == -int(y[1]) for x, y in diff):
            ties.append((k, len(diff)))
        else:
            bad.append(k)
    return ties, bad


def _stage_chain(oracle_bin, dhigh_prefix, tmp_path, name, iterations):
    """Stage-isolated: every oracle stage consumes the *reference's* upstream DBs and must reproduce
    the reference's output DB key by key."""
    g = os.path.join(GOLD, name)
    t = lambda s: str(tmp_path / s)
    inp_keyed = mmdb.load_keyed(os.path.join(g, "reads.keyed.gz"))
    for it in range(iterations):
        mmdb.write_from_keyed(t("in"), inp_keyed, mmdb.DBTYPE_NUCLEOTIDES)
        gold = {s: mmdb.load_keyed(os.path.join(g, "%s_%d.keyed.gz" % (s, it))) for s in ("pref", "aln", "corr", "asm")}
        mmdb.write_from_keyed(t("pref_ref"), gold["pref"], mmdb.DBTYPE_PREFILTER_REV_RES)
        mmdb.write_from_keyed(t("aln_ref"), gold["aln"], mmdb.DBTYPE_ALIGNMENT_RES)
        mmdb.write_from_keyed(t("corr_ref"), gold["corr"], mmdb.DBTYPE_NUCLEOTIDES)
        run(oracle_bin, "kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "2")
        run(oracle_bin, "rescorediagonal", t("in"), t("in"), t("pref_ref"), t("aln"), *R_FLAGS, "--threads", "2")
        run(oracle_bin, "ancient_correction", t("in"), t("aln_ref"), t("corr"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "2")
        run(oracle_bin, "ancient_read_assemble", t("corr_ref"), t("aln_ref"), t("asm"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "2")
        for s in ("pref", "aln", "corr", "asm"):
            got, exp = mmdb.canon(mmdb.read_db(t(s))), mmdb.canon(gold[s])
            bad = [k for k in sorted(set(got) | set(exp)) if got.get(k) != exp.get(k)]
            if s == "pref":
                ties, bad = pref_sign_ties(got, exp)
                assert sum(n for _, n in ties) <= 1, ties
            assert not bad, "%s iteration %d stage %s: %d keys differ, e.g. %s" % (name, it, s, len(bad), bad[:5])
        inp_keyed = gold["asm"]


def test_chain_synth2k(oracle_bin, dhigh_prefix, tmp_path):
    _stage_chain(oracle_bin, dhigh_prefix, tmp_path, "synth2k", 2)


def test_chain_mixed3k(oracle_bin, dhigh_prefix, tmp_path):
    _stage_chain(oracle_bin, dhigh_prefix, tmp_path, "mixed3k", 3)


def test_chain_example(oracle_bin, dhigh_prefix, tmp_path):
    _stage_chain(oracle_bin, dhigh_prefix, tmp_path, "example", 1)


REF_BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "carpedeam_ref")


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref (the reference's own object code) not built")
def test_kmermatcher_small_databases_against_reference_binary(oracle_bin, tmp_path):
    """The oracle against the reference's own kmermatcher on 60 databases of 2..14 short reads, where the reference's quirks
    (first-group strand rule, per-target scan running on into the next representative's tuples and past the end of the group
    tuples into the left-over ones) decide most records.  Same generator as tests/test_gpu_kmermatch.py's fuzz test."""
    import numpy as np
    rng = np.random.default_rng(99)
    letters = np.frombuffer(b"ACGT", np.uint8)
    t = lambda s: str(tmp_path / s)
    ties_total = 0
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
        run(oracle_bin, "kmermatcher", t("in"), t("po"), *K_FLAGS, "--threads", "1")
        run(REF_BIN, "kmermatcher", t("in"), t("pr"), *K_FLAGS, "--threads", "1")
        strip = lambda db: mmdb.canon({k: (v[0], 0) for k, v in db.items()})
        ties, bad = pref_sign_ties(strip(mmdb.read_db(t("po"))), strip(mmdb.read_db(t("pr"))))
        assert not bad, (case, seqs, bad[:3])
        ties_total += sum(n for _, n in ties)
        for f in os.listdir(tmp_path):
            os.remove(t(f))
    assert ties_total <= 60      # at most the one run-dependent strand tie (N1) per database


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref (the reference's own object code) not built")
def test_chain_small_databases_against_reference_binary(oracle_bin, dhigh_prefix, tmp_path):
    """All four stages of the oracle against the reference's own object code on 40 databases of 6..60 reads (both strands, end
    damage, N, duplicates) over three iterations; stage-isolated: both programs consume the oracle's upstream DB.  Same
    generator as tests/test_gpu_chain.py's fuzz test, which pins the device path to the oracle on the same inputs."""
    import numpy as np
    rng = np.random.default_rng(2024)
    letters = np.frombuffer(b"ACGT", np.uint8)
    t = lambda s: str(tmp_path / s)
    canon = lambda p: mmdb.canon(mmdb.read_db(p))
    for case in range(40):
        genome = rng.integers(0, 4, 300)
        seqs = []
        for _ in range(int(rng.integers(6, 61))):
            L = int(rng.integers(30, 121)); st = int(rng.integers(0, 300 - L))
            c = genome[st:st + L].copy()
            if rng.random() < 0.5:
                c = (3 - c)[::-1]
            for j in range(3):
                if c[j] == 1 and rng.random() < 0.3:
                    c[j] = 3
                if c[L - 1 - j] == 2 and rng.random() < 0.3:
                    c[L - 1 - j] = 0
            sq = letters[c].tobytes().decode()
            if rng.random() < 0.05:
                k = int(rng.integers(0, L)); sq = sq[:k] + "N" + sq[k + 1:]
            seqs.append(sq)
        if rng.random() < 0.4:
            seqs.append(seqs[int(rng.integers(0, len(seqs)))])
        mmdb.write_seqdb(t("in0"), seqs)
        for it in range(3):
            i, o = t("in%d" % it), t("in%d" % (it + 1))
            ctxt = (case, it, seqs)
            for exe, sfx in ((oracle_bin, ""), (REF_BIN, "R")):
                run(exe, "kmermatcher", i, t("pref" + sfx), *K_FLAGS, "--threads", "1")
                run(exe, "rescorediagonal", i, i, t("pref"), t("aln" + sfx), *R_FLAGS, "--threads", "1")
                run(exe, "ancient_correction", i, t("aln"), t("corr" + sfx), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "1")
                run(exe, "ancient_read_assemble", t("corr"), t("aln"), o if sfx == "" else t("asmR"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "1")
            strip = lambda p: mmdb.canon({k: (v[0], 0) for k, v in mmdb.read_db(p).items()})
            ties, bad = pref_sign_ties(strip(t("pref")), strip(t("prefR")))
            assert not bad, ctxt
            assert canon(t("aln")) == canon(t("alnR")), ctxt
            assert canon(t("corr")) == canon(t("corrR")), ctxt
            assert canon(o) == canon(t("asmR")), ctxt
            for f in ("pref", "prefR", "aln", "alnR", "corr", "corrR", "asmR"):
                for ext in ("", ".index", ".dbtype"):
                    if os.path.exists(t(f) + ext):
                        os.remove(t(f) + ext)


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref (the reference's own object code) not built")
def test_kmermatcher_palindromic_repeats_against_reference_binary(oracle_bin, tmp_path):
    """Tandem repeats of reverse-palindromic units: the same canonical k-mer sits at the same stored position on both strands of a
    sequence, the per-sequence comparator ties, and libstdc++'s std::sort decides (the oracle uses the same std::sort as the
    reference; the device path emulates it, tests/test_gpu_kmermatch.py).  Ties in the reference's global ips4o sort remain
    run-dependent: only a strand sign may differ."""
    import numpy as np
    rng = np.random.default_rng(5)
    units = ["GTACGC", "GTAC", "ACGT", "GATC", "CATG", "GCGC", "AT", "TGCA", "AGCT", "GTACGCGTAC", "ACGTTGCAACGT"]
    letters = "ACGT"
    t = lambda s: str(tmp_path / s)
    for case in range(60):
        seqs = []
        for _ in range(int(rng.integers(2, 12))):
            u = units[int(rng.integers(0, len(units)))]
            L = int(rng.integers(30, 300)); o = int(rng.integers(0, len(u)))
            body = (u * (L // len(u) + 2))[o:o + L]
            left = "".join(letters[int(x)] for x in rng.integers(0, 4, int(rng.integers(0, 4))))
            right = "".join(letters[int(x)] for x in rng.integers(0, 4, int(rng.integers(0, 4))))
            seqs.append(left + body + right)
        mmdb.write_seqdb(t("in"), seqs)
        run(oracle_bin, "kmermatcher", t("in"), t("po"), *K_FLAGS, "--threads", "1")
        run(REF_BIN, "kmermatcher", t("in"), t("pr"), *K_FLAGS, "--threads", "1")
        strip = lambda db: mmdb.canon({k: (v[0], 0) for k, v in db.items()})
        ties, bad = pref_sign_ties(strip(mmdb.read_db(t("po"))), strip(mmdb.read_db(t("pr"))))
        assert not bad, (case, seqs, bad[:3])
        for f in os.listdir(tmp_path):
            os.remove(t(f))


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref (the reference's own object code) not built")
@pytest.mark.parametrize("name,it,min_cov", [("synth2k", 0, 1), ("synth2k", 1, 5), ("mixed3k", 0, 2), ("mixed3k", 2, 1), ("example", 0, 2)])
def test_unsafe_mode_against_reference_binary(oracle_bin, dhigh_prefix, tmp_path, name, it, min_cov):
    """ancient_read_assemble --unsafe 1 (consensusCaller's majority vote, nuclassembleUtil.cpp:535-702): oracle == reference."""
    from gpuutil import gold
    t = lambda s: str(tmp_path / s)
    mmdb.write_from_keyed(t("corr"), gold(name, "corr", it), mmdb.DBTYPE_NUCLEOTIDES)
    mmdb.write_from_keyed(t("aln"), gold(name, "aln", it), mmdb.DBTYPE_ALIGNMENT_RES)
    flags = " ".join(A_FLAGS).replace("--unsafe 0", "--unsafe 1").replace("--min-cov-safe 5", "--min-cov-safe %d" % min_cov).split()
    for exe, out in ((oracle_bin, "o"), (REF_BIN, "r")):
        run(exe, "ancient_read_assemble", t("corr"), t("aln"), t(out), *flags, "--ancient-damage", dhigh_prefix, "--threads", "2")
    a, b = mmdb.canon(mmdb.read_db(t("o"))), mmdb.canon(mmdb.read_db(t("r")))
    assert a == b
    safe = mmdb.canon(gold(name, "asm", it))
    assert sum(1 for k in safe if a.get(k) != safe[k]) > 20


def test_chain_letters(oracle_bin, dhigh_prefix, tmp_path):
    """reads with lower-case stretches, IUPAC codes and non-letters (NucleotideMatrix.cpp:17-61 maps them for kmermatcher and the
    score; correction/extension see nucleotideMap[c], i.e. base 0 for everything that is not ACGT; output keeps the original byte)"""
    _stage_chain(oracle_bin, dhigh_prefix, tmp_path, "letters", 3)


def test_hamming_mode_of_rescorediagonal(oracle_bin, tmp_path):
    """linclust's pre-clustering call (--rescore-mode 0 --wrapped-scoring 1) on tests/golden/hamming: 680 contigs with copies, rotations,
    reverse complements, near copies, pieces, N / lower-case / IUPAC letters and two rotated 70 000-letter contigs; golden = the
    reference's object code (make_golden.py hamming)."""
    from stageflags import HAMMING_FLAGS
    g = os.path.join(GOLD, "hamming")
    t = lambda s: str(tmp_path / s)
    mmdb.write_from_keyed(t("in"), mmdb.load_keyed(os.path.join(g, "in.keyed.gz")), mmdb.DBTYPE_NUCLEOTIDES)
    mmdb.write_from_keyed(t("pref"), mmdb.load_keyed(os.path.join(g, "pref.keyed.gz")), mmdb.DBTYPE_PREFILTER_REV_RES)
    run(oracle_bin, "rescorediagonal", t("in"), t("in"), t("pref"), t("res"), *HAMMING_FLAGS, "--threads", "4")
    exp = mmdb.load_keyed(os.path.join(g, "res.keyed.gz"))
    assert mmdb.canon(mmdb.read_db(t("res"))) == mmdb.canon(exp)
    assert mmdb.read_dbtype(t("res")) == mmdb.DBTYPE_PREFILTER_REV_RES
    recs = [l for v in exp.values() for l in v[0].decode().split("\n") if l]
    assert len(recs) > 1000 and sum(1 for l in recs if l.split("\t")[2] != "0") > 300 and sum(1 for l in recs if l.split("\t")[1].startswith("-")) > 150
