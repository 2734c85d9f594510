"""GPU parity of ancient_read_assemble: extended sequences identical to oracle/goldens, likelihood scores bit-exact."""
import os

import numpy as np
import pytest

from carpedeam_amd import capi, mmdb
from gpuutil import DATASETS, diff_keys, gold, run_oracle, seqdb_to_keyed
from stageflags import A_FLAGS, K_FLAGS, R_FLAGS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(dhigh_prefix):
    c = capi.Ctx(0)
    c.damage_load(dhigh_prefix)
    return c


@pytest.fixture(params=["records", "records-wide-margins", "queries"])
def form(request, monkeypatch):
    """The forms of the extension kernel (extend.hip): A-D with one thread per record (k_xr_*, the default where it applies; it takes
    the likelihoods in plain double wherever their error bound decides and in software x87 elsewhere - CDM_EXTEND_MARGIN widens the
    bounds, so that many more candidates go through the exact second scoring) and one lane per query for everything (k_extend;
    CDM_EXTEND=queries)."""
    if request.param == "queries":
        monkeypatch.setenv("CDM_EXTEND", "queries")
    if request.param == "records-wide-margins":
        monkeypatch.setenv("CDM_EXTEND_MARGIN", "3e11")
    return request.param


def extend(ctx, corr_keyed, aln_keyed, want_scores=False):
    db = ctx.upload_keyed_seqdb(corr_keyed)
    _, keys, _ = db.meta()
    off, rec = capi.parse_aln_db(aln_keyed, keys)
    res = ctx.extend(db, ctx.upload_alns(db, off, rec), want_scores=want_scores)
    if want_scores:
        out, scores = res
        return seqdb_to_keyed(*out.download()), (off, rec, keys, scores)
    return seqdb_to_keyed(*res.download())


@pytest.mark.parametrize("name,its", DATASETS)
def test_extension_matches_golden_and_scores_match_oracle(ctx, oracle_bin, dhigh_prefix, tmp_path, name, its, form):
    for it in range(its):
        corr, aln = gold(name, "corr", it), gold(name, "aln", it)
        got, (off, rec, keys, scores) = extend(ctx, corr, aln, want_scores=True)
        assert not diff_keys(got, gold(name, "asm", it)), (name, it)
        # likelihood scores of the first scoring round: the oracle logs them (the module never prints them)
        t = lambda s: str(tmp_path / s)
        mmdb.write_from_keyed(t("corr"), corr, mmdb.DBTYPE_NUCLEOTIDES)
        mmdb.write_from_keyed(t("aln"), aln, mmdb.DBTYPE_ALIGNMENT_RES)
        os.environ["ORACLE_SCORES"] = t("scores.tsv")
        try:
            run_oracle(oracle_bin, "ancient_read_assemble", t("corr"), t("aln"), t("asm"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "2")
        finally:
            del os.environ["ORACLE_SCORES"]
        exp = {}
        for line in open(t("scores.tsv")):
            q, tk, s, _ = line.split("\t")
            exp[(int(q), int(tk))] = float.fromhex(s)
        got_scores = {}
        for i, k in enumerate(keys):
            for r in range(int(off[i]), int(off[i + 1])):
                if not np.isnan(scores[r]):
                    got_scores[(int(k), int(keys[rec[r]["target"]]))] = float(scores[r])
        assert set(got_scores) == set(exp), (name, it, len(got_scores), len(exp))
        assert len(exp) > 100
        bad = [k for k in exp if got_scores[k] != exp[k]]          # bit-exact (tolerance of the north star: 1e-6 relative)
        assert not bad, (name, it, bad[:3], [(got_scores[k], exp[k]) for k in bad[:3]])


def test_extension_with_N_and_max_seq_len(ctx, oracle_bin, dhigh_prefix, tmp_path, form):
    from carpedeam_amd import synth
    rng = np.random.default_rng(21)
    seqs = synth.generate_strings(1500, seed=13, mixed=(40, 150))
    seqs = ["".join("N" if rng.random() < 0.004 else c for c in s) for s in seqs]
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in"), seqs)
    run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "4")
    run_oracle(oracle_bin, "rescorediagonal", t("in"), t("in"), t("pref"), t("aln"), *R_FLAGS, "--threads", "4")
    run_oracle(oracle_bin, "ancient_correction", t("in"), t("aln"), t("corr"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "4")
    for max_len in (200000, 230):
        flags = " ".join(A_FLAGS).replace("--max-seq-len 200000", "--max-seq-len %d" % max_len).split()
        run_oracle(oracle_bin, "ancient_read_assemble", t("corr"), t("aln"), t("asm"), *flags, "--ancient-damage", dhigh_prefix, "--threads", "4")
        db = ctx.upload_keyed_seqdb(mmdb.read_db(t("corr")))
        _, keys, _ = db.meta()
        off, rec = capi.parse_aln_db(mmdb.read_db(t("aln")), keys)
        par = capi.AncientParams.default()
        par.max_seq_len = max_len
        got = seqdb_to_keyed(*ctx.extend(db, ctx.upload_alns(db, off, rec), par).download())
        exp = mmdb.read_db(t("asm"))
        assert not diff_keys(got, exp), max_len
        assert sum(v[1] for v in exp.values()) > 50


def test_extension_deep_coverage_many_rounds(ctx, oracle_bin, dhigh_prefix, tmp_path, form):
    """100x coverage, two iterations: dozens of candidates per query, many equal scores (heap tie order), several
    re-alignment rounds per query, and extended sequences as queries in the second iteration."""
    from carpedeam_amd import synth
    seqs = synth.generate_strings(5000, L=100, seed=23, coverage=100)
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in0"), seqs)
    for it in range(2):
        i = t("in%d" % it)
        run_oracle(oracle_bin, "kmermatcher", i, t("pref"), *K_FLAGS, "--threads", "4")
        run_oracle(oracle_bin, "rescorediagonal", i, i, t("pref"), t("aln"), *R_FLAGS, "--threads", "4")
        run_oracle(oracle_bin, "ancient_correction", i, t("aln"), t("corr"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "4")
        run_oracle(oracle_bin, "ancient_read_assemble", t("corr"), t("aln"), t("in%d" % (it + 1)), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "4")
        got = extend(ctx, mmdb.read_db(t("corr")), mmdb.read_db(t("aln")))
        exp = mmdb.read_db(t("in%d" % (it + 1)))
        assert not diff_keys(got, exp), it
        assert sum(v[1] for v in exp.values()) > 100


def test_both_forms_identical_at_scale(ctx, monkeypatch):
    """1 M reads at 20x coverage and a mixed-length set with pile-ups of hundreds of records: sequences, flags and every likelihood
    score of the record-parallel form equal those of the lane-per-query form."""
    for n, lo, hi, seed in ((1_000_000, 100, 100, 5), (300_000, 40, 150, 9)):
        db = ctx.synth(n, lo, hi, seed)
        alns = ctx.rescore(db, ctx.kmermatch(db))
        corr = ctx.correct(db, alns)
        out = {}
        for f, env, want in (("records", {}, False), ("margins", {"CDM_EXTEND_MARGIN": "3e11"}, False), ("exact", {}, True), ("queries", {"CDM_EXTEND": "queries"}, True)):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            res = ctx.extend(corr, alns, want_scores=want)      # (asking for the scores takes every likelihood in software x87)
            for k in env:
                monkeypatch.delenv(k)
            asm, scores = res if want else (res, None)
            seqs, keys, ext = asm.download()
            out[f] = (seqs, keys, ext, scores)
        b = out["queries"]
        assert int((~np.isnan(b[3])).sum()) > n // 2 and int(b[2].sum()) > n // 20
        for f in ("records", "margins", "exact"):
            a = out[f]
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), f
        assert np.array_equal(out["exact"][3], b[3], equal_nan=True)


def test_sparse_alignment_sets_through_both_forms(ctx, monkeypatch):
    """Alignment sets in which most queries have NO record (an uploaded set need not hold one for every sequence): hundreds of queries
    start inside one window of the record-parallel kernel - more than it takes in one go -, then sets with a single surviving query
    and with none.  Sequences, flags and scores equal those of the lane-per-query form."""
    db = ctx.synth(200_000, 100, 100, 11)
    alns = ctx.rescore(db, ctx.kmermatch(db))
    corr = ctx.correct(db, alns)
    off, rec = alns.download()
    cnt = np.diff(off.astype(np.int64))
    n = len(cnt)
    rng = np.random.default_rng(5)
    busiest = int(np.argmax(cnt))
    for keep in (rng.random(n) < 0.01, rng.random(n) < 0.2, np.arange(n) == busiest, np.zeros(n, bool)):
        kept = np.where(keep, cnt, 0)
        off2 = np.zeros(n + 1, np.uint64)
        off2[1:] = np.cumsum(kept)
        rec2 = rec[np.repeat(keep, cnt)]
        assert len(rec2) == int(off2[-1])
        sparse = ctx.upload_alns(corr, off2, rec2)
        out = {}
        for f in ("records", "queries"):
            if f == "queries":
                monkeypatch.setenv("CDM_EXTEND", "queries")
            asm, scores = ctx.extend(corr, sparse, want_scores=True)
            monkeypatch.delenv("CDM_EXTEND", raising=False)
            plain = ctx.extend(corr, sparse) if f == "records" else None      # (without the scores: the plain-double likelihoods)
            out[f] = asm.download() + (scores,) + ((plain.download(),) if plain else ())
        a, b = out["records"], out["queries"]
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3], equal_nan=True)
        assert a[4][0] == b[0] and np.array_equal(a[4][2], b[2])
        if keep.sum() > 1000:
            assert int(b[2].sum()) > 50


@pytest.mark.parametrize("name,it,min_cov", [("synth2k", 0, 1), ("synth2k", 1, 2), ("mixed3k", 0, 5), ("mixed3k", 2, 1), ("example", 0, 2)])
def test_unsafe_mode_consensus_matches_oracle(ctx, oracle_bin, dhigh_prefix, tmp_path, name, it, min_cov):
    """--unsafe 1 (consensusCaller's majority vote over the extending targets, nuclassembleUtil.cpp:570-702): the oracle's unsafe
    mode is pinned to the reference's object code in tests/test_oracle_golden.py; the result differs from the safe mode's in
    dozens to hundreds of sequences on these inputs, so the comparison below is not vacuous."""
    corr, aln = gold(name, "corr", it), gold(name, "aln", it)
    t = lambda s: str(tmp_path / s)
    mmdb.write_from_keyed(t("corr"), corr, mmdb.DBTYPE_NUCLEOTIDES)
    mmdb.write_from_keyed(t("aln"), aln, mmdb.DBTYPE_ALIGNMENT_RES)
    flags = " ".join(A_FLAGS).replace("--unsafe 0", "--unsafe 1").replace("--min-cov-safe 5", "--min-cov-safe %d" % min_cov).split()
    run_oracle(oracle_bin, "ancient_read_assemble", t("corr"), t("aln"), t("asm"), *flags, "--ancient-damage", dhigh_prefix, "--threads", "4")
    db = ctx.upload_keyed_seqdb(corr)
    _, keys, _ = db.meta()
    off, rec = capi.parse_aln_db(aln, keys)
    par = capi.AncientParams.default()
    par.unsafe, par.min_cov_safe = 1, min_cov
    got = seqdb_to_keyed(*ctx.extend(db, ctx.upload_alns(db, off, rec), par).download())
    exp = mmdb.read_db(t("asm"))
    assert not diff_keys(got, exp)
    assert len(diff_keys(exp, gold(name, "asm", it))) > 20          # (unsafe != safe on this input)


def test_unsafe_mode_deep_coverage_rounds(ctx, oracle_bin, dhigh_prefix, tmp_path):
    """100x coverage: coverage well above --min-cov-safe everywhere, several re-alignment rounds, each with its own consensus."""
    from carpedeam_amd import synth
    seqs = synth.generate_strings(3000, L=100, seed=29, coverage=100)
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in"), seqs)
    run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "4")
    run_oracle(oracle_bin, "rescorediagonal", t("in"), t("in"), t("pref"), t("aln"), *R_FLAGS, "--threads", "4")
    run_oracle(oracle_bin, "ancient_correction", t("in"), t("aln"), t("corr"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "4")
    flags = " ".join(A_FLAGS).replace("--unsafe 0", "--unsafe 1").split()
    run_oracle(oracle_bin, "ancient_read_assemble", t("corr"), t("aln"), t("asm"), *flags, "--ancient-damage", dhigh_prefix, "--threads", "4")
    db = ctx.upload_keyed_seqdb(mmdb.read_db(t("corr")))
    _, keys, _ = db.meta()
    off, rec = capi.parse_aln_db(mmdb.read_db(t("aln")), keys)
    par = capi.AncientParams.default()
    par.unsafe = 1
    got = seqdb_to_keyed(*ctx.extend(db, ctx.upload_alns(db, off, rec), par).download())
    assert not diff_keys(got, mmdb.read_db(t("asm")))


@pytest.mark.parametrize("thr", [0.3, 0.7, 0.95, 0.05])
def test_extension_with_another_likelihood_ratio_threshold(ctx, oracle_bin, dhigh_prefix, tmp_path, form, thr):
    """--likelihood-ratio-threshold != 0.5: the device decides x < log(1 / thr - 1) where the reference decides 1 / (1 + expl(x)) > thr; a
    candidate inside the few ulps where the two could differ makes the call refuse (extend.hip, ratioWindow) - none does here, and the result
    is the oracle's (which takes the reference's expression)"""
    flags = list(A_FLAGS)
    flags[flags.index("--likelihood-ratio-threshold") + 1] = repr(thr)
    t = lambda s: str(tmp_path / s)
    for name, it in (("synth2k", 0), ("mixed3k", 1)):
        corr, aln = gold(name, "corr", it), gold(name, "aln", it)
        mmdb.write_from_keyed(t("corr"), corr, mmdb.DBTYPE_NUCLEOTIDES)
        mmdb.write_from_keyed(t("aln"), aln, mmdb.DBTYPE_ALIGNMENT_RES)
        run_oracle(oracle_bin, "ancient_read_assemble", t("corr"), t("aln"), t("asm"), *flags, "--ancient-damage", dhigh_prefix, "--threads", "2")
        db = ctx.upload_keyed_seqdb(corr)
        _, keys, _ = db.meta()
        off, rec = capi.parse_aln_db(aln, keys)
        par = capi.AncientParams.default()
        par.likelihood_threshold = thr
        got = seqdb_to_keyed(*ctx.extend(db, ctx.upload_alns(db, off, rec), par).download())
        want = mmdb.read_db(t("asm"))
        assert not diff_keys(got, want), (name, it, thr)
        assert thr == 0.5 or got != gold(name, "asm", it) or thr in (0.3, 0.7)      # (a far threshold changes the result: the flag is not ignored)
