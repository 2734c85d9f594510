"""GPU parity of rescorediagonal: device integers + host text codec must reproduce the reference's alignment DB text."""
import numpy as np
import pytest

from carpedeam_amd import capi, mmdb
from gpuutil import DATASETS, diff_keys, gold, run_oracle, stage_input
from stageflags import K_FLAGS, R_FLAGS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return capi.Ctx(0)


def rescore_text(ctx, seq_keyed, pref_keyed):
    db = ctx.upload_keyed_seqdb(seq_keyed)
    lens, keys, _ = db.meta()
    off, rec = capi.parse_pref_db(pref_keyed, keys)
    alns = ctx.rescore(db, ctx.upload_hits(db, off, rec))
    aoff, arec = alns.download()
    return {k: (v, 0) for k, v in capi.alns_to_text(aoff, arec, keys, lens, db.residues).items()}


@pytest.mark.parametrize("name,its", DATASETS)
def test_rescore_matches_golden(ctx, name, its):
    for it in range(its):
        got = rescore_text(ctx, stage_input(name, it), gold(name, "pref", it))
        assert not diff_keys(got, gold(name, "aln", it)), (name, it)


def test_rescore_with_N_matches_oracle(ctx, oracle_bin, tmp_path):
    from carpedeam_amd import synth
    rng = np.random.default_rng(11)
    seqs = synth.generate_strings(1500, seed=9, mixed=(30, 150))
    seqs = ["".join("N" if rng.random() < 0.01 else c for c in s) for s in seqs]
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in"), seqs)
    run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "4")
    run_oracle(oracle_bin, "rescorediagonal", t("in"), t("in"), t("pref"), t("aln"), *R_FLAGS, "--threads", "4")
    got = rescore_text(ctx, mmdb.read_db(t("in")), mmdb.read_db(t("pref")))
    assert not diff_keys(got, mmdb.read_db(t("aln")))


def test_evalue_gate_is_monotone():
    """the device decides `evalue <= thr` as `score >= minScore[qLen]`: check on the host that the pass set is an up-set"""
    l = capi.lib()
    for db_res in (200000, 5000000000):
        for L in list(range(20, 400, 7)) + [1000, 5000, 20000]:
            passed = [l.cdm_evalue(float(s), float(L), db_res) <= 0.001 for s in range(0, 2 * L + 1)]
            first = passed.index(True) if True in passed else len(passed)
            assert all(passed[first:]), (db_res, L)


def test_uploads_are_range_checked(ctx):
    """records that come from files are validated on the host before any kernel indexes with them"""
    db = ctx.upload_seqs(["ACGT" * 10, "ACGTT" * 8])
    off = np.array([0, 1, 1], np.uint64)
    with pytest.raises(capi.CdmError):
        ctx.upload_hits(db, off, np.array([(7, 0, 0)], capi.HIT_DTYPE))
    with pytest.raises(capi.CdmError):
        ctx.upload_alns(db, off, np.array([(1, 10, 5, 0, 60, 0, 60, 1.0)], capi.ALN_DTYPE))      # q_end beyond the query
    with pytest.raises(capi.CdmError):
        ctx.upload_alns(db, off, np.array([(1, 10, 5, 0, 20, 0, 30, 1.0)], capi.ALN_DTYPE))      # spans differ: not ungapped
    ctx.upload_alns(db, off, np.array([(1, 10, 5, 0, 20, 5, 25, 1.0)], capi.ALN_DTYPE))


def test_identity_record_of_a_sequence_that_scores_zero_against_itself(ctx, oracle_bin, tmp_path, dhigh_prefix):
    """More than 40 % N: the self alignment scores 0 on every probed diagonal (N columns cost 3, the sum is clamped at 0), the E-value
    gate fails, but the identity record is written whatever its score (rescorediagonal.cpp:304) - with the untouched constructor
    values of the alignment: coordinates -1, identity 1.00.  The device writes that record (the oracle restates the reference and
    writes it too); the modules that would index a sequence with it refuse the set (the reference faults there)."""
    from carpedeam_amd import synth
    seqs = synth.generate_strings(300, seed=4, mixed=(40, 120))
    rng = np.random.default_rng(5)
    heavy = [3, 77, 150, 299]
    for i in heavy:                                   # half of the letters of a few reads become N
        s = list(seqs[i])
        for j in rng.choice(len(s), size=len(s) // 2 + 3, replace=False):
            s[j] = "N"
        seqs[i] = "".join(s)
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in"), seqs)
    run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "2")
    run_oracle(oracle_bin, "rescorediagonal", t("in"), t("in"), t("pref"), t("aln"), *R_FLAGS, "--threads", "2")
    want = mmdb.read_db(t("aln"))
    assert all(b"\t-1\t-1\t" in want[i][0] for i in heavy)        # (the input does reach the case)
    db = ctx.upload_keyed_seqdb(mmdb.read_db(t("in")))
    lens, keys, _ = db.meta()
    off, rec = capi.parse_pref_db(mmdb.read_db(t("pref")), keys)
    alns = ctx.rescore(db, ctx.upload_hits(db, off, rec))
    aoff, arec = alns.download()
    got = {k: (v, 0) for k, v in capi.alns_to_text(aoff, arec, keys, lens, db.residues).items()}
    assert not diff_keys(got, want)
    for stage in (lambda: ctx.correct(db, alns), lambda: ctx.extend(db, alns)):
        with pytest.raises(capi.CdmError, match="coordinates -1"):
            stage()


def test_hamming_mode_matches_golden(ctx):
    """cdm_rescore_hamming (linclust's pre-clustering: --rescore-mode 0 --wrapped-scoring 1) on tests/golden/hamming - the reference's
    object code on contigs with copies, rotations, reverse complements, N / lower-case / IUPAC letters, diagonals beyond 16 bit."""
    import os
    from gpuutil import GOLD
    g = os.path.join(GOLD, "hamming")
    seq = mmdb.load_keyed(os.path.join(g, "in.keyed.gz"))
    db = ctx.upload_keyed_seqdb(seq)
    _, keys, _ = db.meta()
    off, rec = capi.parse_pref_db(mmdb.load_keyed(os.path.join(g, "pref.keyed.gz")), keys)
    kept = ctx.rescore_hamming(db, ctx.upload_hits(db, off, rec))
    koff, krec = kept.download()
    krec = krec.copy()
    krec["diagonal"] = krec["diagonal"].astype(np.int16)          # (the text carries the diagonal as a short, QueryMatcher.h:121)
    got = {k: (v, 0) for k, v in capi.hits_to_text(koff, krec, keys).items()}
    exp = mmdb.load_keyed(os.path.join(g, "res.keyed.gz"))
    assert not diff_keys(got, exp)
    assert len(krec) > 1000
    # other criteria: every --seq-id-mode, a coverage mode that looks at both sequences, a length threshold - against the oracle below
