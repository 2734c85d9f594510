"""GPU parity of ancient_correction (HIP path through the C ABI) against the oracle and the reference's goldens."""
import ctypes
import gzip
import os

import numpy as np
import pytest

from carpedeam_amd import capi, mmdb
from gpuutil import DATASETS, GOLD, diff_keys, gold, run_oracle, seqdb_to_keyed, stage_input
from stageflags import A_FLAGS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(dhigh_prefix):
    c = capi.Ctx(0)
    c.damage_load(dhigh_prefix)
    return c


def test_damage_tables_match_reference(ctx):
    got = ctx.damage_get()
    lines = open(os.path.join(GOLD, "functions", "damage_dhigh.txt")).read().strip().split("\n")
    for ln in lines:
        f = ln.split(" ")
        r, i = (1 if f[0] == "rev" else 0), int(f[1])
        exp = np.array([np.longdouble(float.fromhex(x)) if "p" not in x else _ld(x) for x in f[2:]], np.longdouble).reshape(4, 4)
        assert (got[r, i] == exp).all(), (ln, got[r, i])


def _ld(hexstr):
    """parse a C99 long double hex literal exactly"""
    neg = hexstr.startswith("-")
    h = hexstr.lstrip("-")[2:]
    mant, exp = h.split("p")
    ip, _, fp = mant.partition(".")
    v = np.longdouble(int(ip + fp, 16)) * np.longdouble(2) ** np.longdouble(int(exp) - 4 * len(fp))
    return -v if neg else v


def test_call_bases_known_answers(ctx):
    """the reference's mostLikeliBaseRead answers (3000 pile-ups incl. heavy coverage and near ties) through the
    device's software-x87 call path: bit-exact arg-max"""
    rows = [l.rstrip("\n").split("\t") for l in gzip.open(os.path.join(GOLD, "functions", "mostlikeli.tsv.gz"), "rt") if l.strip()]
    vec = np.zeros((len(rows), 48), np.uint32)
    exp = np.zeros(len(rows), np.uint8)
    for i, (a, b) in enumerate(rows):
        f = list(map(int, a.split(" ")))
        vec[i, :4] = f[:4]
        cnt, rev = np.array(f[4:48]), np.array(f[48:92])
        vec[i, 4:] = cnt | (rev << 16)
        exp[i] = int(b)
    out = np.zeros(len(rows), np.uint8)
    l = capi.lib()
    l.cdm_debug_call_bases.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_uint32, ctypes.c_void_p]
    rc = l.cdm_debug_call_bases(ctx.h, vec.ctypes.data_as(ctypes.c_void_p), len(rows), out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, l.cdm_last_error()
    bad = np.nonzero(out != exp)[0]
    assert bad.size == 0, "first mismatches at %s" % bad[:10]


@pytest.mark.parametrize("name,its", DATASETS)
def test_correction_matches_oracle_and_golden(ctx, oracle_bin, dhigh_prefix, tmp_path, name, its):
    for it in range(its):
        seq_keyed, aln_keyed = stage_input(name, it), gold(name, "aln", it)
        db = ctx.upload_keyed_seqdb(seq_keyed)
        _, keys, _ = db.meta()
        off, rec = capi.parse_aln_db(aln_keyed, keys)
        out = ctx.correct(db, ctx.upload_alns(db, off, rec))
        got = seqdb_to_keyed(*out.download())
        # oracle on the same DBs
        t = lambda s: str(tmp_path / s)
        mmdb.write_from_keyed(t("in"), seq_keyed, mmdb.DBTYPE_NUCLEOTIDES)
        mmdb.write_from_keyed(t("aln"), aln_keyed, mmdb.DBTYPE_ALIGNMENT_RES)
        run_oracle(oracle_bin, "ancient_correction", t("in"), t("aln"), t("corr"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "4")
        assert not diff_keys(got, mmdb.read_db(t("corr"))), (name, it)
        assert not diff_keys(got, gold(name, "corr", it)), (name, it)


def test_correction_with_N_and_ragged_lengths(ctx, oracle_bin, dhigh_prefix, tmp_path):
    """N letters (treated as 'A' in the tables, kept when coverage <= 1) and reads of length 30..150."""
    from carpedeam_amd import synth
    rng = np.random.default_rng(7)
    seqs = synth.generate_strings(1500, seed=5, mixed=(30, 150))
    seqs = ["".join("N" if rng.random() < 0.01 else c for c in s) for s in seqs]
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in"), seqs)
    from stageflags import K_FLAGS, R_FLAGS
    run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "4")
    run_oracle(oracle_bin, "rescorediagonal", t("in"), t("in"), t("pref"), t("aln"), *R_FLAGS, "--threads", "4")
    run_oracle(oracle_bin, "ancient_correction", t("in"), t("aln"), t("corr"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "4")
    db = ctx.upload_seqs(seqs)
    _, keys, _ = db.meta()
    off, rec = capi.parse_aln_db(mmdb.read_db(t("aln")), keys)
    got = seqdb_to_keyed(*ctx.correct(db, ctx.upload_alns(db, off, rec)).download())
    exp = mmdb.read_db(t("corr"))
    assert not diff_keys(got, exp)
    assert sum(v[0].count(b"N") for v in got.values()) > 0


def test_seqdb_roundtrip(ctx):
    seqs = ["ACGT" * 7 + "N", "A", "", "TTTTGGGGCCCCAAAANACGTACGTACGTAC", "G" * 16, "C" * 17, "T" * 33]
    db = ctx.upload_seqs(seqs, keys=[3, 5, 6, 10, 11, 12, 99], ext=[0, 1, 0, 0, 1, 0, 0])
    got, keys, ext = db.download()
    assert [g.decode() for g in got] == seqs and list(keys) == [3, 5, 6, 10, 11, 12, 99] and list(ext) == [0, 1, 0, 0, 1, 0, 0]
    # letters beyond ACGTN come back as they went in (the DB keeps the original bytes beside the mapped codes)
    odd = ["ACGTacgt", "ACGTNRYKM" * 5, "acgtn" * 9 + "*-.1", "ACGT" * 9, "x"]
    db = ctx.upload_seqs(odd)
    got, _, _ = db.download()
    assert [g.decode() for g in got] == odd


def test_correction_deep_pileups(ctx, oracle_bin, dhigh_prefix, tmp_path):
    """100x coverage: queries with more than 64 alignment records (general kernel), avCov >= 50 (inside hits are gated
    out, correction.cpp:319) and pile-ups deep enough that several candidate bases get close likelihoods."""
    from carpedeam_amd import synth
    from stageflags import K_FLAGS, R_FLAGS
    seqs = synth.generate_strings(6000, L=100, seed=11, coverage=100)
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in"), seqs)
    run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "4")
    run_oracle(oracle_bin, "rescorediagonal", t("in"), t("in"), t("pref"), t("aln"), *R_FLAGS, "--threads", "4")
    run_oracle(oracle_bin, "ancient_correction", t("in"), t("aln"), t("corr"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "4")
    aln = mmdb.read_db(t("aln"))
    assert max(v[0].count(b"\n") for v in aln.values()) > 64
    db = ctx.upload_seqs(seqs)
    _, keys, _ = db.meta()
    off, rec = capi.parse_aln_db(aln, keys)
    got = seqdb_to_keyed(*ctx.correct(db, ctx.upload_alns(db, off, rec)).download())
    assert not diff_keys(got, mmdb.read_db(t("corr")))
