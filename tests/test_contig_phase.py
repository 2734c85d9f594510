"""The contig phase (data/nuclassemble.sh:148-196): kmermatcher with the contig parameters, rescorediagonal,
ancient_correction on contigs and ancient_contig_merge.  CPU: the oracle against goldens made by the reference's object code
(tests/golden/make_golden.py contigs).  -m gpu: the device path against the same goldens, through the C ABI and the host binary."""
import os
import subprocess

import numpy as np
import pytest

from carpedeam_amd import mmdb
from gpuutil import diff_keys, gold, run_oracle
from stageflags import AC_FLAGS, KC_FLAGS, R_FLAGS
from test_oracle_golden import pref_sign_ties

CASES = [("mixed3k", 2, 0), ("mixed3k", 2, 1), ("synth2k", 1, 0), ("synth2k", 1, 1), ("letters", 2, 0), ("letters", 2, 1)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cgold(name, stage, step):
    return mmdb.load_keyed(os.path.join(ROOT, "tests", "golden", name, "%s_%d.keyed.gz" % (stage, step)))


def contig_input(name, last_it, step):
    return gold(name, "asm", last_it) if step == 0 else cgold(name, "cmerge", step - 1)


@pytest.mark.parametrize("name,last_it,step", CASES)
def test_oracle_contig_phase_matches_reference_goldens(oracle_bin, dhigh_prefix, tmp_path, name, last_it, step):
    t = lambda s: str(tmp_path / s)
    mmdb.write_from_keyed(t("in"), contig_input(name, last_it, step), mmdb.DBTYPE_NUCLEOTIDES)
    run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *KC_FLAGS, "--threads", "2")
    strip = lambda db: mmdb.canon({k: (v[0], 0) for k, v in db.items()})
    ties, bad = pref_sign_ties(strip(mmdb.read_db(t("pref"))), strip(cgold(name, "cpref", step)))
    assert not bad and sum(n for _, n in ties) <= 1
    mmdb.write_from_keyed(t("pref_ref"), cgold(name, "cpref", step), mmdb.DBTYPE_PREFILTER_REV_RES)
    mmdb.write_from_keyed(t("aln_ref"), cgold(name, "caln", step), mmdb.DBTYPE_ALIGNMENT_RES)
    mmdb.write_from_keyed(t("corr_ref"), cgold(name, "ccorr", step), mmdb.DBTYPE_NUCLEOTIDES)
    dmg = ["--ancient-damage", dhigh_prefix, "--threads", "2"]
    run_oracle(oracle_bin, "rescorediagonal", t("in"), t("in"), t("pref_ref"), t("aln"), *R_FLAGS, "--threads", "2")
    assert not diff_keys(mmdb.read_db(t("aln")), cgold(name, "caln", step))
    run_oracle(oracle_bin, "ancient_correction", t("in"), t("aln_ref"), t("corr"), *AC_FLAGS, *dmg)
    assert not diff_keys(mmdb.read_db(t("corr")), cgold(name, "ccorr", step))
    run_oracle(oracle_bin, "ancient_contig_merge", t("corr_ref"), t("aln_ref"), t("merge"), *AC_FLAGS, *dmg)
    want = cgold(name, "cmerge", step)
    assert not diff_keys(mmdb.read_db(t("merge")), want)
    assert len(diff_keys(want, cgold(name, "ccorr", step))) > 50          # contigs were merged in this step


@pytest.mark.gpu
@pytest.mark.parametrize("name,last_it,step", CASES)
def test_device_contig_phase_matches_reference_goldens(dhigh_prefix, tmp_path, name, last_it, step):
    from carpedeam_amd import capi
    from gpuutil import seqdb_to_keyed
    ctx = capi.Ctx(0)
    ctx.damage_load(dhigh_prefix)
    db = ctx.upload_keyed_seqdb(contig_input(name, last_it, step))
    lens, keys, _ = db.meta()
    kp = capi.KmerParams.reads_default()
    kp.kmer_size, kp.include_only_extendable = 22, 1
    hoff, hrec = ctx.kmermatch(db, kp).download()
    strip = lambda db_: mmdb.canon({k: (v[0], 0) for k, v in db_.items()})
    ties, bad = pref_sign_ties(mmdb.canon({k: (v, 0) for k, v in capi.hits_to_text(hoff, hrec, keys).items()}), strip(cgold(name, "cpref", step)))
    assert not bad and sum(n for _, n in ties) <= 1
    # downstream stages consume the reference's own upstream DBs (stage isolation)
    off, rec = capi.parse_pref_db(cgold(name, "cpref", step), keys)
    alns = ctx.rescore(db, ctx.upload_hits(db, off, rec))
    aoff, arec = alns.download()
    assert not diff_keys({k: (v, 0) for k, v in capi.alns_to_text(aoff, arec, keys, lens, db.residues).items()}, cgold(name, "caln", step))
    aoff, arec = capi.parse_aln_db(cgold(name, "caln", step), keys)
    alns = ctx.upload_alns(db, aoff, arec)
    assert not diff_keys(seqdb_to_keyed(*ctx.correct(db, alns).download()), cgold(name, "ccorr", step))
    corr = ctx.upload_keyed_seqdb(cgold(name, "ccorr", step))
    merged = ctx.contig_merge(corr, ctx.upload_alns(corr, aoff, arec))
    assert not diff_keys(seqdb_to_keyed(*merged.download()), cgold(name, "cmerge", step))


@pytest.mark.gpu
def test_contig_merge_module_on_db_files(dhigh_prefix, tmp_path):
    from carpedeam_amd import build
    build.build()
    exe = os.path.join(ROOT, "carpedeam_amd", "carpedeam")
    t = lambda s: str(tmp_path / s)
    mmdb.write_from_keyed(t("corr"), cgold("mixed3k", "ccorr", 1), mmdb.DBTYPE_NUCLEOTIDES)
    mmdb.write_from_keyed(t("aln"), cgold("mixed3k", "caln", 1), mmdb.DBTYPE_ALIGNMENT_RES)
    r = subprocess.run([exe, "ancient_contig_merge", t("corr"), t("aln"), t("out"), *AC_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1000:]
    assert not diff_keys(mmdb.read_db(t("out")), cgold("mixed3k", "cmerge", 1))


REF = os.path.join(ROOT, "oracle", "_ref", "carpedeam_ref")
UNSAFE_CASES = [("mixed3k", 0, 1), ("mixed3k", 1, 5), ("synth2k", 0, 2), ("synth2k", 1, 1), ("letters", 0, 1), ("letters", 1, 2)]


def unsafe_flags(min_cov):
    return " ".join(AC_FLAGS).replace("--unsafe 0", "--unsafe 1").replace("--min-cov-safe 5", "--min-cov-safe %d" % min_cov).split()


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref (the reference's object code) is not built here")
@pytest.mark.parametrize("name,step,min_cov", UNSAFE_CASES)
def test_oracle_unsafe_contig_merge_against_reference_binary(oracle_bin, dhigh_prefix, tmp_path, name, step, min_cov):
    """ancient_contig_merge --unsafe 1 (consensusCaller's majority vote over the extending contigs): oracle == a live run of the
    reference's object code, and the mode is not vacuous on these inputs"""
    t = lambda s: str(tmp_path / s)
    mmdb.write_from_keyed(t("corr"), cgold(name, "ccorr", step), mmdb.DBTYPE_NUCLEOTIDES)
    mmdb.write_from_keyed(t("aln"), cgold(name, "caln", step), mmdb.DBTYPE_ALIGNMENT_RES)
    for exe, out in ((oracle_bin, "o"), (REF, "r")):
        run_oracle(exe, "ancient_contig_merge", t("corr"), t("aln"), t(out), *unsafe_flags(min_cov), "--ancient-damage", dhigh_prefix, "--threads", "2")
    a, b = mmdb.canon(mmdb.read_db(t("o"))), mmdb.canon(mmdb.read_db(t("r")))
    assert a == b
    safe = mmdb.canon(cgold(name, "cmerge", step))
    assert sum(1 for k in safe if a.get(k) != safe[k]) > 15


@pytest.mark.gpu
@pytest.mark.parametrize("name,step,min_cov", UNSAFE_CASES)
def test_contig_merge_unsafe_mode_matches_oracle(oracle_bin, dhigh_prefix, tmp_path, name, step, min_cov):
    """the module with --unsafe 1 (the consensus is worked out on the host strings in that mode) against the oracle"""
    from carpedeam_amd import build
    build.build()
    exe = os.path.join(ROOT, "carpedeam_amd", "carpedeam")
    t = lambda s: str(tmp_path / s)
    mmdb.write_from_keyed(t("corr"), cgold(name, "ccorr", step), mmdb.DBTYPE_NUCLEOTIDES)
    mmdb.write_from_keyed(t("aln"), cgold(name, "caln", step), mmdb.DBTYPE_ALIGNMENT_RES)
    run_oracle(oracle_bin, "ancient_contig_merge", t("corr"), t("aln"), t("o"), *unsafe_flags(min_cov), "--ancient-damage", dhigh_prefix, "--threads", "2")
    r = subprocess.run([exe, "ancient_contig_merge", t("corr"), t("aln"), t("g"), *unsafe_flags(min_cov), "--ancient-damage", dhigh_prefix, "--threads", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1000:]
    assert not diff_keys(mmdb.read_db(t("g")), mmdb.read_db(t("o")))
    assert diff_keys(mmdb.read_db(t("g")), cgold(name, "cmerge", step))


@pytest.mark.gpu
def test_contig_merge_counts_g_over_exotic_letter_as_g_to_a(oracle_bin, dhigh_prefix, tmp_path):
    """a database the module fuzzer found (seed 71, case 21): contigs with lower-case / IUPAC letters where a consensus G stands over a
    target letter that nucleotideMap sends to base 0 - ancientMatchCount counts that as a G->A column (nuclassembleUtil.cpp:1122-1140),
    which decides the order of the queue and with it which contig donates the extension"""
    from carpedeam_amd import build
    build.build()
    exe = os.path.join(ROOT, "carpedeam_amd", "carpedeam")
    t = lambda s: str(tmp_path / s)
    mmdb.write_from_keyed(t("in"), mmdb.load_keyed(os.path.join(ROOT, "tests", "golden", "fuzzcases", "contig_ga_letters.keyed.gz")), mmdb.DBTYPE_NUCLEOTIDES)
    dmg = ["--ancient-damage", dhigh_prefix, "--threads", "2"]
    run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *KC_FLAGS, "--threads", "1")
    run_oracle(oracle_bin, "rescorediagonal", t("in"), t("in"), t("pref"), t("aln"), *R_FLAGS, "--threads", "2")
    run_oracle(oracle_bin, "ancient_correction", t("in"), t("aln"), t("corr"), *AC_FLAGS, *dmg)
    run_oracle(oracle_bin, "ancient_contig_merge", t("corr"), t("aln"), t("o"), *AC_FLAGS, *dmg)
    r = subprocess.run([exe, "ancient_contig_merge", t("corr"), t("aln"), t("g"), *AC_FLAGS, *dmg], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1000:]
    assert not diff_keys(mmdb.read_db(t("g")), mmdb.read_db(t("o")))
    assert diff_keys(mmdb.read_db(t("g")), mmdb.read_db(t("corr")))        # (something is merged at all)


@pytest.mark.gpu
def test_fused_loop_through_both_phases(dhigh_prefix, tmp_path):
    """`ancient_reads_loop --num-iter-reads-only 3 --num-iterations 5`: three reads iterations and two contig iterations in one process,
    every intermediate in HBM, ends in the reference's own contig-phase golden (which chains its own DBs the same way)."""
    from carpedeam_amd import build
    build.build()
    exe = os.path.join(ROOT, "carpedeam_amd", "carpedeam")
    t = lambda s: str(tmp_path / s)
    mmdb.write_from_keyed(t("in"), gold("mixed3k", "reads"), mmdb.DBTYPE_NUCLEOTIDES)
    r = subprocess.run([exe, "ancient_reads_loop", t("in"), t("out"), "--ancient-damage", dhigh_prefix, "--num-iter-reads-only", "3", "--num-iterations", "5"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1000:]
    bad = diff_keys(mmdb.read_db(t("out")), cgold("mixed3k", "cmerge", 1))
    # the goldens chain the reference's own prefilter DBs (its run-dependent strand tie): what differs, if anything,
    # must be what the oracle chain - the deterministic rule - differs in as well
    if bad:
        from gpuutil import run_oracle as ro
        from conftest import ROOT as R2
        oracle = os.path.join(R2, "oracle", "_build", "cdm_oracle")
        from stageflags import A_FLAGS, K_FLAGS
        cur = t("in")
        for it in range(5):
            kf, af = (K_FLAGS, A_FLAGS) if it < 3 else (KC_FLAGS, AC_FLAGS)
            ro(oracle, "kmermatcher", cur, t("p"), *kf, "--threads", "4")
            ro(oracle, "rescorediagonal", cur, cur, t("p"), t("a"), *R_FLAGS, "--threads", "4")
            ro(oracle, "ancient_correction", cur, t("a"), t("c"), *af, "--ancient-damage", dhigh_prefix, "--threads", "4")
            ro(oracle, "ancient_read_assemble" if it < 3 else "ancient_contig_merge", t("c"), t("a"), t("n%d" % it), *af, "--ancient-damage", dhigh_prefix, "--threads", "4")
            cur = t("n%d" % it)
        assert not diff_keys(mmdb.read_db(t("out")), mmdb.read_db(cur))
