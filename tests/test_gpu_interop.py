"""Interop in the other direction: DB files WRITTEN by the MI355X modules are read by the reference's own object code
(oracle/_ref/carpedeam_ref, skipped when that build is absent), and the reference module's result on them equals the MI355X
module's result on the same files.  Plus the front end of INTEGRATION.md (carpedeam_amd/carpedeam, csrc/host/front.c) over one
iteration of the reads loop of data/nuclassemble.sh:100-146 (the whole workflow through it: tests/test_workflow.py)."""
import os
import subprocess

import pytest

from carpedeam_amd import mmdb
from gpuutil import diff_keys, gold, stage_input
from stageflags import A_FLAGS, K_FLAGS, R_FLAGS
from test_oracle_golden import pref_sign_ties

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GPU = os.path.join(ROOT, "carpedeam_amd", "carpedeam")
REF = os.path.join(ROOT, "oracle", "_ref", "carpedeam_ref")
WRAP = GPU      # the front end: modules of the hot path -> carpedeam_mi355x, anything else -> $CARPEDEAM_REF_BIN


def run(exe, *args, env=None):
    r = subprocess.run([exe] + list(args), capture_output=True, text=True, env=env)
    assert r.returncode == 0, (exe, args[0], r.stderr[-1500:])


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref (the reference's object code) is not built here")
@pytest.mark.parametrize("name,it,split", [("synth2k", 0, False), ("mixed3k", 2, False), ("example", 0, False), ("letters", 1, False), ("mixed3k", 1, True), ("example", 0, True)])
def test_reference_modules_read_gpu_written_dbs(tmp_path, dhigh_prefix, name, it, split, monkeypatch):
    from carpedeam_amd import build
    build.build()
    if split:       # large result DBs are written as X.0 .. X.n like the reference's own (host/mmdb.cpp); forced here on small ones
        monkeypatch.setenv("CDM_SPLIT_MIN", "1")
    t = lambda s: str(tmp_path / s)
    dmg = ["--ancient-damage", dhigh_prefix, "--threads", "4"]
    mmdb.write_from_keyed(t("in"), stage_input(name, it), mmdb.DBTYPE_NUCLEOTIDES)
    # prefilter DB written by the MI355X kmermatcher -> the reference's rescorediagonal
    run(GPU, "kmermatcher", t("in"), t("pref_g"), *K_FLAGS, "--threads", "4")
    run(REF, "rescorediagonal", t("in"), t("in"), t("pref_g"), t("aln_r"), *R_FLAGS, "--threads", "4")
    run(GPU, "rescorediagonal", t("in"), t("in"), t("pref_g"), t("aln_g"), *R_FLAGS, "--threads", "4")
    assert os.path.exists(t("aln_g.0")) == split and os.path.exists(t("pref_g.3")) == split
    assert not diff_keys(mmdb.read_db(t("aln_r")), mmdb.read_db(t("aln_g")))
    # alignment DB written by the MI355X rescorediagonal -> the reference's ancient_correction
    run(REF, "ancient_correction", t("in"), t("aln_g"), t("corr_r"), *A_FLAGS, *dmg)
    run(GPU, "ancient_correction", t("in"), t("aln_g"), t("corr_g"), *A_FLAGS, *dmg)
    assert not diff_keys(mmdb.read_db(t("corr_r")), mmdb.read_db(t("corr_g")))
    # sequence DB written by the MI355X ancient_correction (+ that alignment DB) -> the reference's ancient_read_assemble
    run(REF, "ancient_read_assemble", t("corr_g"), t("aln_g"), t("asm_r"), *A_FLAGS, *dmg)
    run(GPU, "ancient_read_assemble", t("corr_g"), t("aln_g"), t("asm_g"), *A_FLAGS, *dmg)
    assert not diff_keys(mmdb.read_db(t("asm_r")), mmdb.read_db(t("asm_g")))
    # the extended sequence DB (wasExtended flags in the index) written by the MI355X module -> the reference's kmermatcher
    # (a LIVE run of the reference: ips4o seeds its sampling from std::random_device, so the order of equal tuples - and with it the
    # strand sign of the hits of the ONE k-mer group N1 is about, all in the list of that group's representative - changes from run
    # to run even with one thread)
    run(REF, "kmermatcher", t("asm_g"), t("pref2_r"), *K_FLAGS, "--threads", "1")
    run(GPU, "kmermatcher", t("asm_g"), t("pref2_g"), *K_FLAGS, "--threads", "4")
    ties, bad = pref_sign_ties(mmdb.canon(mmdb.read_db(t("pref2_g"))), mmdb.canon(mmdb.read_db(t("pref2_r"))))
    assert not bad and len(ties) <= 1                       # sign-only differences, confined to one query's list


def test_dispatcher_script_runs_one_reads_loop_iteration(tmp_path, dhigh_prefix):
    """$MMSEQS = carpedeam_amd/carpedeam: the loop body of data/nuclassemble.sh:100-146, module by module on DB files."""
    from carpedeam_amd import build
    build.build()
    t = lambda s: str(tmp_path / s)
    log = t("dispatch.log")
    env = dict(os.environ, CARPEDEAM_REF_BIN=REF if os.path.exists(REF) else "/bin/false", CARPEDEAM_DISPATCH_LOG=log)
    env.pop("CARPEDEAM_ALLOW_REF_FALLBACK", None)
    dmg = ["--ancient-damage", dhigh_prefix, "--threads", "4"]
    mmdb.write_from_keyed(t("in"), gold("synth2k", "reads"), mmdb.DBTYPE_NUCLEOTIDES)
    run(WRAP, "kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "4", env=env)
    run(WRAP, "rescorediagonal", t("in"), t("in"), t("pref"), t("aln"), *R_FLAGS, "--threads", "4", env=env)
    run(WRAP, "ancient_correction", t("in"), t("aln"), t("corr"), *A_FLAGS, *dmg, env=env)
    run(WRAP, "ancient_read_assemble", t("corr"), t("aln"), t("asm"), *A_FLAGS, *dmg, env=env)
    ties, bad = pref_sign_ties(mmdb.canon(mmdb.read_db(t("pref"))), mmdb.canon(gold("synth2k", "pref", 0)))
    assert not bad and sum(n for _, n in ties) <= 1
    if not ties:
        assert not diff_keys(mmdb.read_db(t("asm")), gold("synth2k", "asm", 0))
    # every one of the four calls ran on the device binary: no hand-over, no refusal, no owned module on the reference
    assert [l.split() for l in open(log)] == [["gpu", m] for m in ("kmermatcher", "rescorediagonal", "ancient_correction", "ancient_read_assemble")]
    # a module that is not one of the four goes to the reference binary
    r = subprocess.run([WRAP, "not_a_hot_path_module"], capture_output=True, text=True, env=env)
    assert r.returncode != 0
    assert [l.split() for l in open(log)][4:] == [["ref", "not_a_hot_path_module"]]


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref (the reference's object code) is not built here")
def test_whole_workflow_against_reference_object_code():
    """scripts/loop_vs_ref.py on 20 000 mixed-length reads: 5 read + 7 contig iterations with cyclecheck through `ancient_reads_loop`
    and through the reference's own modules chained as data/nuclassemble.sh chains them.  The reference's sort breaks strand ties
    differently from run to run (about 40 of a million result sequences over the twelve iterations, 0 in the 100 k-read
    runs kept under profiles/), hence the small allowance; anything systematic shows up as hundreds."""
    import re
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "loop_vs_ref.py"), "20000", "4", "twice"], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-800:])
    m = re.search(r"result (\d+) sequences, (\d+) residues, \d+ circular set aside; (\d+) sequences differ", r.stdout)
    m2 = re.search(r"the reference against its own second run: (\d+) sequences differ; MI355X against the second run: (\d+)", r.stdout)
    assert m and m2, r.stdout[-800:]
    assert int(m.group(1)) == 20000 and int(m.group(2)) > 4_000_000     # (the contigs grew: 20 000 reads are 2.1 M letters)
    # two runs of the reference itself differ by `spread` sequences (one strand tie early on grows into some twenty by iteration 12); the
    # device result has to lie as close to one of them as they lie to each other.  The zero-tolerance check of the same twelve
    # iterations is test_twelve_iterations_against_the_oracle_chain below.
    d_first, spread, d_second = int(m.group(3)), int(m2.group(1)), int(m2.group(2))
    assert min(d_first, d_second) <= max(10, spread), (d_first, d_second, spread)

def loop_vs(*args):
    import re
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "loop_vs_ref.py")] + list(args), capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-800:])
    m = re.search(r"result (\d+) sequences, (\d+) residues, (\d+) circular set aside; (\d+) sequences differ", r.stdout)
    assert m, r.stdout[-800:]
    return [int(x) for x in m.groups()]


def test_twelve_iterations_against_the_oracle_chain():
    """100 000 mixed-length reads through all twelve iterations of the workflow loop (5 read + 7 contig iterations, the script's
    cyclecheck step after each contig iteration): `ancient_reads_loop` against the ORACLE's modules chained through DB files as
    data/nuclassemble.sh chains them.  The oracle has one deterministic strand-tie rule, so nothing may differ."""
    n, residues, _, differ = loop_vs("100000", "16", "oracle")
    assert n == 100000 and residues > 20_000_000
    assert differ == 0


def test_contig_phase_takes_its_identity_threshold_from_the_workflow_flag():
    """--min-seqid-corr-contigs is the contig phase's --min-seq-id (Nuclassembler.cpp:124-126: `par.seqIdThr = par.corrContigSeqId`
    before the contig parameter strings are made), --min-seq-id the read phase's; non-default values on both sides, zero tolerance."""
    n, _, _, differ = loop_vs("20000", "16", "oracle", "contigid=0.96", "seqid=0.93")
    assert n == 20000 and differ == 0
