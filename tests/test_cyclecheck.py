"""cyclecheck (src/assembler/cyclecheck.cpp) and the workflow loop's cyclecheck() step (data/nuclassemble.sh:19-60).
CPU: the oracle against goldens made by the reference's object code (tests/golden/make_golden.py cycle).  -m gpu: the device path
against the same goldens through the C ABI, the `cyclecheck` module on DB files, and the fused loop with circular genomes."""
import os
import subprocess

import pytest

from carpedeam_amd import mmdb
from gpuutil import diff_keys, run_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = [("whole", ["--chop-cycle", "0", "--max-seq-len", "300000"]), ("chop", ["--chop-cycle", "1", "--max-seq-len", "300000"]),
            ("chop_max3000", ["--chop-cycle", "1", "--max-seq-len", "3000"]), ("defaults", [])]


def g(d, name):
    return mmdb.load_keyed(os.path.join(ROOT, "tests", "golden", d, name + ".keyed.gz"))


def test_goldens_are_what_the_generator_makes():
    import cyclecases
    seqs = cyclecases.cases()
    want = g("cycle", "in")
    assert [want[k][0].rstrip(b"\n").decode() for k in sorted(want)] == seqs
    assert 40 < len(g("cycle", "chop")) < len(seqs) - 40          # both verdicts are well represented
    assert len(g("cycle", "chop_max3000")) < len(g("cycle", "chop"))
    reads = cyclecases.circular_reads()
    want = g("circ", "reads")
    assert [want[k][0].rstrip(b"\n").decode() for k in sorted(want)] == reads


@pytest.mark.parametrize("name,flags", VARIANTS)
def test_oracle_cyclecheck_matches_reference_goldens(oracle_bin, tmp_path, name, flags):
    t = lambda s: str(tmp_path / s)
    mmdb.write_from_keyed(t("in"), g("cycle", "in"), mmdb.DBTYPE_NUCLEOTIDES)
    run_oracle(oracle_bin, "cyclecheck", t("in"), t("out"), *flags)
    assert not diff_keys(mmdb.read_db(t("out")), g("cycle", name))


def test_oracle_cyclecheck_on_loop_contigs(oracle_bin, tmp_path):
    """the contigs the reference's loop produced (circ goldens): the cut circular ones of the last step are found again in the
    final result, and running the oracle on the final result flags nothing that was cut properly twice"""
    final = g("circ", "final")
    cyc = g("circ", "cyc_6")
    assert cyc and all(final[k] == v for k, v in cyc.items())


@pytest.mark.gpu
@pytest.mark.parametrize("name,flags", VARIANTS)
def test_device_cyclecheck_matches_reference_goldens(name, flags):
    from carpedeam_amd import capi
    from gpuutil import seqdb_to_keyed
    ctx = capi.Ctx(0)
    src = g("cycle", "in")
    db = ctx.upload_keyed_seqdb(src)
    f = dict(zip(flags[::2], flags[1::2]))
    cyc, rest, split = ctx.cyclecheck(db, int(f.get("--max-seq-len", 65535)), f.get("--chop-cycle", "0") == "1")
    want = g("cycle", name)
    got = seqdb_to_keyed(*cyc.download())
    assert not diff_keys(got, want)
    other = seqdb_to_keyed(*rest.download())
    assert not diff_keys(other, {k: v for k, v in src.items() if k not in want})
    keys = sorted(src)
    assert [keys[i] for i in range(len(keys)) if split[i]] == sorted(want)
    if f.get("--chop-cycle") == "1":
        assert all(len(want[keys[i]][0]) - 1 == split[i] for i in range(len(keys)) if split[i])


REF = os.path.join(ROOT, "oracle", "_ref", "carpedeam_ref")


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref (the reference's object code) is not built here")
@pytest.mark.parametrize("name,flags", VARIANTS)
def test_oracle_cyclecheck_on_letters_matches_reference(oracle_bin, tmp_path, name, flags):
    """contigs with lower-case stretches, IUPAC codes and non-letters: the oracle against a live run of the reference's object code"""
    import cyclecases
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in"), cyclecases.letter_cases())
    run_oracle(oracle_bin, "cyclecheck", t("in"), t("o"), *flags)
    run_oracle(REF, "cyclecheck", t("in"), t("r"), *flags, "--threads", "2")
    assert mmdb.read_db(t("r")) and not diff_keys(mmdb.read_db(t("o")), mmdb.read_db(t("r")))


@pytest.mark.gpu
@pytest.mark.parametrize("name,flags", VARIANTS)
def test_device_cyclecheck_on_letters_matches_oracle(oracle_bin, tmp_path, name, flags):
    import cyclecases
    from carpedeam_amd import capi
    from gpuutil import seqdb_to_keyed
    t = lambda s: str(tmp_path / s)
    seqs = cyclecases.letter_cases()
    mmdb.write_seqdb(t("in"), seqs)
    run_oracle(oracle_bin, "cyclecheck", t("in"), t("o"), *flags)
    ctx = capi.Ctx(0)
    f = dict(zip(flags[::2], flags[1::2]))
    cyc, rest, split = ctx.cyclecheck(ctx.upload_seqs(seqs), int(f.get("--max-seq-len", 65535)), f.get("--chop-cycle", "0") == "1")
    want = mmdb.read_db(t("o"))
    assert want and not diff_keys(seqdb_to_keyed(*cyc.download()), want)
    assert not diff_keys(seqdb_to_keyed(*rest.download()), {k: v for k, v in mmdb.read_db(t("in")).items() if k not in want})


@pytest.mark.gpu
def test_cyclecheck_module_on_db_files(tmp_path):
    from carpedeam_amd import build
    build.build()
    exe = os.path.join(ROOT, "carpedeam_amd", "carpedeam")
    t = lambda s: str(tmp_path / s)
    mmdb.write_from_keyed(t("in"), g("cycle", "in"), mmdb.DBTYPE_NUCLEOTIDES)
    for name, flags in VARIANTS:
        r = subprocess.run([exe, "cyclecheck", t("in"), t(name), *flags, "--threads", "3", "-v", "0"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1000:]
        assert not diff_keys(mmdb.read_db(t(name)), g("cycle", name))
        assert mmdb.read_dbtype(t(name)) == mmdb.DBTYPE_NUCLEOTIDES
    r = subprocess.run([exe, "cyclecheck", t("in"), t("x"), "-k", "22"], capture_output=True, text=True)
    assert r.returncode != 0 and "Unrecognized parameter" in r.stderr
    # no circular contig at all: an empty DB, as the reference writes it
    import cyclecases
    import numpy as np
    mmdb.write_seqdb(t("lin"), [bytes(bytearray(cyclecases.rnd(np.random.RandomState(3), 300))).decode()])
    r = subprocess.run([exe, "cyclecheck", t("lin"), t("none")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1000:]
    assert mmdb.read_db(t("none")) == {}


@pytest.mark.gpu
def test_fused_loop_sets_circular_contigs_aside(dhigh_prefix, tmp_path):
    """three read iterations + four contig iterations on reads from circular genomes: the loop's result (linear contigs + the cut
    circular ones) equals what the reference's modules give when chained as data/nuclassemble.sh chains them"""
    from carpedeam_amd import build
    build.build()
    exe = os.path.join(ROOT, "carpedeam_amd", "carpedeam")
    t = lambda s: str(tmp_path / s)
    mmdb.write_from_keyed(t("in"), g("circ", "reads"), mmdb.DBTYPE_NUCLEOTIDES)
    r = subprocess.run([exe, "ancient_reads_loop", t("in"), t("out"), "--ancient-damage", dhigh_prefix, "--num-iter-reads-only", "3", "--num-iterations", "7"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1000:]
    assert "circular contigs set aside" in r.stderr
    got, want = mmdb.read_db(t("out")), g("circ", "final")
    bad = diff_keys(got, want)
    assert not bad, (len(bad), bad[:5])
    # the same over three ranks (the library's RCCL transport over its stand-in, one device): every rank runs the cycle check on the whole
    # merged DB, rank 0 keeps the circular contigs
    r = subprocess.run([exe, "ancient_reads_loop", t("in"), t("out3"), "--ancient-damage", dhigh_prefix, "--num-iter-reads-only", "3", "--num-iterations", "7", "--gpus", "3"],
                       capture_output=True, text=True, env=dict(os.environ, CDM_LOOP_TRANSPORT="standin"))
    assert r.returncode == 0, r.stderr[-1000:]
    assert "circular contigs set aside" in r.stderr and "on 3 ranks" in r.stderr
    assert not diff_keys(mmdb.read_db(t("out3")), want)
    r = subprocess.run([exe, "ancient_reads_loop", t("in"), t("out0"), "--ancient-damage", dhigh_prefix, "--num-iter-reads-only", "3", "--num-iterations", "7", "--cycle-check", "0"],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "set aside" not in r.stderr
    assert diff_keys(mmdb.read_db(t("out0")), want)
