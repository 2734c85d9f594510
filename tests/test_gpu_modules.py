"""The drop-in surface: the `carpedeam` host binary run module by module on DB files with the reference's argv/flags;
output DBs must equal the reference's goldens key by key (payload and wasExtended flag)."""
import os
import subprocess

import pytest

from carpedeam_amd import mmdb
from gpuutil import diff_keys, gold, stage_input
from stageflags import A_FLAGS, HAMMING_FLAGS, K_FLAGS, LINCLUST_K_FLAGS, R_FLAGS
from test_oracle_golden import pref_sign_ties

pytestmark = pytest.mark.gpu
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "carpedeam_amd", "carpedeam")


def run(*args):
    r = subprocess.run([BIN] + list(args), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


@pytest.mark.parametrize("name,it", [("synth2k", 0), ("synth2k", 1), ("mixed3k", 2), ("example", 0), ("letters", 0), ("letters", 2)])
def test_modules_reproduce_reference_dbs(tmp_path, dhigh_prefix, name, it):
    from carpedeam_amd import build
    build.build()
    t = lambda s: str(tmp_path / s)
    mmdb.write_from_keyed(t("in"), stage_input(name, it), mmdb.DBTYPE_NUCLEOTIDES)
    run("kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "4")
    ties, bad = pref_sign_ties(mmdb.canon(mmdb.read_db(t("pref"))), mmdb.canon(gold(name, "pref", it)))
    assert not bad and sum(n for _, n in ties) <= 1
    assert mmdb.read_dbtype(t("pref")) == mmdb.DBTYPE_PREFILTER_REV_RES
    # downstream modules consume the reference's own upstream DBs (stage isolation, as for the oracle)
    mmdb.write_from_keyed(t("pref_ref"), gold(name, "pref", it), mmdb.DBTYPE_PREFILTER_REV_RES)
    run("rescorediagonal", t("in"), t("in"), t("pref_ref"), t("aln"), *R_FLAGS, "--threads", "4")
    assert not diff_keys(mmdb.read_db(t("aln")), gold(name, "aln", it))
    assert mmdb.read_dbtype(t("aln")) == mmdb.DBTYPE_ALIGNMENT_RES
    mmdb.write_from_keyed(t("aln_ref"), gold(name, "aln", it), mmdb.DBTYPE_ALIGNMENT_RES)
    run("ancient_correction", t("in"), t("aln_ref"), t("corr"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "4")
    assert not diff_keys(mmdb.read_db(t("corr")), gold(name, "corr", it))
    mmdb.write_from_keyed(t("corr_ref"), gold(name, "corr", it), mmdb.DBTYPE_NUCLEOTIDES)
    run("ancient_read_assemble", t("corr_ref"), t("aln_ref"), t("asm"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "4")
    assert not diff_keys(mmdb.read_db(t("asm")), gold(name, "asm", it))


def test_error_behaviour(tmp_path):
    r = subprocess.run([BIN, "ancient_correction", str(tmp_path / "nope"), str(tmp_path / "nope2"), str(tmp_path / "out")], capture_output=True, text=True)
    assert r.returncode != 0 and "Could not open" in r.stderr
    r = subprocess.run([BIN, "frobnicate"], capture_output=True, text=True)
    assert r.returncode != 0 and "Invalid Command" in r.stderr
    mmdb.write_seqdb(str(tmp_path / "s"), ["ACGT" * 10])
    mmdb.write_db(str(tmp_path / "a"), [(0, b"0\t74\t1.00\t1E-10\t0\t39\t40\t0\t39\t40\n")], mmdb.DBTYPE_ALIGNMENT_RES)
    r = subprocess.run([BIN, "ancient_correction", str(tmp_path / "s"), str(tmp_path / "a"), str(tmp_path / "o"), "--ancient-damage", str(tmp_path / "missing")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "Profile not 12 fields" in r.stderr


@pytest.mark.parametrize("name,extra", [("mixed3k", []), ("letters", []), ("letters", ["--num-iterations", "5"])])
def test_fused_reads_loop_equals_stage_by_stage(tmp_path, dhigh_prefix, oracle_bin, name, extra):
    """`ancient_reads_loop` (all iterations in one process, intermediates in HBM) ends in exactly the sequence DB the oracle's
    stage-by-stage chain (3 iterations x 4 modules on DB files, the deterministic strand-tie rule of DESIGN.md N1) ends in;
    against the reference's goldens - which chain the reference's own prefilter DBs with its run-dependent tie - only the
    sequences a sign-only prefilter difference can reach may differ, and that set is checked to be what the oracle differs in."""
    from carpedeam_amd import build
    from gpuutil import run_oracle
    build.build()
    t = lambda s: str(tmp_path / s)
    mmdb.write_from_keyed(t("in"), gold(name, "reads"), mmdb.DBTYPE_NUCLEOTIDES)
    run("ancient_reads_loop", t("in"), t("out"), "--ancient-damage", dhigh_prefix, "--num-iter-reads-only", "3", *extra)
    got = mmdb.read_db(t("out"))
    cur = t("in")
    for it in range(3):
        nxt = t("o%d" % it)
        run_oracle(oracle_bin, "kmermatcher", cur, t("opref"), *K_FLAGS, "--threads", "4")
        run_oracle(oracle_bin, "rescorediagonal", cur, cur, t("opref"), t("oaln"), *R_FLAGS, "--threads", "4")
        run_oracle(oracle_bin, "ancient_correction", cur, t("oaln"), t("ocorr"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "4")
        run_oracle(oracle_bin, "ancient_read_assemble", t("ocorr"), t("oaln"), nxt, *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "4")
        cur = nxt
    if extra:       # two contig iterations behind the reads loop (data/nuclassemble.sh:148-232), cyclecheck included
        from stageflags import AC_FLAGS, KC_FLAGS
        for it in range(3, 5):
            nxt = t("o%d" % it)
            run_oracle(oracle_bin, "kmermatcher", cur, t("opref"), *KC_FLAGS, "--threads", "4")
            run_oracle(oracle_bin, "rescorediagonal", cur, cur, t("opref"), t("oaln"), *R_FLAGS, "--threads", "4")
            run_oracle(oracle_bin, "ancient_correction", cur, t("oaln"), t("ocorr"), *AC_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "4")
            run_oracle(oracle_bin, "ancient_contig_merge", t("ocorr"), t("oaln"), nxt, *AC_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "4")
            run_oracle(oracle_bin, "cyclecheck", nxt, t("ocyc"), "--chop-cycle", "1", "--max-seq-len", "200000")
            assert mmdb.read_db(t("ocyc")) == {}        # (nothing circular in these reads: the loop's result is the merge's)
            cur = nxt
    want = mmdb.read_db(cur)
    assert not diff_keys(got, want)
    if not extra:
        assert diff_keys(got, gold(name, "asm", 2)) == diff_keys(want, gold(name, "asm", 2))


def test_reads_loop_takes_fastq(tmp_path, dhigh_prefix):
    """`ancient_reads_loop <reads.fq>`: FASTQ parsed on the host straight into the upload (no sequence DB on disk), laid out
    as createdb would - same result as createdb + the loop on the DB."""
    import gzip
    from carpedeam_amd import build
    build.build()
    t = lambda s: str(tmp_path / s)
    reads = gold("synth2k", "reads")
    with gzip.open(t("r.fq.gz"), "wt") as f:
        for k in sorted(reads):
            s = reads[k][0].decode().strip()
            f.write("@r%d\n%s\n+\n%s\n" % (k, s, "F" * len(s)))
    run("createdb", t("r.fq.gz"), t("db"))
    run("ancient_reads_loop", t("db"), t("out_db"), "--ancient-damage", dhigh_prefix, "--num-iter-reads-only", "2")
    run("ancient_reads_loop", t("r.fq.gz"), t("out_fq"), "--ancient-damage", dhigh_prefix, "--num-iter-reads-only", "2")
    assert not diff_keys(mmdb.read_db(t("out_fq")), mmdb.read_db(t("out_db")))
    run("createhdb", t("out_fq"), t("out_fq"))
    run("convert2fasta", t("out_fq"), t("out.fasta"))
    assert open(t("out.fasta")).read().count(">") == len(reads)


@pytest.mark.parametrize("extra", [[], ["--seq-id-mode", "1", "--cov-mode", "0", "-c", "0.9"], ["--min-seq-id", "0.999"], ["--seq-id-mode", "2", "--min-aln-len", "700", "--min-seq-id", "0.5"]])
def test_hamming_mode_of_rescorediagonal_on_db_files(tmp_path, oracle_bin, extra):
    """linclust's pre-clustering call of `ancient_assemble` through the host binary (lib/mmseqs/data/workflow/linclust.sh:21-31): its
    kmermatcher, then rescorediagonal --rescore-mode 0 --wrapped-scoring 1; golden from the reference's object code for the workflow's
    flags, the oracle for the other criteria."""
    from carpedeam_amd import build
    from gpuutil import GOLD, run_oracle
    build.build()
    g = os.path.join(GOLD, "hamming")
    t = lambda s: str(tmp_path / s)
    mmdb.write_from_keyed(t("in"), mmdb.load_keyed(os.path.join(g, "in.keyed.gz")), mmdb.DBTYPE_NUCLEOTIDES)
    run("kmermatcher", t("in"), t("pref"), *LINCLUST_K_FLAGS, "--threads", "4")
    ties, bad = pref_sign_ties(mmdb.canon(mmdb.read_db(t("pref"))), mmdb.canon(mmdb.load_keyed(os.path.join(g, "pref.keyed.gz"))))
    assert not bad and sum(n for _, n in ties) <= 1
    mmdb.write_from_keyed(t("pref_ref"), mmdb.load_keyed(os.path.join(g, "pref.keyed.gz")), mmdb.DBTYPE_PREFILTER_REV_RES)
    flags = list(HAMMING_FLAGS)
    for k, v in zip(extra[::2], extra[1::2]):
        flags[flags.index(k) + 1] = v
    run("rescorediagonal", t("in"), t("in"), t("pref_ref"), t("res"), *flags, "--threads", "4")
    assert mmdb.read_dbtype(t("res")) == mmdb.DBTYPE_PREFILTER_REV_RES
    if not extra:
        exp = mmdb.load_keyed(os.path.join(g, "res.keyed.gz"))
    else:
        run_oracle(oracle_bin, "rescorediagonal", t("in"), t("in"), t("pref_ref"), t("res_o"), *flags, "--threads", "4")
        exp = mmdb.read_db(t("res_o"))
    assert not diff_keys(mmdb.read_db(t("res")), exp)


def test_sequence_db_data_files_end_with_their_nul_on_a_dirty_heap(tmp_path, dhigh_prefix, oracle_bin):
    """Every entry of a sequence DB's data file is "SEQ\\n\\0" (DBWriter::writeEnd), the last one included.  The module's download
    buffer is small here (below the huge-page threshold), so it comes from the C library's heap: MALLOC_PERTURB_ fills every block
    malloc hands out with a non-zero byte, and the data FILES - not the keyed payloads - are compared byte for byte."""
    from carpedeam_amd import build
    from gpuutil import run_oracle
    build.build()
    t = lambda s: str(tmp_path / s)
    env = dict(os.environ, MALLOC_PERTURB_="165")
    mmdb.write_from_keyed(t("in"), gold("synth2k", "reads"), mmdb.DBTYPE_NUCLEOTIDES)
    mmdb.write_from_keyed(t("aln"), gold("synth2k", "aln", 0), mmdb.DBTYPE_ALIGNMENT_RES)
    dmg = ["--ancient-damage", dhigh_prefix, "--threads", "1"]
    for mod, a, b in (("ancient_correction", "in", "corr"), ("ancient_read_assemble", "corr", "asm")):
        r = subprocess.run([BIN, mod, t(a), t("aln"), t(b), *A_FLAGS, *dmg], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        run_oracle(oracle_bin, mod, t(a), t("aln"), t("o" + b), *A_FLAGS, *dmg)
        data = open(t(b), "rb").read()
        assert data[-2:] == b"\n\0"
        # one writer thread on both sides: entries in key order, so the files themselves are equal
        assert data == open(t("o" + b), "rb").read()
        assert open(t(b) + ".index").read() == open(t("o" + b) + ".index").read()
    # a result DB read back from split data files (X.0 .. X.n) keeps its sentinel as well: the parse ends at the entry's NUL
    r = subprocess.run([BIN, "ancient_reads_loop", t("in"), t("loop"), "--ancient-damage", dhigh_prefix, "--num-iter-reads-only", "1"],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(t("loop"), "rb").read()[-2:] == b"\n\0"


def test_reads_loop_over_several_ranks_equals_one_device(tmp_path, dhigh_prefix):
    """`ancient_reads_loop --gpus N`: the read iterations AND the contig iterations split over N ranks (host threads of the module,
    csrc/dist.hip cdm_reads_iteration_dist / cdm_contig_iteration_dist) - the DB of the single-device run, file for file.  On this pool's one-GPU boxes the ranks share the device and
    the collectives are the module's in-process transport (CDM_LOOP_TRANSPORT=threads) or the library's RCCL transport over its stand-in for
    RCCL's calls (=standin: 3 ranks exchange group keys, 6 ranks the k-mer tuples as well); RCCL itself runs with its one possible
    rank (CDM_LOOP_FORCE_COMM=1: every collective of the calling sequence, with itself as the only peer)."""
    from carpedeam_amd import build
    build.build()
    t = lambda s: str(tmp_path / s)
    mmdb.write_from_keyed(t("in"), gold("mixed3k", "reads"), mmdb.DBTYPE_NUCLEOTIDES)
    args = ["--ancient-damage", dhigh_prefix, "--num-iter-reads-only", "3", "--num-iterations", "5"]
    run("ancient_reads_loop", t("in"), t("one"), *args)
    for name, extra, env in (("two", ["--gpus", "2"], {"CDM_LOOP_TRANSPORT": "threads"}), ("three", ["--gpus", "3"], {"CDM_LOOP_TRANSPORT": "threads"}),
                             ("standin3", ["--gpus", "3"], {"CDM_LOOP_TRANSPORT": "standin"}), ("standin6", ["--gpus", "6"], {"CDM_LOOP_TRANSPORT": "standin"}),
                             ("rccl", [], {"CDM_LOOP_FORCE_COMM": "1"})):
        r = subprocess.run([BIN, "ancient_reads_loop", t("in"), t(name), *args, *extra], capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        assert ("on %d ranks" % (int(extra[1]) if extra else 1)) in r.stderr
        assert not diff_keys(mmdb.read_db(t(name)), mmdb.read_db(t("one"))), name
