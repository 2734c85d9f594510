"""The steps either side of the hot path on the host (no GPU involved): createdb, createhdb and
convert2fasta of the MI355X host binary against the reference's own object code (oracle/_ref, when built here) byte for byte,
and against digests of the reference's output committed under tests/golden/ (made by tests/golden/make_golden.py)."""
import gzip
import hashlib
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "carpedeam_amd", "carpedeam")
REF = os.path.join(ROOT, "oracle", "_ref", "carpedeam_ref")
GOLD = os.path.join(ROOT, "tests", "golden")
DB_FILES = ("", ".index", ".dbtype", ".lookup", ".source", "_h", "_h.index", "_h.dbtype")


@pytest.fixture(scope="module")
def exe():
    from carpedeam_amd import build
    build.build()
    return EXE


def run(binary, *args):
    r = subprocess.run([binary] + list(args), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]


def digest(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def example_fastq(tmp_path):
    """the reference's example reads (tests/golden/example/reads.keyed.gz holds them in file order) as a FASTQ file"""
    rows = [l.rstrip("\n") for l in gzip.open(os.path.join(GOLD, "example", "reads.keyed.gz"), "rt")]
    seqs = [rows[i] for i in range(1, len(rows), 2)]
    p = str(tmp_path / "reads.fq")
    with open(p, "w") as f:
        for i, s in enumerate(seqs):
            f.write("@read_%d sample=%d\n%s\n+\n%s\n" % (i, i % 7, s, "I" * len(s)))
    return p, seqs


def tricky_fasta(tmp_path):
    p = str(tmp_path / "multi.fa")
    with open(p, "w") as f:
        f.write(">sp|P12345|NAME_X some protein-like header\nACGTACGTAC\nGGGTTTAA\n\n>plain\r\nACGT\r\nNNNN\r\n>gi|123|ref|XY_1.1| thing\nAC\n>lonely_header_no_seq\n>last\nTTTTGGGGCCCCAAAA")
    return p


@pytest.mark.parametrize("shuffle", ["1", "0"])
def test_createdb_equals_reference(exe, tmp_path, shuffle):
    fq, seqs = example_fastq(tmp_path)
    fa = tricky_fasta(tmp_path)
    gz = str(tmp_path / "second.fq.gz")
    with gzip.open(gz, "wt") as f:
        for i, s in enumerate(seqs[:100]):
            f.write("@gz_%d\n%s\n+\n%s\n" % (i, s, "#" * len(s)))
    run(exe, "createdb", fq, fa, gz, str(tmp_path / "mine"), "--shuffle", shuffle)
    got = {ext: digest(str(tmp_path / "mine") + ext) for ext in DB_FILES}
    # committed digests of the reference's createdb on exactly these inputs
    want = json.load(open(os.path.join(GOLD, "example", "createdb_digests.json")))["shuffle" + shuffle]
    assert got == want
    if os.path.exists(REF):
        run(REF, "createdb", fq, fa, gz, str(tmp_path / "ref"), "--shuffle", shuffle, "-v", "0")
        assert got == {ext: digest(str(tmp_path / "ref") + ext) for ext in DB_FILES}
    # the sequence DB holds the reads in createdb's order
    from carpedeam_amd import mmdb
    db = mmdb.read_db(str(tmp_path / "mine"))
    n = len(seqs) + 5 + 100
    assert len(db) == n
    order = list(range(n)) if shuffle == "0" else [i for s in range(32) for i in range(s, n, 32)]
    assert all(db[k][0] == seqs[order[k]].encode() + b"\n" for k in range(n) if order[k] < len(seqs))


def test_createhdb_and_convert2fasta_equal_reference(exe, tmp_path):
    fq, seqs = example_fastq(tmp_path)
    run(exe, "createdb", fq, str(tmp_path / "db"), "--shuffle", "1")
    run(exe, "convert2fasta", str(tmp_path / "db"), str(tmp_path / "mine.fasta"))
    # assembled sequences get generated headers: createhdb without and with a cycle DB (a DB whose keys are the circular ones)
    from carpedeam_amd import mmdb
    mmdb.write_seqdb(str(tmp_path / "asm"), seqs[:50])
    mmdb.write_db(str(tmp_path / "cyc"), [(3, b"x\n"), (17, b"y\n")], mmdb.DBTYPE_NUCLEOTIDES)
    run(exe, "createhdb", str(tmp_path / "asm"), str(tmp_path / "asm"))
    run(exe, "convert2fasta", str(tmp_path / "asm"), str(tmp_path / "asm.fasta"))
    first = open(str(tmp_path / "asm.fasta")).read().split("\n")[:4]
    assert first == [">0 len:%d" % len(seqs[0]), seqs[0], ">1 len:%d" % len(seqs[1]), seqs[1]]
    run(exe, "createhdb", str(tmp_path / "asm"), str(tmp_path / "cyc"), str(tmp_path / "asm"))
    run(exe, "convert2fasta", str(tmp_path / "asm"), str(tmp_path / "asm_cyc.fasta"))
    lines = open(str(tmp_path / "asm_cyc.fasta")).read().split("\n")
    assert lines[0] == ">0 len:%d cycle:0" % len(seqs[0]) and lines[6] == ">3 len:%d cycle:1" % len(seqs[3])
    got = {k: digest(str(tmp_path / k)) for k in ("mine.fasta", "asm.fasta", "asm_cyc.fasta", "asm_h", "asm_h.index")}
    assert got == json.load(open(os.path.join(GOLD, "example", "createdb_digests.json")))["fasta"]
    if os.path.exists(REF):
        run(REF, "convert2fasta", str(tmp_path / "db"), str(tmp_path / "ref.fasta"), "-v", "0")
        assert digest(str(tmp_path / "ref.fasta")) == got["mine.fasta"]
        mmdb.write_seqdb(str(tmp_path / "rasm"), seqs[:50])
        run(REF, "createhdb", str(tmp_path / "rasm"), str(tmp_path / "cyc"), str(tmp_path / "rasm"), "-v", "0")
        run(REF, "convert2fasta", str(tmp_path / "rasm"), str(tmp_path / "rasm_cyc.fasta"), "-v", "0")
        assert digest(str(tmp_path / "rasm_cyc.fasta")) == got["asm_cyc.fasta"]
        assert digest(str(tmp_path / "rasm_h")) == got["asm_h"] and digest(str(tmp_path / "rasm_h.index")) == got["asm_h.index"]


def test_createdb_errors(exe, tmp_path):
    r = subprocess.run([exe, "createdb", str(tmp_path / "missing.fq"), str(tmp_path / "db")], capture_output=True, text=True)
    assert r.returncode != 0 and "Cannot open" in r.stderr
    open(str(tmp_path / "empty.fa"), "w").write("no records here\n")
    r = subprocess.run([exe, "createdb", str(tmp_path / "empty.fa"), str(tmp_path / "db")], capture_output=True, text=True)
    assert r.returncode != 0 and "have no entry" in r.stderr
    r = subprocess.run([exe, "createdb", str(tmp_path / "empty.fa"), str(tmp_path / "db"), "--dbtype", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "not supported" in r.stderr
