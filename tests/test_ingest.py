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


def odd_inputs(tmp_path):
    """malformed and unusual FASTA/FASTQ, one file per case: what kseq (lib/mmseqs/lib/ksw2/kseq.h:184-233) makes of it is the contract"""
    cases = {
        "ws_in_names.fa": ">n1\vvertical tab ends the name\nACGT\n>n2\fform feed\nGGCC\n>n3\rcarriage return inside\nTTAA\n>n4\tt\nAC\n>n5 \r\nACGTAC\n>n6  two blanks\nAC\n",
        "garbage_before_first.fa": "junk line\nmore junk >mid-line start\nACGT\nAC\n>real one\nGGGG\n",
        "crlf_multiline.fa": ">a x\r\nACGT\r\nAC\r\n\r\nGT\r\n>b\r\nA\r\n>c\r\n\r\n>d\r\nACGTNN\r\n",
        "multiline.fq": "@q1 c\nACGT\nACGT\n+\nIIII\nIIII\n@q2\nGG\n+q2 again\nII\n\n\n@q3\nTTTT\n+\nII\nII\n",
        "qual_too_long.fq": "@ok1\nACGT\n+\nIIII\n@bad\nACGT\n+\nIIIIII\n@never_seen\nGGGG\n+\nIIII\n",
        "qual_cut_short.fq": "@ok1\nACGT\n+\nIIII\n@ok2\nAC\n+\nII\n@cut\nACGTACGT\n+\nIII",
        "at_in_quality.fq": "@r1\nACGTACGT\n+\n@@@@IIII\n@r2\nGGGG\n+\n@III\n",
        "junk_after_quality.fq": "@r1\nACGT\n+\nIIII\nstray text then @r2 header mid-line\nGGCC\n+\nIIII\n",
        "no_final_newline.fa": ">x\nACGT\n>y only a header",
        "plus_at_end.fq": "@r1\nACGT\n+\nIIII\n@r2\nGG\n+",
    }
    paths = []
    for name, text in cases.items():
        p = str(tmp_path / name)
        open(p, "wb").write(text.encode("latin1"))
        paths.append(p)
    return paths


def test_createdb_on_odd_inputs_equals_reference(exe, tmp_path):
    want = json.load(open(os.path.join(GOLD, "example", "createdb_digests.json")))["odd"]
    for p in odd_inputs(tmp_path):
        name = os.path.basename(p)
        run(exe, "createdb", p, str(tmp_path / ("m_" + name)), "--shuffle", "0", "--dbtype", "2")
        got = {ext: digest(str(tmp_path / ("m_" + name)) + ext) for ext in DB_FILES}
        assert got == want[name], name
        if os.path.exists(REF):
            run(REF, "createdb", p, str(tmp_path / ("r_" + name)), "--shuffle", "0", "--dbtype", "2", "-v", "0")
            assert got == {ext: digest(str(tmp_path / ("r_" + name)) + ext) for ext in DB_FILES}, name


@pytest.mark.parametrize("threads", ["2", "5", "16"])
def test_parallel_parse_equals_serial(exe, tmp_path, threads, monkeypatch):
    """plain files are parsed by all threads on stretches of the file whose starts are guessed and then confirmed by the parser of
    the stretch in front (host/ingest.cpp: parsePlainParallel); forced here on small files: whatever the guesses hit, the DB is the serial one"""
    fq, seqs = example_fastq(tmp_path)
    fa = tricky_fasta(tmp_path)
    want = json.load(open(os.path.join(GOLD, "example", "createdb_digests.json")))
    monkeypatch.setenv("CDM_INGEST_PAR_MIN", "1")
    monkeypatch.setenv("CDM_TIMING", "1")
    r = subprocess.run([exe, "createdb", fq, str(tmp_path / "p"), "--shuffle", "1", "--threads", threads], capture_output=True, text=True)
    assert r.returncode == 0 and "parsed by all threads" in r.stderr            # (well-formed four-line FASTQ: the parallel path holds)
    monkeypatch.delenv("CDM_INGEST_PAR_MIN")
    run(exe, "createdb", fq, str(tmp_path / "s"), "--shuffle", "1", "--threads", threads)
    assert {x: digest(str(tmp_path / "p") + x) for x in DB_FILES} == {x: digest(str(tmp_path / "s") + x) for x in DB_FILES}
    monkeypatch.setenv("CDM_INGEST_PAR_MIN", "1")
    for p in odd_inputs(tmp_path) + [fa]:
        name = os.path.basename(p)
        run(exe, "createdb", p, str(tmp_path / ("m_" + name)), "--shuffle", "0", "--dbtype", "2", "--threads", threads)
        if name in want["odd"]:
            assert {ext: digest(str(tmp_path / ("m_" + name)) + ext) for ext in DB_FILES} == want["odd"][name], name


def test_createdb_rejects_an_entry_without_a_name(exe, tmp_path):
    """a '>' followed by white space is an entry with an empty name: "Fasta entry N is invalid" and a non-zero exit (createdb.cpp:154-157)"""
    p = str(tmp_path / "bad.fa")
    open(p, "w").write(">ok\nACGT\njunk > not a name\nACGT\n")      # (the second '>' is found character-wise after... no: line-wise, it is sequence)
    run(exe, "createdb", p, str(tmp_path / "fine"), "--dbtype", "2")
    open(p, "w").write("junk > no name here\nACGT\n>ok\nACGT\n")
    for binary in [exe] + ([REF] if os.path.exists(REF) else []):
        r = subprocess.run([binary, "createdb", p, str(tmp_path / "out"), "--dbtype", "2"], capture_output=True, text=True)
        assert r.returncode != 0 and "Fasta entry 0 is invalid" in (r.stdout + r.stderr)


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
    # what the workflow really hands over (data/nuclassemble.sh:222-241): a cycle "DB" that is nothing but an index file written by
    # awk - createhdb opens both DBs with USE_INDEX only (src/util/createhdb.cpp:21-31)
    mmdb.write_seqdb(str(tmp_path / "asm2"), seqs[:50])
    open(str(tmp_path / "cyc_only.index"), "w").write(open(str(tmp_path / "cyc.index")).read())
    run(exe, "createhdb", str(tmp_path / "asm2"), str(tmp_path / "cyc_only"), str(tmp_path / "asm2"))
    assert digest(str(tmp_path / "asm2_h")) == got["asm_h"] and digest(str(tmp_path / "asm2_h.index")) == got["asm_h.index"]
    os.remove(str(tmp_path / "asm2"))                       # ... and the sequence DB's data file is not opened either
    run(exe, "createhdb", str(tmp_path / "asm2"), str(tmp_path / "cyc_only"), str(tmp_path / "asm2"))
    assert digest(str(tmp_path / "asm2_h")) == got["asm_h"]
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


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref (the reference's object code) is not built here")
def test_ingest_keeps_letters_beyond_acgtn(exe, tmp_path):
    """soft-masked lower case, IUPAC codes, U and stray bytes go into the DB as they stand (the modules map them):
    createdb on FASTA and FASTQ and convert2fasta, byte for byte the reference's files"""
    t = lambda s: str(tmp_path / s)
    with open(t("in.fa"), "w") as f:
        f.write(">r1 soft-masked\n" + "ACGTacgtNN" * 3 + "RY\nacgtnnnn\n>r2\n" + "GGGGccccTTTTaaaa" * 2 + "*\n>r3\nACGU\n>r4 x\n" + "acgt" * 20 + "swbdhvK\n")
    with open(t("in.fq"), "w") as f:      # (the first read has 2 of 11 letters beyond ACGTUN: createdb's guess says amino acids, dbtype 0)
        f.write("@q1\nacgtACGTNRY\n+\nIIIIIIIIIII\n@q2\nGGGGcccc\n+\nIIIIIIII\n")
    for src in ("in.fa", "in.fq"):
        run(exe, "createdb", t(src), t("g_" + src), "-v", "0")
        run(REF, "createdb", t(src), t("r_" + src), "-v", "0")
        for ext in ("", ".index", "_h", "_h.index", ".lookup", ".dbtype"):
            assert open(t("g_" + src) + ext, "rb").read() == open(t("r_" + src) + ext, "rb").read(), (src, ext)
    assert ("ACGTacgtNN" * 3 + "RYacgtnnnn\n\0").encode() in open(t("g_in.fa"), "rb").read()
    assert open(t("g_in.fa.dbtype"), "rb").read()[0] == 1 and open(t("g_in.fq.dbtype"), "rb").read()[0] == 0
    # the modules of this path refuse what is not a nucleotide DB (the reference would run its protein mode on it)
    r = subprocess.run([exe, "kmermatcher", t("g_in.fq"), t("pref"), "-k", "20"], capture_output=True, text=True)
    assert r.returncode != 0 and ("nucleotide sequence DBs only" in r.stderr or "MI355X device" in r.stderr)
    run(exe, "createdb", t("in.fq"), t("g2"), "--dbtype", "2", "-v", "0")
    assert open(t("g2.dbtype"), "rb").read()[0] == 1
    run(exe, "convert2fasta", t("g_in.fa"), t("g.fasta"), "-v", "0")
    run(REF, "convert2fasta", t("r_in.fa"), t("r.fasta"), "-v", "0")
    assert open(t("g.fasta"), "rb").read() == open(t("r.fasta"), "rb").read()


def test_index_lines_without_three_columns_are_skipped(exe, tmp_path):
    """A blank or cut-off line of a hand-made index is no entry (DBReader needs its three columns): the DB reads as without it,
    whatever the number of reader threads (the index is parsed in slices)."""
    db = str(tmp_path / "db")
    run(exe, "createdb", example_fastq(tmp_path)[0], db)
    run(exe, "convert2fasta", db, str(tmp_path / "clean.fasta"))
    for name in ("db.index", "db_h.index"):
        lines = open(str(tmp_path / name)).read().split("\n")
        lines[3:3] = ["", "17"]                        # in the middle: a blank line and a line with one column
        open(str(tmp_path / name), "w").write("\n".join(lines) + "\n\n")       # and a blank line behind the last entry
    for threads in ("1", "3", "8"):
        out = str(tmp_path / ("holes%s.fasta" % threads))
        run(exe, "convert2fasta", db, out, "--threads", threads)
        assert digest(out) == digest(str(tmp_path / "clean.fasta"))
