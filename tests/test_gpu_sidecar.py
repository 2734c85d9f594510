"""Binary side-cars between the owned module processes (csrc/host/sidecar.h): the text DBs stay byte for byte what they
are without them - what every reference module reads -, an owned consumer that finds a side-car of the very files it is given takes it
instead of the text, and a side-car of other files (stale) is ignored."""
import os
import shutil
import subprocess

import pytest

from carpedeam_amd import mmdb
from gpuutil import gold
from stageflags import A_FLAGS, K_FLAGS, R_FLAGS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "carpedeam_amd", "carpedeam")


def run(*args, env=None):
    e = dict(os.environ, CDM_TIMING="1")
    e.update(env or {})
    r = subprocess.run([BIN] + list(args), capture_output=True, text=True, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stderr


def db_files(path):
    """name -> bytes of every file of the text DB at `path` (data file or its parts, index, dbtype) - not the side-car"""
    d, base = os.path.dirname(path), os.path.basename(path)
    out = {}
    for f in sorted(os.listdir(d)):
        if (f == base or f.startswith(base + ".")) and not f.endswith(".cdmbin"):
            out[f[len(base):]] = open(os.path.join(d, f), "rb").read()
    return out


def chain(tmp, tag, dhigh, env, name="mixed3k", it=0):
    t = lambda s: os.path.join(tmp, tag + "_" + s)
    mmdb.write_from_keyed(t("in"), gold(name, "reads") if it == 0 else gold(name, "asm", it - 1), mmdb.DBTYPE_NUCLEOTIDES)
    logs = [run("kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "4", env=env),
            run("rescorediagonal", t("in"), t("in"), t("pref"), t("aln"), *R_FLAGS, "--threads", "4", env=env),
            run("ancient_correction", t("in"), t("aln"), t("corr"), *A_FLAGS, "--ancient-damage", dhigh, "--threads", "4", env=env),
            run("ancient_read_assemble", t("corr"), t("aln"), t("asm"), *A_FLAGS, "--ancient-damage", dhigh, "--threads", "4", env=env)]
    return t, logs


@pytest.mark.parametrize("name,it", [("mixed3k", 0), ("letters", 0), ("letters", 1), ("synth2k", 0)])
def test_text_dbs_are_identical_with_and_without_side_cars(tmp_path, dhigh_prefix, name, it):
    from carpedeam_amd import build
    build.build()
    on, logs_on = chain(str(tmp_path), "on", dhigh_prefix, {}, name, it)
    off, logs_off = chain(str(tmp_path), "off", dhigh_prefix, {"CDM_SIDECAR": "0"}, name, it)
    for db in ("pref", "aln", "corr", "asm"):
        a, b = db_files(on(db)), db_files(off(db))
        assert a.keys() == b.keys() and all(a[k] == b[k] for k in a), db
        assert os.path.exists(on(db) + ".cdmbin") and not os.path.exists(off(db) + ".cdmbin"), db
    assert os.path.exists(on("in") + ".cdmbin")                  # the first module that read the input DB left its sequences packed
    # the consumers took the side-cars (their laps say so), and none was read with side-cars off
    assert "prefilter records read" in logs_on[1] and "prefilter text parsed" not in logs_on[1]
    assert "alignment records read" in logs_on[2] and "alignment records read" in logs_on[3]
    assert "_in: from the text" in logs_on[0] and "_in: from its side-car" in logs_on[1] and "_in: from its side-car" in logs_on[2] and "_corr: from its side-car" in logs_on[3]
    assert all("records read" not in l for l in logs_off) and "alignment text parsed" in logs_off[2]


def test_a_side_car_of_other_files_is_ignored(tmp_path, dhigh_prefix):
    """the prefilter DB is replaced behind kmermatcher's back (another k: other hits) while its side-car stays: rescorediagonal must follow
    the text; the same for an alignment DB and for a sequence DB"""
    from carpedeam_amd import build
    build.build()
    t = lambda s: str(tmp_path / s)
    mmdb.write_from_keyed(t("in"), gold("mixed3k", "reads"), mmdb.DBTYPE_NUCLEOTIDES)
    run("kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "4")
    k18 = [x if x != "20" else "18" for x in K_FLAGS]
    run("kmermatcher", t("in"), t("pref18"), *k18, "--threads", "4", env={"CDM_SIDECAR": "0"})
    # what rescorediagonal makes of the k = 18 hits, nothing but text involved
    run("rescorediagonal", t("in"), t("in"), t("pref18"), t("aln18"), *R_FLAGS, "--threads", "4", env={"CDM_SIDECAR": "0"})
    assert db_files(t("pref")) != db_files(t("pref18"))
    # the k = 18 files take the place of the k = 20 ones; pref.cdmbin (k = 20) stays
    for f in os.listdir(str(tmp_path)):
        if f == "pref" or (f.startswith("pref.") and not f.endswith(".cdmbin")):
            os.unlink(t(f))
    for f in os.listdir(str(tmp_path)):
        if f == "pref18" or f.startswith("pref18."):
            shutil.copy(t(f), t("pref" + f[len("pref18"):]))
    assert os.path.exists(t("pref.cdmbin"))
    log = run("rescorediagonal", t("in"), t("in"), t("pref"), t("aln"), *R_FLAGS, "--threads", "4")
    assert "prefilter text parsed" in log
    a, b = mmdb.read_db(t("aln")), mmdb.read_db(t("aln18"))
    assert a == b
    # a sequence DB rewritten in place (one letter changed: same size, another time stamp): its side-car is not taken
    run("ancient_correction", t("in"), t("aln"), t("corr"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "4")
    assert os.path.exists(t("corr.cdmbin"))
    data = bytearray(open(t("corr"), "rb").read())
    i = data.index(b"A")
    data[i:i + 1] = b"C"
    open(t("corr"), "wb").write(bytes(data))
    log = run("ancient_read_assemble", t("corr"), t("aln"), t("asm"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "4")
    assert "corr: from the text" in log and "alignment records read" in log
    run("ancient_read_assemble", t("corr"), t("aln"), t("asm_text"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "4", env={"CDM_SIDECAR": "0"})
    assert mmdb.read_db(t("asm")) == mmdb.read_db(t("asm_text"))


def test_rmdb_and_mvdb_take_the_side_car_along(tmp_path, dhigh_prefix):
    from carpedeam_amd import build
    build.build()
    t = lambda s: str(tmp_path / s)
    mmdb.write_from_keyed(t("in"), gold("synth2k", "reads"), mmdb.DBTYPE_NUCLEOTIDES)
    run("kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "4")
    run("mvdb", t("pref"), t("moved"))
    assert os.path.exists(t("moved.cdmbin")) and not os.path.exists(t("pref.cdmbin"))
    log = run("rescorediagonal", t("in"), t("in"), t("moved"), t("aln"), *R_FLAGS, "--threads", "4")
    assert "prefilter records read" in log              # (a rename keeps sizes and times)
    run("rmdb", t("moved"))
    assert not [f for f in os.listdir(str(tmp_path)) if f.startswith("moved")]


def test_createdb_leaves_the_sequence_side_car(tmp_path):
    """createdb packs upper-case ACGTN reads on the host exactly as the device's upload does (kmermatcher on the side-car = kmermatcher on the
    text); a read set with any other letter gets its side-car from the first module that uploads it"""
    import numpy as np
    from carpedeam_amd import build, synth
    build.build()
    t = lambda s: str(tmp_path / s)
    rng = np.random.default_rng(7)
    reads = synth.generate_strings(3000, seed=21, mixed=(30, 150))
    reads = ["".join("N" if rng.random() < 0.01 else c for c in s) for s in reads]
    open(t("r.fa"), "w").write("".join(">r%d\n%s\n" % (i, s) for i, s in enumerate(reads)))
    run("createdb", t("r.fa"), t("db"), "--shuffle", "0")
    assert os.path.exists(t("db.cdmbin"))
    log = run("kmermatcher", t("db"), t("pref"), *K_FLAGS, "--threads", "4")
    assert "db: from its side-car" in log
    run("kmermatcher", t("db"), t("pref_text"), *K_FLAGS, "--threads", "4", env={"CDM_SIDECAR": "0"})
    assert db_files(t("pref")) == db_files(t("pref_text"))
    log = run("rescorediagonal", t("db"), t("db"), t("pref"), t("aln"), *R_FLAGS, "--threads", "4")
    run("rescorediagonal", t("db"), t("db"), t("pref_text"), t("aln_text"), *R_FLAGS, "--threads", "4", env={"CDM_SIDECAR": "0"})
    assert db_files(t("aln")) == db_files(t("aln_text"))
    # lower-case letters: the device's letter mapping and raw plane are not restated on the host
    open(t("l.fa"), "w").write("".join(">r%d\n%s\n" % (i, s if i % 7 else s.lower()) for i, s in enumerate(reads[:500])))
    run("createdb", t("l.fa"), t("ldb"), "--shuffle", "0")
    assert not os.path.exists(t("ldb.cdmbin"))
    log = run("kmermatcher", t("ldb"), t("lpref"), *K_FLAGS, "--threads", "4")
    assert "ldb: from the text" in log and os.path.exists(t("ldb.cdmbin"))


@pytest.mark.parametrize("name", ["mixed3k", "letters"])
def test_streamed_sequence_dbs_are_the_same_files(tmp_path, dhigh_prefix, name):
    """A sequence DB's text goes from the device into its data file piece by piece where that pays (tmpfs; CDM_STREAM_DB=1 forces it anywhere,
    =0 never: csrc/host/main.cpp streamSeqDb, cdm_seqdb_download_stream) - the same files, byte for byte, as the whole text downloaded and
    then written, side-cars included in what the next module reads"""
    from carpedeam_amd import build
    build.build()
    on, logs_on = chain(str(tmp_path), "stream", dhigh_prefix, {"CDM_STREAM_DB": "1"}, name)
    off, logs_off = chain(str(tmp_path), "whole", dhigh_prefix, {"CDM_STREAM_DB": "0"}, name)
    for db in ("corr", "asm"):
        a, b = db_files(on(db)), db_files(off(db))
        assert a.keys() == b.keys() and all(a[k] == b[k] for k in a), db
        assert os.path.exists(on(db) + ".cdmbin")
    assert "sequences down into the DB's data file" in logs_on[2] and "sequences down into the DB's data file" in logs_on[3]
    assert "sequences down into the DB's data file" not in logs_off[2]
    assert "_corr: from its side-car" in logs_on[3]


def test_download_stream_pieces_equal_the_whole_download():
    """cdm_seqdb_download_stream through the C ABI: pieces of 1 MB of a DB with empty sequences in it, against cdm_seqdb_download"""
    import ctypes as C
    import numpy as np
    from carpedeam_amd import capi
    ctx = capi.Ctx(0)
    rng = np.random.default_rng(3)
    seqs = [b"" if i % 97 == 0 else bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), int(rng.integers(1, 400)), p=[.24, .24, .24, .24, .04])) for i in range(20000)]
    db = ctx.upload_seqs(seqs)
    lens, _, _ = db.meta()
    offs = np.zeros(db.n, np.uint64)
    offs[1:] = np.cumsum(lens[:-1].astype(np.uint64) + 2)
    total = int(offs[-1]) + int(lens[-1]) + 1
    whole = np.zeros(total + 1, np.uint8)
    db.download_into(whole, offs)
    got = np.zeros(total + 1, np.uint8)
    pieces = []
    SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64)

    def sink(user, data, offset, nbytes):
        got[offset:offset + nbytes] = np.frombuffer(C.string_at(data, nbytes), np.uint8)
        pieces.append((int(offset), int(nbytes)))
        return 0

    fn = capi.lib().cdm_seqdb_download_stream
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, SINK, C.c_void_p]
    assert fn(ctx.h, db.h, offs.ctypes.data, 1 << 20, SINK(sink), None) == 0
    assert len(pieces) > 3 and pieces[0][0] == 0 and all(pieces[i][0] + pieces[i][1] == pieces[i + 1][0] for i in range(len(pieces) - 1))
    assert np.array_equal(got[:total], whole[:total])
