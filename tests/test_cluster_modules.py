"""The host-side modules of `ancient_assemble`'s final redundancy reduction (linclust's tail: clust, createsubdb, filterdb, mergeclusters,
result2repseq) and the scripts' file modules (rmdb, mvdb) of the MI355X host binary (csrc/host/cluster.cpp) against the reference's
own object code (oracle/_ref/carpedeam_full): the reference's whole workflow runs once on its example reads behind a logging front
(argv[0] routing, as the product's front end does it), and every logged call of these modules is repeated with the MI355X binary on
the very DB files the reference's module read; random cluster inputs on top."""
import os
import shlex
import subprocess

import numpy as np
import pytest

from carpedeam_amd import mmdb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_FULL = os.path.join(ROOT, "oracle", "_ref", "carpedeam_full")
EXE = os.path.join(ROOT, "carpedeam_amd", "carpedeam_mi355x")
EXAMPLE = os.path.join(ROOT, "tests", "golden", "example")
OUT_POS = {"clust": 2, "createsubdb": 2, "filterdb": 1, "mergeclusters": 1, "result2repseq": 2}
pytestmark = pytest.mark.skipif(not os.path.exists(REF_FULL), reason="oracle/_ref (the reference's object code) is not built here")


@pytest.fixture(scope="module")
def exe():
    from carpedeam_amd import build
    build.build()
    return EXE


def read_index(path):
    return [tuple(l.split("\t")) for l in open(path + ".index").read().split("\n") if l]


def positional(args):
    """the positional arguments of a logged module call (every flag of these modules takes a value)"""
    pos, i = [], 0
    while i < len(args):
        if args[i].startswith("-") and len(args[i]) > 1 and not args[i][1].isdigit():
            i += 2
        else:
            pos.append(i)
            i += 1
    return pos


@pytest.fixture(scope="module")
def reference_workflow(tmp_path_factory, dhigh_prefix):
    d = tmp_path_factory.mktemp("refwf")
    log, wrap = str(d / "calls.log"), str(d / "logwrap.sh")
    open(wrap, "w").write('#!/bin/bash\nprintf "%%q " "$@" >> %s\necho >> %s\nexec -a %s %s "$@"\n' % (log, log, wrap, REF_FULL))
    os.chmod(wrap, 0o755)
    r = subprocess.run([wrap, "ancient_assemble", os.path.join(EXAMPLE, "test_data.fq.gz"), str(d / "out.fa"), str(d / "tmp"), "--ancient-damage", dhigh_prefix,
                        "--threads", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return [shlex.split(l) for l in open(log) if l.strip()]


def test_every_call_of_the_workflow_is_reproduced(exe, reference_workflow, tmp_path):
    seen = set()
    for n, call in enumerate(reference_workflow):
        mod, args = call[0], call[1:]
        if mod not in OUT_POS:
            continue
        pos = positional(args)
        want_path = args[pos[OUT_POS[mod]]]
        inputs = [args[p] for j, p in enumerate(pos) if j != OUT_POS[mod]]
        if not os.path.exists(want_path + ".index") or not all(os.path.exists(i) or os.path.exists(i + ".index") or os.path.exists(i + ".0") for i in inputs):
            continue        # (the scripts removed this step's files again: rmdb)
        got_path = str(tmp_path / ("%s_%d" % (mod, n)))
        mine = list(args)
        mine[pos[OUT_POS[mod]]] = got_path
        r = subprocess.run([exe, mod] + mine, capture_output=True, text=True)
        assert r.returncode == 0, (mod, r.stderr[-1500:])
        assert mmdb.read_dbtype(got_path) == mmdb.read_dbtype(want_path), mod
        got, want = mmdb.read_db(got_path), mmdb.read_db(want_path)
        assert got == want, (mod, [k for k in set(got) | set(want) if got.get(k) != want.get(k)][:5])
        if mod in ("clust", "createsubdb"):         # one writer thread in the reference: the index file itself is determined
            assert read_index(got_path) == read_index(want_path), mod
        else:                                        # (the reference's writer threads order the data file as they are scheduled: keys, lengths and flags)
            assert sorted((k, l, e) for k, o, l, e in read_index(got_path)) == sorted((k, l, e) for k, o, l, e in read_index(want_path)), mod
        if mod == "createsubdb" and "--subdb-mode" in args and args[args.index("--subdb-mode") + 1] == "1":
            assert os.path.islink(got_path) and os.path.realpath(got_path) == os.path.realpath(want_path)
        if mod == "createsubdb":                     # (result2repseq links them too, but the workflow's createhdb writes its own _h files over that DB afterwards)
            for suffix in ("_h", "_h.index", "_h.dbtype", ".lookup", ".source"):
                assert os.path.lexists(got_path + suffix) == os.path.lexists(want_path + suffix), (mod, suffix)
        seen.add(mod)
    assert seen == set(OUT_POS), seen


def test_clust_on_random_cluster_inputs(exe, tmp_path):
    """greedy incremental clustering on random sequence lengths (many ties) and random, asymmetric result lists"""
    rng = np.random.default_rng(5)
    t = lambda s: str(tmp_path / s)
    for case in range(25):
        n = int(rng.integers(1, 60))
        keys = sorted(rng.choice(200, n, replace=False).tolist())
        seqs = {k: ("A" * int(rng.integers(1, 6)) + "\n").encode() for k in keys}
        mmdb.write_db(t("seq"), sorted(seqs.items()), mmdb.DBTYPE_NUCLEOTIDES)
        res = []
        for k in keys:
            members = [k] if rng.random() < 0.8 else []
            members += rng.choice(keys, int(rng.integers(0, min(n, 5) + 1)), replace=False).tolist()
            res.append((k, "".join("%d\t%d\t0.9\n" % (m, rng.integers(1, 99)) for m in members).encode()))
        mmdb.write_db(t("res"), res, mmdb.DBTYPE_ALIGNMENT_RES)
        for out, binary in (("mine", exe), ("ref", REF_FULL)):
            for f in (t(out), t(out) + ".index", t(out) + ".dbtype"):
                if os.path.exists(f):
                    os.remove(f)
            r = subprocess.run([binary, "clust", t("seq"), t("res"), t(out), "--cluster-mode", "2", "--threads", "3", "-v", "0"], capture_output=True, text=True)
            assert r.returncode == 0, (out, r.stderr[-800:])
        assert open(t("mine"), "rb").read() == open(t("ref"), "rb").read(), case
        assert open(t("mine.index")).read() == open(t("ref.index")).read(), case


def test_refusals_and_file_modules(exe, tmp_path):
    t = lambda s: str(tmp_path / s)
    mmdb.write_db(t("a"), [(0, b"x\n"), (3, b"y\n")], mmdb.DBTYPE_NUCLEOTIDES)
    r = subprocess.run([exe, "clust", t("a"), t("a"), t("o"), "--cluster-mode", "0"], capture_output=True, text=True)
    assert r.returncode == 77 and "greedy" in r.stderr                      # set cover / connected component: refused before any work
    r = subprocess.run([exe, "filterdb", t("a"), t("o")], capture_output=True, text=True)
    assert r.returncode == 77
    open(t("a.lookup"), "w").write("0\tn\t0\n")
    assert subprocess.run([exe, "mvdb", t("a"), t("b")]).returncode == 0
    assert not os.path.exists(t("a")) and not os.path.exists(t("a.index")) and mmdb.read_db(t("b")) == {0: (b"x\n", 0), 3: (b"y\n", 0)} and os.path.exists(t("b.lookup"))
    assert subprocess.run([exe, "rmdb", t("b"), "-v", "3"]).returncode == 0
    assert not any(os.path.exists(t("b") + s) for s in ("", ".index", ".dbtype", ".lookup"))
