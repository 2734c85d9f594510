"""`align` of the MI355X host binary (csrc/host/align.cpp: linclust's gapped step on the assembled contigs - banded ksw2 extension
restated lane by lane, the wrapped ungapped seed, ALP's gapped E-values) against the reference's object code (oracle/_ref/carpedeam_full)
on random contig sets: copies with substitutions, insertions and deletions, rotations of circular contigs, reverse complements,
fragments, N / IUPAC / lower-case letters; prefilter lists by the reference's kmermatcher with linclust's flags.  Alignment DB text must
be identical (scores, identities, E-values, coordinates, order of the records)."""
import os
import subprocess

import numpy as np
import pytest

from carpedeam_amd import mmdb
from stageflags import LINCLUST_K_FLAGS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_FULL = os.path.join(ROOT, "oracle", "_ref", "carpedeam_full")
EXE = os.path.join(ROOT, "carpedeam_amd", "carpedeam_mi355x")
pytestmark = pytest.mark.skipif(not os.path.exists(REF_FULL), reason="oracle/_ref (the reference's object code) is not built here")
ALIGN_FLAGS = ("-a 0 --alignment-mode 2 --alignment-output-mode 0 --wrapped-scoring %d -e 0.001 --min-seq-id %s --min-aln-len 0 --seq-id-mode 0 --alt-ali 0 -c %s --cov-mode %d "
               "--max-seq-len 200000 --comp-bias-corr 1 --max-rejected 2147483647 --max-accept 2147483647 --add-self-matches 0 --db-load-mode 0 --pca 1 --pcb 1.5 --score-bias 0 "
               "--realign 0 --realign-score-bias -0.2 --realign-max-seqs 2147483647 --gap-open 5 --gap-extend 2 --zdrop 200 --threads 1 --compressed 0 -v 0")


@pytest.fixture(scope="module")
def exe():
    from carpedeam_amd import build
    build.build()
    return EXE


def mutate(rng, s, sub, indel):
    out = []
    for c in s:
        r = rng.random()
        if r < indel / 2:
            continue                                    # deletion
        if r < indel:
            out.append("ACGT"[rng.integers(0, 4)])      # insertion in front
        out.append("ACGT"[rng.integers(0, 4)] if rng.random() < sub else c)
    return "".join(out)


def revcomp(s):
    return s.translate(str.maketrans("ACGTacgt", "TGCAtgca"))[::-1]


def contig_set(rng, n_base, lo, hi):
    seqs = []
    for _ in range(n_base):
        L = int(rng.integers(lo, hi))
        base = "".join("ACGT"[i] for i in rng.integers(0, 4, L))
        seqs.append(base)
        for _ in range(int(rng.integers(1, 5))):
            kind = rng.integers(0, 6)
            v = mutate(rng, base, [0.0, 0.004, 0.01, 0.03][rng.integers(0, 4)], [0.0, 0.0, 0.002, 0.008][rng.integers(0, 4)])
            if kind == 1:
                v = revcomp(v)
            elif kind == 2 and len(v) > 40:             # a circular contig cut at another place
                k = int(rng.integers(1, len(v)))
                v = v[k:] + v[:k]
            elif kind == 3 and len(v) > 60:             # a fragment
                a = int(rng.integers(0, len(v) // 3))
                v = v[a: a + int(len(v) * rng.uniform(0.6, 1.0))]
            elif kind == 4:
                v = "".join(("N" if rng.random() < 0.004 else "R" if rng.random() < 0.002 else c.lower() if rng.random() < 0.01 else c) for c in v)
            elif kind == 5:
                v = v + "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(1, 40))))      # an overhang
            if v:
                seqs.append(v)
    order = rng.permutation(len(seqs))
    return [seqs[i] for i in order]


@pytest.mark.parametrize("case,wrapped,seqid,cov,covmode", [(0, 1, "0.97", "0.99", 1), (1, 1, "0.9", "0.8", 1), (2, 0, "0.9", "0.8", 0), (3, 1, "0.5", "0.3", 2), (4, 1, "0.97", "0.99", 1),
                                                          (5, 0, "0.95", "0.5", 1), (6, 1, "0.8", "0.9", 0), (7, 1, "0.97", "0.99", 1),
                                                          # contigs beyond 65 536 letters: diagonals that wrap the prefilter's 16 bits (the probe loops of the seed)
                                                          (8, 1, "0.97", "0.99", 1), (9, 0, "0.9", "0.8", 1), (10, 1, "0.9", "0.5", 1), (11, 1, "0.97", "0.9", 1), (12, 1, "0.6", "0.6", 0),
                                                          (13, 0, "0.97", "0.99", 1), (14, 1, "0.99", "0.99", 1), (15, 1, "0.3", "0.2", 2)])
def test_align_equals_reference(exe, tmp_path, case, wrapped, seqid, cov, covmode):
    rng = np.random.default_rng(100 + case)
    seqs = contig_set(rng, 4, 66000, 90000) if case in (8, 9) else contig_set(rng, 14, *([(60, 400), (200, 3000), (30, 900), (500, 6000)][case % 4]))
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("db"), seqs)
    kflags = [f if f != "0.99" else cov for f in LINCLUST_K_FLAGS]
    kflags[kflags.index("--cov-mode") + 1] = str(covmode)
    r = subprocess.run([REF_FULL, "kmermatcher", t("db"), t("pref")] + kflags + ["--threads", "1", "-v", "0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1000:]
    flags = (ALIGN_FLAGS % (wrapped, seqid, cov, covmode)).split()
    for out, binary in (("mine", exe), ("ref", REF_FULL)):
        r = subprocess.run([binary, "align", t("db"), t("db"), t("pref"), t(out)] + flags, capture_output=True, text=True)
        assert r.returncode == 0, (out, r.stderr[-1500:])
    got, want = mmdb.read_db(t("mine")), mmdb.read_db(t("ref"))
    n_rec = sum(v[0].count(b"\n") for v in want.values())
    gapped = sum(1 for v in want.values() for l in v[0].split(b"\n") if l and abs(int(l.split(b"\t")[5]) - int(l.split(b"\t")[4])) != abs(int(l.split(b"\t")[8]) - int(l.split(b"\t")[7])))
    bad = [k for k in set(got) | set(want) if got.get(k) != want.get(k)]
    assert not bad, (len(bad), n_rec, [(k, got.get(k), want.get(k)) for k in bad[:2]])
    assert n_rec > len(seqs) // 2
    if case in (1, 3):
        assert gapped > 0                               # (records whose two spans differ: alignments with gaps)
    assert mmdb.read_dbtype(t("mine")) == mmdb.read_dbtype(t("ref"))


def test_gapped_gumbel_parameters_are_the_reference_objects(exe):
    """the constants of csrc/host/evalue.cpp (ALP's estimate for nucleotide.out, gap open 5, gap extend 2) against the fixture made by the
    reference's object code - and against that code itself where it is built"""
    want = dict(l.split() for l in open(os.path.join(ROOT, "tests", "golden", "functions", "alp_gapped_5_2.txt")) if len(l.split()) == 2)
    src = open(os.path.join(ROOT, "carpedeam_amd", "csrc", "host", "evalue.cpp")).read()
    for name in ("lambda", "K", "a_I", "alpha_I", "sigma", "b_I", "beta_I", "tau", "vi_y_thr", "c_y_thr"):
        assert want[name] in src, name
    ref = os.path.join(ROOT, "oracle", "_ref", "carpedeam_ref")
    if os.path.exists(ref) and os.path.exists("/root/reference/lib/mmseqs/data/nucleotide.out"):
        r = subprocess.run([ref, "probe", "alp", "/root/reference/lib/mmseqs/data/nucleotide.out", "5", "2"], capture_output=True, text=True)
        assert r.returncode == 0
        live = dict(l.split() for l in r.stdout.split("\n") if len(l.split()) == 2)
        assert all(live[k] == v for k, v in want.items())
