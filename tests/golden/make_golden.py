#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference's own object code.

Run in the build container only (needs /root/reference and oracle/_ref/carpedeam_ref built by
`make -C oracle -f Makefile.ref`).  Nothing here reads or copies reference *source*; the outputs
are data: inputs (synthetic reads from carpedeam_amd/synth.py, the reference's example reads) and
the reference binary's outputs on them (keyed DB dumps, function-level known answers).

    python tests/golden/make_golden.py
"""
import gzip
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from carpedeam_amd import mmdb, synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "carpedeam_ref")
OUT = os.path.join(ROOT, "tests", "golden")
REFROOT = "/root/reference"

# stage flags exactly as `carpedeam ancient_assemble` emits them on defaults (SURVEY.md 3.1;
# createParameterString passes every parameter of the module's list, src/workflow/Nuclassembler.cpp:103-118)
K_FLAGS = ("--kmer-per-seq 200 --kmer-per-seq-scale 0.2 --hash-shift 67 --ignore-multi-kmer 1 --mask 0 "
           "--adjust-kmer-len 0 --cov-mode 1 -c 0 --include-only-extendable 0 -k 20").split()
R_FLAGS = ("--rescore-mode 3 -e 0.001 --min-seq-id 0.9 --seq-id-mode 0 --sort-results 0 -a 0 --filter-hits 0 "
           "--cov-mode 1 -c 0").split()
A_FLAGS = ("--rescore-mode 3 --max-seq-len 200000 --min-seq-id 0.9 --ext-random-align 0.85 --excess-penalty 0.0625 "
           "--min-ryseq-id-corr-reads 0.99 --likelihood-ratio-threshold 0.5 --unsafe 0 --min-cov-safe 5").split()


def run(*args, inp=None):
    r = subprocess.run([REF] + list(args), input=inp, capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit("reference failed: %s\n%s" % (" ".join(args), r.stderr[-2000:]))
    return r.stdout


def gz_write(path, text):
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(text.encode("latin1"))


def chain(tmp, name, seqs, iterations, prefix, threads):
    """Run `iterations` read-loop iterations of the four stages; dump every DB keyed."""
    d = os.path.join(OUT, name)
    os.makedirs(d, exist_ok=True)
    inp = os.path.join(tmp, name + "_reads")
    mmdb.write_seqdb(inp, seqs)
    gz_write(os.path.join(d, "reads.keyed.gz"), mmdb.dump_keyed(inp))
    sums = {}
    for it in range(iterations):
        p = lambda s: os.path.join(tmp, "%s_%s_%d" % (name, s, it))
        # kmermatcher single-threaded: the reference's parallel sort leaves the order of equal tuples
        # (and thereby the strand sign of a hit whose best diagonal has mixed strands) run-dependent
        run("kmermatcher", inp, p("pref"), *K_FLAGS, "--threads", "1")
        run("rescorediagonal", inp, inp, p("pref"), p("aln"), *R_FLAGS, "--threads", str(threads))
        run("ancient_correction", inp, p("aln"), p("corr"), *A_FLAGS, "--ancient-damage", prefix, "--threads", str(threads))
        run("ancient_read_assemble", p("corr"), p("aln"), p("asm"), *A_FLAGS, "--ancient-damage", prefix, "--threads", str(threads))
        for s in ("pref", "aln", "corr", "asm"):
            txt = mmdb.dump_keyed(p(s))
            gz_write(os.path.join(d, "%s_%d.keyed.gz" % (s, it)), txt)
            sums["%s_%d" % (s, it)] = hashlib.sha256(txt.encode("latin1")).hexdigest()
        inp = p("asm")
    with open(os.path.join(d, "sha256.txt"), "w") as f:
        for k in sorted(sums):
            f.write("%s  %s\n" % (sums[k], k))


def probes(tmp, prefix):
    d = os.path.join(OUT, "functions")
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(12345)
    # D1: damage tables for dhigh, the all-zero template and the empty prefix
    open(os.path.join(d, "damage_dhigh.txt"), "w").write(run("probe", "damage", prefix))
    open(os.path.join(d, "damage_template.txt"), "w").write(run("probe", "damage", os.path.join(tmp, "tmpl")))
    open(os.path.join(d, "damage_empty.txt"), "w").write(run("probe", "damage", ""))
    open(os.path.join(d, "seqerr.txt"), "w").write(run("probe", "seqerr", "0.01") + run("probe", "seqerr", "0.001"))
    open(os.path.join(d, "alp.txt"), "w").write(run("probe", "alp", os.path.join(REFROOT, "lib/mmseqs/data/nucleotide.out")))
    # C2: mostLikeliBaseRead vectors: realistic pile-ups, heavy pile-ups, reverse-strand mixes, near ties
    lines = []
    for i in range(3000):
        qLen = int(rng.integers(30, 200))
        qIter = int(rng.integers(0, qLen))
        if i % 3 == 0:
            qIter = int(rng.choice([0, 1, 2, 3, 4, qLen - 5, qLen - 4, qLen - 3, qLen - 2, qLen - 1]))
        qBase = int(rng.integers(0, 4))
        wasCorr = int(rng.integers(0, 5) == 0)
        cnt = np.zeros((4, 11), dtype=int)
        rev = np.zeros((4, 11), dtype=int)
        cov = int(rng.integers(2, 9)) if i % 10 else int(rng.integers(20, 400))
        dom = qBase if rng.random() < 0.8 else int(rng.integers(0, 4))
        for _ in range(cov):
            t = dom if rng.random() < 0.85 else int(rng.integers(0, 4))
            cl = 5 if rng.random() < 0.8 else int(rng.integers(0, 11))
            cnt[t, cl] += 1
            if rng.random() < 0.4:
                rev[t, cl] += 1
        lines.append(" ".join(map(str, [qBase, qIter, qLen, wasCorr] + cnt.ravel().tolist() + rev.ravel().tolist())))
    inp = "\n".join(lines) + "\n"
    outp = run("probe", "mostlikeli", prefix, inp=inp)
    gz_write(os.path.join(d, "mostlikeli.tsv.gz"), "".join(a + "\t" + b + "\n" for a, b in zip(lines, outp.split("\n"))))
    # E3: overlap likelihoods (calcLikelihoodConsensus via r_s_pair), safe-mode consensus = N^L q N^L
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    lines = []
    for i in range(1500):
        L = int(rng.integers(40, 160))
        dbLen = int(rng.integers(40, L + 1))   # the query is the k-mer group's longest member, so dbLen <= qLen on this path
        q = acgt[rng.integers(0, 4, L)].tobytes().decode()
        aln = int(rng.integers(30, min(L, dbLen)))
        left = bool(rng.integers(0, 2))
        if left:      # query prefix overlaps the target suffix
            qs, qe, ds, de = 0, aln - 1, dbLen - aln, dbLen - 1
        else:         # query suffix overlaps the target prefix
            qs, qe, ds, de = L - aln, L - 1, 0, aln - 1
        t = list(acgt[rng.integers(0, 4, dbLen)].tobytes().decode())
        for k in range(aln):
            t[ds + k] = q[qs + k]
        for k in range(int(rng.integers(0, 4))):   # a few mismatches, biased to deamination-like ones
            pos = int(rng.integers(0, aln))
            c = q[qs + pos]
            t[ds + pos] = {"C": "T", "G": "A"}.get(c, "ACGT"[int(rng.integers(0, 4))]) if rng.random() < 0.7 else "ACGT"[int(rng.integers(0, 4))]
        if rng.random() < 0.1:
            t[int(rng.integers(0, dbLen))] = "N"
        t = "".join(t)
        cons = "N" * L + q + "N" * L
        isRev = int(rng.integers(0, 2))
        maxL = aln + int(rng.integers(0, 20))
        maxR = aln + int(rng.integers(0, 20))
        lines.append("%s %s %d %d %d %d %d %d %d %d %d %d %d 0.85 0.0625" % (cons, t, L, 7, qs, qe, ds, de, dbLen, aln, isRev, maxL, maxR))
    inp = "\n".join(lines) + "\n"
    outp = run("probe", "overlap", prefix, inp=inp)
    gz_write(os.path.join(d, "overlap.tsv.gz"), "".join(a + "\t" + b + "\n" for a, b in zip(lines, outp.split("\n"))))
    # R3: E-value / bit score table
    lines = ["%d %d" % (s, ql) for ql in (20, 35, 50, 75, 100, 150, 300, 1000, 20000) for s in range(0, 301, 3)]
    inp = "\n".join(lines) + "\n"
    txt = ""
    for dbRes in (200000, 5000000000):
        outp = run("probe", "evalue", str(dbRes), os.path.join(REFROOT, "lib/mmseqs/data/nucleotide.out"), inp=inp)
        txt += "".join("%d %s\t%s\n" % (dbRes, a, b) for a, b in zip(lines, outp.split("\n")))
    gz_write(os.path.join(d, "evalue.tsv.gz"), txt)


def main():
    if not os.path.exists(REF):
        sys.exit("build oracle/_ref first: make -C oracle -f Makefile.ref")
    with tempfile.TemporaryDirectory() as tmp:
        prefix = os.path.join(tmp, "dhigh")
        synth.write_dhigh_profiles(prefix)
        for s in ("5p", "3p"):   # our writer must reproduce the reference's example profiles byte for byte
            assert open(prefix + s + ".prof", "rb").read() == open(os.path.join(REFROOT, "example", "dhigh%s.prof" % s), "rb").read()
        with open(os.path.join(tmp, "tmpl5p.prof"), "w") as f5, open(os.path.join(tmp, "tmpl3p.prof"), "w") as f3:
            body = synth.PROF_HEADER + "\n" + ("\t".join(["0.0"] * 12) + "\n") * 5
            f5.write(body)
            f3.write(body)
        probes(tmp, prefix)
        chain(tmp, "synth2k", synth.generate_strings(2000, L=100, seed=1), 2, prefix, 4)
        chain(tmp, "mixed3k", synth.generate_strings(3000, seed=2, mixed=(60, 150)), 3, prefix, 4)
        seqs = []
        with gzip.open(os.path.join(REFROOT, "example", "test_data.fq.gz"), "rt") as f:
            for i, l in enumerate(f):
                if i % 4 == 1:
                    seqs.append(l.strip())
        chain(tmp, "example", seqs, 1, prefix, 4)


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] in ("createdb", "contigs", "cycle", "letters", "workflow", "hamming")):
    main()


def contig_goldens(threads=4):
    """tests/golden/<name>/c{pref,aln,corr,merge}_<step>.keyed.gz: the contig phase (kmermatcher -k 22 --include-only-extendable 1,
    rescorediagonal, ancient_correction, ancient_contig_merge) run by the reference's own object code, starting from the last
    reads-loop golden of the data set (python tests/golden/make_golden.py contigs)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpuutil import gold
    from stageflags import AC_FLAGS, KC_FLAGS
    with tempfile.TemporaryDirectory() as tmp:
        prefix = os.path.join(tmp, "dhigh")
        synth.write_dhigh_profiles(prefix)
        for name, last_it, steps in (("mixed3k", 2, 2), ("synth2k", 1, 2), ("letters", 2, 2)):
            if len(sys.argv) > 2 and name not in sys.argv[2:]:      # (make_golden.py contigs <name>...: only those data sets)
                continue
            d = os.path.join(OUT, name)
            cur = os.path.join(tmp, name + "_c0")
            mmdb.write_from_keyed(cur, gold(name, "asm", last_it), mmdb.DBTYPE_NUCLEOTIDES)
            for step in range(steps):
                p = lambda s: os.path.join(tmp, "%s_%s_%d" % (name, s, step))
                run("kmermatcher", cur, p("cpref"), *KC_FLAGS, "--threads", "1")
                run("rescorediagonal", cur, cur, p("cpref"), p("caln"), *R_FLAGS, "--threads", str(threads))
                run("ancient_correction", cur, p("caln"), p("ccorr"), *AC_FLAGS, "--ancient-damage", prefix, "--threads", str(threads))
                run("ancient_contig_merge", p("ccorr"), p("caln"), p("cmerge"), *AC_FLAGS, "--ancient-damage", prefix, "--threads", str(threads))
                for s_ in ("cpref", "caln", "ccorr", "cmerge"):
                    gz_write(os.path.join(d, "%s_%d.keyed.gz" % (s_, step)), mmdb.dump_keyed(p(s_)))
                cur = p("cmerge")


def createdb_digests():
    """tests/golden/example/createdb_digests.json: sha256 of what the reference's createdb / createhdb / convert2fasta write for
    the inputs tests/test_ingest.py builds (run separately: python tests/golden/make_golden.py createdb)"""
    import gzip as gz
    import hashlib
    import json
    import pathlib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_ingest as T
    out = {}
    with tempfile.TemporaryDirectory() as d:
        d = pathlib.Path(d)
        fq, seqs = T.example_fastq(d)
        fa = T.tricky_fasta(d)
        second = str(d / "second.fq.gz")
        with gz.open(second, "wt") as f:
            for i, s in enumerate(seqs[:100]):
                f.write("@gz_%d\n%s\n+\n%s\n" % (i, s, "#" * len(s)))
        for sh in ("1", "0"):
            T.run(T.REF, "createdb", fq, fa, second, str(d / ("ref" + sh)), "--shuffle", sh, "-v", "0")
            out["shuffle" + sh] = {ext: T.digest(str(d / ("ref" + sh)) + ext) for ext in T.DB_FILES}
        T.run(T.REF, "createdb", fq, str(d / "db"), "--shuffle", "1", "-v", "0")
        T.run(T.REF, "convert2fasta", str(d / "db"), str(d / "mine.fasta"), "-v", "0")
        mmdb.write_seqdb(str(d / "asm"), seqs[:50])
        mmdb.write_db(str(d / "cyc"), [(3, b"x\n"), (17, b"y\n")], mmdb.DBTYPE_NUCLEOTIDES)
        T.run(T.REF, "createhdb", str(d / "asm"), str(d / "asm"), "-v", "0")
        T.run(T.REF, "convert2fasta", str(d / "asm"), str(d / "asm.fasta"), "-v", "0")
        fasta = {"mine.fasta": T.digest(str(d / "mine.fasta")), "asm.fasta": T.digest(str(d / "asm.fasta"))}
        T.run(T.REF, "createhdb", str(d / "asm"), str(d / "cyc"), str(d / "asm"), "-v", "0")
        T.run(T.REF, "convert2fasta", str(d / "asm"), str(d / "asm_cyc.fasta"), "-v", "0")
        fasta.update({"asm_cyc.fasta": T.digest(str(d / "asm_cyc.fasta")), "asm_h": T.digest(str(d / "asm_h")), "asm_h.index": T.digest(str(d / "asm_h.index"))})
        out["fasta"] = fasta
        odd = {}
        for p in T.odd_inputs(d):
            name = os.path.basename(p)
            T.run(T.REF, "createdb", p, str(d / ("r_" + name)), "--shuffle", "0", "--dbtype", "2", "-v", "0")
            odd[name] = {ext: T.digest(str(d / ("r_" + name)) + ext) for ext in T.DB_FILES}
        out["odd"] = odd
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "example", "createdb_digests.json"), "w"), indent=1)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "createdb":
    createdb_digests()
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "contigs":
    contig_goldens()


def cycle_goldens(threads=4):
    """tests/golden/cycle/: the reference's cyclecheck on the contig set of tests/cyclecases.py (whole / chopped / with a
    --max-seq-len that skips contigs), and tests/golden/circ/: reads from circular genomes through 3 read iterations and 4 contig
    iterations of the workflow loop INCLUDING the script's cyclecheck() step (data/nuclassemble.sh:19-60: circular contigs are cut,
    set aside and concatenated to the result at the end).  python tests/golden/make_golden.py cycle"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cyclecases
    from stageflags import AC_FLAGS, KC_FLAGS
    d = os.path.join(OUT, "cycle")
    os.makedirs(d, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        t = lambda s: os.path.join(tmp, s)
        mmdb.write_seqdb(t("in"), cyclecases.cases())
        gz_write(os.path.join(d, "in.keyed.gz"), mmdb.dump_keyed(t("in")))
        for name, flags in (("whole", ["--chop-cycle", "0", "--max-seq-len", "300000"]), ("chop", ["--chop-cycle", "1", "--max-seq-len", "300000"]),
                            ("chop_max3000", ["--chop-cycle", "1", "--max-seq-len", "3000"]), ("defaults", [])):
            run("cyclecheck", t("in"), t(name), *flags, "--threads", str(threads))
            gz_write(os.path.join(d, name + ".keyed.gz"), mmdb.dump_keyed(t(name)))
        # the loop
        d = os.path.join(OUT, "circ")
        os.makedirs(d, exist_ok=True)
        prefix = t("dhigh")
        synth.write_dhigh_profiles(prefix)
        cur = t("reads")
        mmdb.write_seqdb(cur, cyclecases.circular_reads())
        gz_write(os.path.join(d, "reads.keyed.gz"), mmdb.dump_keyed(cur))
        dmg = ["--ancient-damage", prefix, "--threads", str(threads)]
        cyc_all = {}
        for it in range(7):
            p = lambda s: t("%s_%d" % (s, it))
            contigs = it >= 3
            run("kmermatcher", cur, p("pref"), *(KC_FLAGS if contigs else K_FLAGS), "--threads", "1")
            run("rescorediagonal", cur, cur, p("pref"), p("aln"), *R_FLAGS, "--threads", str(threads))
            run("ancient_correction", cur, p("aln"), p("corr"), *(AC_FLAGS if contigs else A_FLAGS), *dmg)
            if not contigs:
                run("ancient_read_assemble", p("corr"), p("aln"), p("asm"), *A_FLAGS, *dmg)
                cur = p("asm")
                continue
            run("ancient_contig_merge", p("corr"), p("aln"), p("asm"), *AC_FLAGS, *dmg)
            run("cyclecheck", p("asm"), p("cyc"), "--chop-cycle", "1", "--max-seq-len", "200000", "--threads", str(threads))
            gz_write(os.path.join(d, "cyc_%d.keyed.gz" % it), mmdb.dump_keyed(p("cyc")))
            cyc = mmdb.read_db(p("cyc"))
            cyc_all.update(cyc)
            rest = {k: v for k, v in mmdb.read_db(p("asm")).items() if k not in cyc}      # the "_noneCycle" index of the script
            mmdb.write_from_keyed(p("rest"), rest, mmdb.DBTYPE_NUCLEOTIDES)
            cur = p("rest")
        final = dict(mmdb.read_db(cur))
        final.update(cyc_all)                                                           # concatdbs --preserve-keys
        mmdb.write_from_keyed(t("final"), final, mmdb.DBTYPE_NUCLEOTIDES)
        gz_write(os.path.join(d, "final.keyed.gz"), mmdb.dump_keyed(t("final")))
        print("circ: %d circular contigs set aside, %d entries in the result" % (len(cyc_all), len(final)))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "cycle":
    cycle_goldens()


def letter_reads(n=1500, seed=5):
    """synthetic reads carrying what real FASTA carries beside ACGTN: soft-masked (lower case) stretches, whole lower-case reads,
    IUPAC ambiguity codes in both cases, U, X, and a few bytes that are no nucleotide letters at all"""
    rng = np.random.RandomState(seed)
    seqs = synth.generate_strings(n, seed=seed, mixed=(60, 150))
    iupac = "RYSWKMBDHVUNXryswkmbdhvunx"
    out = []
    for s in seqs:
        b = bytearray(s.encode())
        r = rng.rand()
        if r < 0.10:                                     # a soft-masked stretch
            a = rng.randint(0, len(b) - 10); e = a + rng.randint(5, min(60, len(b) - a))
            b[a:e] = bytes(b[a:e]).lower()
        elif r < 0.14:                                   # an entirely lower-case read
            b = bytearray(bytes(b).lower())
        elif r < 0.30:                                   # one to three ambiguity codes
            for _ in range(rng.randint(1, 4)):
                b[rng.randint(0, len(b))] = ord(iupac[rng.randint(0, len(iupac))])
        elif r < 0.32:                                   # bytes that are not letters
            b[rng.randint(0, len(b))] = ord("*-.1"[rng.randint(0, 4)])
        out.append(b.decode())
    return out


def letters_goldens():
    """tests/golden/letters/: the reads loop of the reference's own object code on reads with lower-case and IUPAC letters
    (python tests/golden/make_golden.py letters)"""
    with tempfile.TemporaryDirectory() as tmp:
        prefix = os.path.join(tmp, "dhigh")
        synth.write_dhigh_profiles(prefix)
        chain(tmp, "letters", letter_reads(), 3, prefix, 4)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "letters":
    letters_goldens()


def workflow_golden(threads=8):
    """tests/golden/example/{test_data.fq.gz, ancient_assemble.fasta}: BASELINE config 1 - the reference's whole program
    (oracle/_ref/carpedeam_full: its own main, command tables, workflow drivers and scripts) run as
    `ancient_assemble example/test_data.fq.gz out.fa tmp --ancient-damage example/dhigh`; the reads file is the reference's example
    data set, kept next to the result so the test runs where /root/reference is absent (python tests/golden/make_golden.py workflow)"""
    import shutil
    full = os.path.join(ROOT, "oracle", "_ref", "carpedeam_full")
    ex = "/root/reference/example"
    d = os.path.join(OUT, "example")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "out.fa")
        r = subprocess.run([full, "ancient_assemble", os.path.join(ex, "test_data.fq.gz"), out, os.path.join(tmp, "tmp"),
                            "--ancient-damage", os.path.join(ex, "dhigh"), "--threads", str(threads)], capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit("reference workflow failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])
        shutil.copyfile(out, os.path.join(d, "ancient_assemble.fasta"))
    shutil.copyfile(os.path.join(ex, "test_data.fq.gz"), os.path.join(d, "test_data.fq.gz"))
    print("workflow golden: %d contigs" % open(os.path.join(d, "ancient_assemble.fasta")).read().count(">"))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "workflow":
    workflow_golden()


LINCLUST_K_FLAGS = ("--alph-size nucl:5,aa:13 --min-seq-id 0.97 --kmer-per-seq 200 --spaced-kmer-mode 0 --kmer-per-seq-scale 0.200 --adjust-kmer-len 0 --mask 0 "
                    "--mask-lower-case 0 --cov-mode 1 -k 20 -c 0.99 --max-seq-len 200000 --hash-shift 67 --split-memory-limit 0 --include-only-extendable 0 "
                    "--ignore-multi-kmer 1").split()
HAMMING_FLAGS = ("--rescore-mode 0 --wrapped-scoring 1 --filter-hits 0 -e 0.001 -c 0.99 -a 0 --cov-mode 1 --min-seq-id 0.97 --min-aln-len 0 --seq-id-mode 0 "
                 "--add-self-matches 0 --sort-results 0").split()


def hamming_contigs(seed=31):
    """A contig set for linclust's pre-clustering: random contigs of 150..4000 letters and, of some of them, exact copies, rotations
    (a circular contig cut elsewhere), reverse complements, rotated reverse complements, copies with a few substitutions, shorter
    pieces, copies with N / lower-case stretches / an IUPAC code; two rotated copies of a 70 000-letter contig (diagonals beyond 16 bit)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    L = np.frombuffer(b"ACGT", np.uint8)
    rc = lambda s: s[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))
    seqs = []
    for i in range(260):
        n = int(rng.integers(150, 4000)) if i else 70000
        s = L[rng.integers(0, 4, n)].tobytes().decode()
        seqs.append(s)
        for _ in range(int(rng.integers(0, 4)) if i else 2):
            r = rng.random()
            t = s
            if r < 0.2:
                pass
            elif r < 0.45:
                k = int(rng.integers(1, n)); t = s[k:] + s[:k]
            elif r < 0.6:
                t = rc(s)
            elif r < 0.75:
                k = int(rng.integers(1, n)); t = rc(s[k:] + s[:k])
            elif r < 0.85:
                b = bytearray(s.encode())
                for _m in range(int(rng.integers(1, max(2, n // 40)))):
                    k = int(rng.integers(0, n)); b[k] = L[(int(np.searchsorted(L, b[k])) + 1) % 4]
                t = b.decode()
            elif r < 0.93:
                a = int(rng.integers(0, n // 50 + 1)); t = s[a: n - int(rng.integers(0, n // 50 + 1))]
            else:
                b = bytearray(s.encode())
                k = int(rng.integers(0, n)); b[k] = ord("N")
                a = int(rng.integers(0, n - 20)); b[a: a + 20] = bytes(b[a: a + 20]).lower()
                b[int(rng.integers(0, n))] = ord("R")
                t = b.decode()
            if rng.random() < 0.3 and r >= 0.2:
                k = int(rng.integers(1, len(t))); t = t[k:] + t[:k]
            seqs.append(t)
    order = rng.permutation(len(seqs))
    return [seqs[i] for i in order]


def hamming_goldens(threads=4):
    """tests/golden/hamming/: the reference's rescorediagonal in linclust's pre-clustering mode (--rescore-mode 0 --wrapped-scoring 1)
    on hamming_contigs() and on kmermatcher's hits with linclust's flags.  python tests/golden/make_golden.py hamming"""
    d = os.path.join(OUT, "hamming")
    os.makedirs(d, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        t = lambda s: os.path.join(tmp, s)
        mmdb.write_seqdb(t("in"), hamming_contigs())
        run("kmermatcher", t("in"), t("pref"), *LINCLUST_K_FLAGS, "--threads", "1")
        run("rescorediagonal", t("in"), t("in"), t("pref"), t("res"), *HAMMING_FLAGS, "--threads", str(threads))
        for s in ("in", "pref", "res"):
            gz_write(os.path.join(d, s + ".keyed.gz"), mmdb.dump_keyed(t(s)))
        res = mmdb.read_db(t("res"))
        lines = sum(v[0].count(b"\n") for v in res.values())
        print("hamming: %d contigs, %d prefilter entries, %d records kept" % (len(mmdb.read_db(t("in"))), len(mmdb.read_db(t("pref"))), lines))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "hamming":
    hamming_goldens()
