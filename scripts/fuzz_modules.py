"""Fuzz the drop-in surface: the `carpedeam` host binary, module by module on DB files, against the oracle's DB files (and the
ancient_reads_loop against the stage-by-stage chain): two iterations of the reads loop, then two of the contig phase (kmermatcher
-k 22, rescorediagonal, ancient_correction, ancient_contig_merge, cyclecheck).  FUZZ_LETTERS=1 adds lower-case stretches, IUPAC
codes and non-letters to the reads.  Test infrastructure; run on a GPU box:
    python scripts/fuzz_modules.py <cases> <seed>"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from carpedeam_amd import mmdb, synth
from gpuutil import diff_keys
from stageflags import A_FLAGS, AC_FLAGS, K_FLAGS, KC_FLAGS, R_FLAGS
from test_oracle_golden import pref_sign_ties

cases, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
letters = np.frombuffer(b"ACGT", np.uint8)
d = tempfile.mkdtemp()
t = lambda s: os.path.join(d, s)
synth.write_dhigh_profiles(t("dhigh"))
ORACLE = os.path.join(ROOT, "oracle", "_build", "cdm_oracle")
BIN = os.path.join(ROOT, "carpedeam_amd", "carpedeam")


def run(exe, *a):
    r = subprocess.run([exe] + list(a), capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("exit code %d: " % r.returncode + exe + " " + " ".join(a) + "\n" + r.stderr[-600:])


fails = 0
for case in range(cases):
    G = int(rng.integers(150, 900))
    genome = rng.integers(0, 4, G)
    seqs = []
    for _ in range(int(rng.integers(5, 120))):
        L = min(int(rng.integers(25, 260)), G - 1); st = int(rng.integers(0, G - L))
        c = genome[st:st + L].copy()
        if rng.random() < 0.5:
            c = (3 - c)[::-1]
        for j in range(3):
            if c[j] == 1 and rng.random() < 0.3:
                c[j] = 3
            if c[L - 1 - j] == 2 and rng.random() < 0.3:
                c[L - 1 - j] = 0
        sq = letters[c].tobytes().decode()
        if rng.random() < 0.1:
            k = int(rng.integers(0, L)); sq = sq[:k] + "N" + sq[k + 1:]
        if os.environ.get("FUZZ_LETTERS") and rng.random() < 0.5:
            b = bytearray(sq.encode())
            if rng.random() < 0.5:
                a0 = int(rng.integers(0, L)); a1 = min(L, a0 + int(rng.integers(1, 60)))
                b[a0:a1] = bytes(b[a0:a1]).lower()
            else:
                odd = b"RYSWKMBDHVUNXryswkmbdhvunx*-.1acgt"
                for _n in range(int(rng.integers(1, 4))):
                    b[int(rng.integers(0, L))] = odd[int(rng.integers(0, len(odd)))]
            sq = b.decode()
        seqs.append(sq)
    # keys need not be 0..n-1: the modules look sequences up by key
    keys = sorted(rng.choice(np.arange(0, 4 * len(seqs)), len(seqs), replace=False).tolist()) if rng.random() < 0.5 else list(range(len(seqs)))
    mmdb.write_db(t("in0"), [(k, (s + "\n").encode()) for k, s in zip(keys, seqs)], mmdb.DBTYPE_NUCLEOTIDES)
    try:
        for it in range(2):
            i, o = t("in%d" % it), t("in%d" % (it + 1))
            th = ["--threads", "1"]
            run(ORACLE, "kmermatcher", i, t("prefO"), *K_FLAGS, *th); run(BIN, "kmermatcher", i, t("pref"), *K_FLAGS, *th)
            gp, ep = mmdb.canon(mmdb.read_db(t("pref"))), mmdb.canon(mmdb.read_db(t("prefO")))
            bad = []
            if gp != ep:
                ties, other = pref_sign_ties(gp, ep)
                if other:
                    bad.append(("pref", str(other)[:200]))
            run(ORACLE, "rescorediagonal", i, i, t("prefO"), t("alnO"), *R_FLAGS, *th); run(BIN, "rescorediagonal", i, i, t("prefO"), t("aln"), *R_FLAGS, *th)
            if diff_keys(mmdb.read_db(t("aln")), mmdb.read_db(t("alnO"))):
                bad.append(("aln", str(diff_keys(mmdb.read_db(t("aln")), mmdb.read_db(t("alnO"))))[:200]))
            run(ORACLE, "ancient_correction", i, t("alnO"), t("corrO"), *A_FLAGS, "--ancient-damage", t("dhigh"), *th)
            run(BIN, "ancient_correction", i, t("alnO"), t("corr"), *A_FLAGS, "--ancient-damage", t("dhigh"), *th)
            if diff_keys(mmdb.read_db(t("corr")), mmdb.read_db(t("corrO"))):
                bad.append(("corr", ""))
            run(ORACLE, "ancient_read_assemble", t("corrO"), t("alnO"), o, *A_FLAGS, "--ancient-damage", t("dhigh"), *th)
            run(BIN, "ancient_read_assemble", t("corrO"), t("alnO"), t("asm"), *A_FLAGS, "--ancient-damage", t("dhigh"), *th)
            if diff_keys(mmdb.read_db(t("asm")), mmdb.read_db(o)):
                bad.append(("asm", ""))
            if bad:
                fails += 1
                print("FAIL case", case, "iter", it, bad, flush=True)
                break
        else:
            # the contig phase (data/nuclassemble.sh:148-232), module by module; every module starts from the oracle's upstream DBs
            dmg = ["--ancient-damage", t("dhigh"), "--threads", "1"]
            acf = AC_FLAGS
            if rng.random() < 0.4:      # the contig merge's --unsafe 1 mode (majority-vote consensus), any --min-cov-safe
                acf = " ".join(AC_FLAGS).replace("--unsafe 0", "--unsafe 1").replace("--min-cov-safe 5", "--min-cov-safe %d" % int(rng.integers(1, 6))).split()
            for it in range(2, 4):
                i, o = t("in%d" % it), t("in%d" % (it + 1))
                th = ["--threads", "1"]
                bad = []
                run(ORACLE, "kmermatcher", i, t("prefO"), *KC_FLAGS, *th); run(BIN, "kmermatcher", i, t("pref"), *KC_FLAGS, *th)
                gp, ep = mmdb.canon(mmdb.read_db(t("pref"))), mmdb.canon(mmdb.read_db(t("prefO")))
                if gp != ep and pref_sign_ties(gp, ep)[1]:
                    bad.append(("cpref", str(pref_sign_ties(gp, ep)[1])[:200]))
                run(ORACLE, "rescorediagonal", i, i, t("prefO"), t("alnO"), *R_FLAGS, *th); run(BIN, "rescorediagonal", i, i, t("prefO"), t("aln"), *R_FLAGS, *th)
                if diff_keys(mmdb.read_db(t("aln")), mmdb.read_db(t("alnO"))):
                    bad.append(("caln", str(diff_keys(mmdb.read_db(t("aln")), mmdb.read_db(t("alnO"))))[:200]))
                run(ORACLE, "ancient_correction", i, t("alnO"), t("corrO"), *AC_FLAGS, *dmg); run(BIN, "ancient_correction", i, t("alnO"), t("corr"), *AC_FLAGS, *dmg)
                if diff_keys(mmdb.read_db(t("corr")), mmdb.read_db(t("corrO"))):
                    bad.append(("ccorr", ""))
                run(ORACLE, "ancient_contig_merge", t("corrO"), t("alnO"), o, *acf, *dmg); run(BIN, "ancient_contig_merge", t("corrO"), t("alnO"), t("mrg"), *acf, *dmg)
                if diff_keys(mmdb.read_db(t("mrg")), mmdb.read_db(o)):
                    bad.append(("cmerge", str(diff_keys(mmdb.read_db(t("mrg")), mmdb.read_db(o)))[:200]))
                cyc = ["--chop-cycle", str(int(rng.integers(0, 2))), "--max-seq-len", "200000"]
                run(ORACLE, "cyclecheck", o, t("cycO"), *cyc); run(BIN, "cyclecheck", o, t("cyc"), *cyc)
                if diff_keys(mmdb.read_db(t("cyc")), mmdb.read_db(t("cycO"))):
                    bad.append(("cyclecheck", str(diff_keys(mmdb.read_db(t("cyc")), mmdb.read_db(t("cycO"))))[:200]))
                if bad:
                    fails += 1
                    print("FAIL case", case, "iter", it, bad, flush=True)
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    mmdb.write_db(os.path.join(ROOT, "gpurun_out", "fuzzm_fail_%d_%d" % (seed, case)), [(k, v[0]) for k, v in sorted(mmdb.read_db(i).items())], mmdb.DBTYPE_NUCLEOTIDES)
                    break
    except RuntimeError as e:
        fails += 1
        print("ERROR case", case, str(e)[:500], flush=True)
    for f in os.listdir(d):
        if not f.startswith("dhigh"):
            os.remove(t(f))
    if case % 10 == 9:
        print("  %d cases, %d failures so far" % (case + 1, fails), flush=True)
print("module fuzz: cases", cases, "failures", fails, flush=True)
