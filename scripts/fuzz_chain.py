"""Exploratory fuzzing of the device chain against the oracle (test infrastructure; run on a GPU box):
    python scripts/fuzz_chain.py <mode> <cases> <seed>
modes: small, deep (100x+ pile-ups, > 64 records per query), repeats (low-complexity / tandem repeats), contigparams (k = 22,
include-only-extendable), longreads (up to 600 bp: general extraction kernel), nrich (N letters), verylong (wide tuple layout), tiling (chains of
extensions), tiny (reads around and below k), palrepeats (tandem repeats of reverse-palindromic
units: comparator ties in the per-sequence k-mer sort), uniform (one read length per case: the slot layout of sort 1), letters (lower-case stretches, IUPAC codes, bytes that are no letters;
FUZZ_LETTERS=1 in the environment puts them on top of any other mode)."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from carpedeam_amd import capi, mmdb, synth
from gpuutil import OracleCrash, diff_keys, run_oracle, seqdb_to_keyed
from stageflags import A_FLAGS, K_FLAGS, R_FLAGS
from test_oracle_golden import pref_sign_ties

mode, cases, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed)
letters = np.frombuffer(b"ACGT", np.uint8)
d = tempfile.mkdtemp()
t = lambda s: os.path.join(d, s)
synth.write_dhigh_profiles(t("dhigh"))
oracle = os.path.join(ROOT, "oracle", "_build", "cdm_oracle")
ctx = capi.Ctx(0)
ctx.damage_load(t("dhigh"))
kflags, kpar = K_FLAGS, None
if mode == "contigparams":
    kflags = " ".join(K_FLAGS).replace("-k 20", "-k 22").replace("--include-only-extendable 0", "--include-only-extendable 1").split()
    kpar = capi.KmerParams(22, 200, 0.2, 67, 1, 1, 1, 0.0)
fails = 0
ties = 0
unsupported = 0
undefined = 0
for case in range(cases):
    G = {"small": 300, "deep": 150, "repeats": 200, "contigparams": 400, "longreads": 1500, "palrepeats": 700, "nrich": 300, "verylong": 6000, "tiling": 600, "tiny": 60, "letters": 300, "uniform": 300}[mode]
    genome = rng.integers(0, 4, G)
    if mode == "repeats" or (mode == "uniform" and case % 3 == 0):
        unit = rng.integers(0, 4, int(rng.integers(1, 9)))
        a = int(rng.integers(0, G - 80)); genome[a:a + 80] = np.resize(unit, 80)
    if mode == "palrepeats":
        units = ["GTACGC", "GTAC", "ACGT", "GATC", "CATG", "GCGC", "AT", "TGCA", "AGCT", "GTACGCGTAC", "ACGTTGCAACGT"]
        for _ in range(2):
            u = np.array(["ACGT".index(ch) for ch in units[int(rng.integers(0, len(units)))]])
            ln = int(rng.integers(60, 300)); a = int(rng.integers(0, G - ln)); genome[a:a + ln] = np.resize(u, ln)
    nreads = {"small": (6, 61), "uniform": (6, 90), "deep": (150, 400), "repeats": (10, 80), "contigparams": (10, 80), "longreads": (10, 60), "palrepeats": (8, 50), "nrich": (10, 80), "verylong": (6, 30), "tiling": (30, 120), "tiny": (2, 40), "letters": (8, 70)}[mode]
    lr = {"small": (30, 121), "uniform": (22, 121), "deep": (40, 101), "repeats": (30, 121), "contigparams": (40, 200), "longreads": (100, 600), "palrepeats": (30, 320), "nrich": (30, 150), "verylong": (600, 3000), "tiling": (40, 90), "tiny": (5, 45), "letters": (30, 140)}[mode]
    seqs = []
    fixedL = int(rng.integers(*lr)) if mode == "uniform" else 0        # uniform: one read length per case (the 8-byte slot layout of sort 1)
    for _ in range(int(rng.integers(*nreads))):
        L = fixedL or int(rng.integers(*lr)); L = min(L, G - 1); st = int(rng.integers(0, G - L))
        c = genome[st:st + L].copy()
        if rng.random() < 0.5:
            c = (3 - c)[::-1]
        for j in range(3):
            if c[j] == 1 and rng.random() < 0.3:
                c[j] = 3
            if c[L - 1 - j] == 2 and rng.random() < 0.3:
                c[L - 1 - j] = 0
        if rng.random() < 0.2:
            k = int(rng.integers(0, L)); c[k] = (c[k] + 1) % 4          # a sequencing error
        sq = letters[c].tobytes().decode()
        nprob = 0.6 if mode == "nrich" else 0.05
        if rng.random() < nprob:
            for _n in range(int(rng.integers(1, 6)) if mode == "nrich" else 1):
                k = int(rng.integers(0, L)); sq = sq[:k] + "N" + sq[k + 1:]
        if mode == "letters" or os.environ.get("FUZZ_LETTERS"):      # FUZZ_LETTERS=1: the letters of that mode on top of any other mode
            r = rng.random()
            b = bytearray(sq.encode())
            if r < 0.25:
                a0 = int(rng.integers(0, L)); a1 = min(L, a0 + int(rng.integers(1, 50)))
                b[a0:a1] = bytes(b[a0:a1]).lower()
            elif r < 0.35:
                b = bytearray(bytes(b).lower())
            elif r < 0.65:
                for _n in range(int(rng.integers(1, 5))):
                    odd = b"RYSWKMBDHVUNXryswkmbdhvunx*-.1acgt"
                    b[int(rng.integers(0, L))] = odd[int(rng.integers(0, len(odd)))]
            sq = b.decode()
        seqs.append(sq)
    if rng.random() < 0.4:
        seqs.append(seqs[int(rng.integers(0, len(seqs)))])
    mmdb.write_seqdb(t("in0"), seqs)
    db = ctx.upload_seqs(seqs)
    try:
        for it in range(3):
            hits = ctx.kmermatch(db, kpar); alns = ctx.rescore(db, hits); corr = ctx.correct(db, alns); asm = ctx.extend(corr, alns)
            i, o = t("in%d" % it), t("in%d" % (it + 1))
            run_oracle(oracle, "kmermatcher", i, t("pref"), *kflags, "--threads", "1")     # (a parallel sort breaks comparator ties differently)
            run_oracle(oracle, "rescorediagonal", i, i, t("pref"), t("aln"), *R_FLAGS, "--threads", "4")
            run_oracle(oracle, "ancient_correction", i, t("aln"), t("corr"), *A_FLAGS, "--ancient-damage", t("dhigh"), "--threads", "4")
            run_oracle(oracle, "ancient_read_assemble", t("corr"), t("aln"), o, *A_FLAGS, "--ancient-damage", t("dhigh"), "--threads", "4")
            lens, keys, _ = db.meta()
            hoff, hrec = hits.download(); aoff, arec = alns.download()
            # strand ties of the reference's global sort (reference nondeterminism): the reference's own answer depends
            # on its thread count there; the case cannot be followed further
            gp = mmdb.canon({k: (v, 0) for k, v in capi.hits_to_text(hoff, hrec, keys).items()}); ep = mmdb.canon({k: (v[0], 0) for k, v in mmdb.read_db(t("pref")).items()})
            if gp != ep:
                tie_list, other = pref_sign_ties(gp, ep)
                if not other:
                    ties += 1
                    break
            bad = [("pref", diff_keys({k: (v, 0) for k, v in capi.hits_to_text(hoff, hrec, keys).items()}, {k: (v[0], 0) for k, v in mmdb.read_db(t("pref")).items()})),
                   ("aln", diff_keys({k: (v, 0) for k, v in capi.alns_to_text(aoff, arec, keys, lens, db.residues).items()}, mmdb.read_db(t("aln")))),
                   ("corr", diff_keys(seqdb_to_keyed(*corr.download()), mmdb.read_db(t("corr")))),
                   ("asm", diff_keys(seqdb_to_keyed(*asm.download()), mmdb.read_db(o)))]
            bad = [(n, b) for n, b in bad if b]
            if bad:
                fails += 1
                print("FAIL", mode, "case", case, "iter", it, [(n, str(b)[:300]) for n, b in bad], flush=True)
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                fn = os.path.join(ROOT, "gpurun_out", "fuzz_fail_%s_%d_%d.txt" % (mode, seed, case))
                open(fn, "w").write("\n".join(seqs) + "\n")
                print("  reads saved:", fn, len(seqs), flush=True)
                break
            db = asm
    except OracleCrash as e:
        # e.g. a sequence whose self alignment scores 0 on every diagonal ("-*CGT"): the reference writes coordinates -1 into its
        # identity record and its own ancient_correction indexes the sequence with them; nothing to compare with
        undefined += 1
    except capi.CdmError as e:
        if "not implemented" in str(e):
            unsupported += 1                    # a documented limit of the device path (refused, never computed differently)
        else:
            print("ERROR", mode, "case", case, str(e)[:300], flush=True); fails += 1
print("mode", mode, "cases", cases, "failures", fails, "sign-tie cases skipped", ties, "refused as unsupported", unsupported, "undefined in the reference (it crashes)", undefined, flush=True)
