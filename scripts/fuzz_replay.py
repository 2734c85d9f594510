"""Replay of a read set a fuzz campaign saved (gpurun_out/fuzz_fail_*.txt, one read per line) through the device chain and the oracle,
three iterations, with the first differences per stage:  python scripts/fuzz_replay.py <reads.txt> [k22]
The environment's CDM_* switches apply (e.g. CDM_EXTEND=queries, CDM_KMER_VOTE=tuples) - that is how a difference is narrowed down."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from carpedeam_amd import capi, mmdb, synth
from gpuutil import diff_keys, run_oracle, seqdb_to_keyed
from stageflags import A_FLAGS, K_FLAGS, R_FLAGS

seqs = [l.rstrip("\n") for l in open(sys.argv[1]) if l.strip()]
d = tempfile.mkdtemp()
t = lambda s: os.path.join(d, s)
synth.write_dhigh_profiles(t("dhigh"))
oracle = os.path.join(ROOT, "oracle", "_build", "cdm_oracle")
ctx = capi.Ctx(0)
ctx.damage_load(t("dhigh"))
kflags, kpar = K_FLAGS, None
if len(sys.argv) > 2 and sys.argv[2] == "k22":
    kflags = " ".join(K_FLAGS).replace("-k 20", "-k 22").replace("--include-only-extendable 0", "--include-only-extendable 1").split()
    kpar = capi.KmerParams(22, 200, 0.2, 67, 1, 1, 1, 0.0)
mmdb.write_seqdb(t("in0"), seqs)
db = ctx.upload_seqs(seqs)
print({k: v for k, v in os.environ.items() if k.startswith("CDM_")}, len(seqs), "reads")
for it in range(3):
    hits = ctx.kmermatch(db, kpar); alns = ctx.rescore(db, hits); corr = ctx.correct(db, alns); asm = ctx.extend(corr, alns)
    i, o = t("in%d" % it), t("in%d" % (it + 1))
    run_oracle(oracle, "kmermatcher", i, t("pref"), *kflags, "--threads", "1")
    run_oracle(oracle, "rescorediagonal", i, i, t("pref"), t("aln"), *R_FLAGS, "--threads", "4")
    run_oracle(oracle, "ancient_correction", i, t("aln"), t("corr"), *A_FLAGS, "--ancient-damage", t("dhigh"), "--threads", "4")
    run_oracle(oracle, "ancient_read_assemble", t("corr"), t("aln"), o, *A_FLAGS, "--ancient-damage", t("dhigh"), "--threads", "4")
    lens, keys, _ = db.meta()
    hoff, hrec = hits.download(); aoff, arec = alns.download()
    bad = [("pref", diff_keys({k: (v, 0) for k, v in capi.hits_to_text(hoff, hrec, keys).items()}, {k: (v[0], 0) for k, v in mmdb.read_db(t("pref")).items()})),
           ("aln", diff_keys({k: (v, 0) for k, v in capi.alns_to_text(aoff, arec, keys, lens, db.residues).items()}, mmdb.read_db(t("aln")))),
           ("corr", diff_keys(seqdb_to_keyed(*corr.download()), mmdb.read_db(t("corr")))),
           ("asm", diff_keys(seqdb_to_keyed(*asm.download()), mmdb.read_db(o)))]
    for n, b in bad:
        print("iteration", it, n, "ok" if not b else "DIFFERS: " + str(b)[:1500])
    if any(b for _, b in bad):
        got, exp = seqdb_to_keyed(*asm.download()), mmdb.read_db(o)
        for k in sorted(exp):
            if got.get(k) != exp[k]:
                print("  key", k, "\n   device", got.get(k), "\n   oracle", exp[k])
        break
    db = asm
