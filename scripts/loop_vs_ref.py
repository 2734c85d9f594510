"""The whole workflow loop (5 read iterations + 7 contig iterations with cyclecheck, data/nuclassemble.sh:96-232) on <reads>
mixed-length synthetic reads, twice on the same DB files: `carpedeam ancient_reads_loop` (MI355X, one process) and the reference's
own object code (oracle/_ref/carpedeam_ref, module by module as the script chains them, <threads> threads).  Prints both wall
times and how many result sequences differ (the reference's run-dependent strand ties, DESIGN.md N1, can reach a few: with
`twice` the reference chain runs a second time and its two results are compared with each other as well).
`oracle` chains oracle/_build/cdm_oracle (deterministic tie rule: zero differences expected) instead of the reference's object code;
`contigid=X` / `seqid=X` run both sides with --min-seqid-corr-contigs X / --min-seq-id X (the workflow turns the former into the
contig phase's --min-seq-id, Nuclassembler.cpp:124-126); `len=LO-HI` sets the read lengths.
Test infrastructure; run on a GPU box:  python scripts/loop_vs_ref.py <reads> [threads] [twice] [oracle] [contigid=X] [seqid=X] [len=LO-HI]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from carpedeam_amd import build, capi, mmdb, synth  # noqa: E402
from stageflags import A_FLAGS, AC_FLAGS, K_FLAGS, KC_FLAGS, R_FLAGS  # noqa: E402

n = int(sys.argv[1])
threads = sys.argv[2] if len(sys.argv) > 2 else "16"
exe = os.path.join(os.path.dirname(build.build()), "carpedeam")
REF = os.path.join(ROOT, "oracle", "_ref", "carpedeam_ref")
opts = dict(a.split("=") for a in sys.argv[3:] if "=" in a)
if "oracle" in sys.argv:
    REF = os.path.join(ROOT, "oracle", "_build", "cdm_oracle")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
seqid, contigid = opts.get("seqid", "0.9"), opts.get("contigid", "0.9")
lo, hi = (int(x) for x in opts.get("len", "60-150").split("-"))


def with_seqid(flags, v):
    flags = list(flags)
    flags[flags.index("--min-seq-id") + 1] = v
    return flags


R_READS, A_READS, R_CONTIGS, A_CONTIGS = with_seqid(R_FLAGS, seqid), with_seqid(A_FLAGS, seqid), with_seqid(R_FLAGS, contigid), with_seqid(AC_FLAGS, contigid)
d = tempfile.mkdtemp()
t = lambda s: os.path.join(d, s)


def run(*a):
    r = subprocess.run(list(a), capture_output=True, text=True)
    if r.returncode:
        sys.exit("failed: %s\n%s" % (" ".join(a[:3]), r.stderr[-1500:]))
    return r


def ref_chain(tag):
    """the reference's modules chained as data/nuclassemble.sh chains them -> (result DB, seconds, seconds per iteration, circular contigs)"""
    dmg = ["--ancient-damage", t("dhigh"), "--threads", threads]
    t0 = time.time()
    cur, cyc_all, laps = t("reads"), {}, []
    for it in range(12):
        ti = time.time()
        p = lambda s: t("%s%s_%d" % (tag, s, it))
        contigs = it >= 5
        run(REF, "kmermatcher", cur, p("pref"), *(KC_FLAGS if contigs else K_FLAGS), "--threads", threads)
        run(REF, "rescorediagonal", cur, cur, p("pref"), p("aln"), *(R_CONTIGS if contigs else R_READS), "--threads", threads)
        run(REF, "ancient_correction", cur, p("aln"), p("corr"), *(A_CONTIGS if contigs else A_READS), *dmg)
        if not contigs:
            run(REF, "ancient_read_assemble", p("corr"), p("aln"), p("asm"), *A_READS, *dmg)
            cur = p("asm")
        else:
            run(REF, "ancient_contig_merge", p("corr"), p("aln"), p("asm"), *A_CONTIGS, *dmg)
            run(REF, "cyclecheck", p("asm"), p("cyc"), "--chop-cycle", "1", "--max-seq-len", "200000", "--threads", threads)
            cyc = mmdb.read_db(p("cyc"))
            if cyc:     # the script's awk step: the circular contigs are set aside (_noneCycle index), concatenated to the result at the end
                cyc_all.update(cyc)
                mmdb.write_from_keyed(p("rest"), {k: v for k, v in mmdb.read_db(p("asm")).items() if k not in cyc}, mmdb.DBTYPE_NUCLEOTIDES)
                cur = p("rest")
            else:
                cur = p("asm")
        laps.append(round(time.time() - ti, 2))
    secs = time.time() - t0
    want = dict(mmdb.read_db(cur))
    want.update(cyc_all)
    for f in os.listdir(d):
        if f.startswith(tag):
            os.remove(t(f))
    return want, secs, laps, cyc_all


def differ(a, b):
    return sum(1 for k in set(a) | set(b) if a.get(k) != b.get(k))


synth.write_dhigh_profiles(t("dhigh"))
ctx = capi.Ctx(0)
seqs, _, _ = ctx.synth(n, lo, hi, 1).download()
del ctx
mmdb.write_seqdb(t("reads"), seqs)
del seqs
t0 = time.time()
r = run(exe, "ancient_reads_loop", t("reads"), t("out"), "--ancient-damage", t("dhigh"), "--num-iter-reads-only", "5", "--num-iterations", "12", "--threads", threads,
        "--min-seq-id", seqid, "--min-seqid-corr-contigs", contigid)
t_gpu = time.time() - t0
print(r.stderr[-1600:])
got = mmdb.read_db(t("out"))
want, t_ref, laps, cyc_all = ref_chain("a_")
print(("reads %d threads %s: MI355X loop %.1f s, " + ("oracle" if "oracle" in sys.argv else "reference") + " modules %.1f s (per iteration %s) -> %.1fx; result %d sequences, %d residues, %d circular set aside; %d sequences differ")
      % (n, threads, t_gpu, t_ref, laps, t_ref / t_gpu, len(got), sum(len(v[0]) - 1 for v in got.values()), len(cyc_all), differ(got, want)), flush=True)
if "twice" in sys.argv:
    again = ref_chain("b_")[0]
    print("the reference against its own second run: %d sequences differ; MI355X against the second run: %d" % (differ(want, again), differ(got, again)))
subprocess.run(["rm", "-rf", d])
