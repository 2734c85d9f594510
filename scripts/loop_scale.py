"""The fused workflow loop at scale: writes <reads> mixed-length synthetic reads (60-150 bp, 20x) as a sequence DB and runs
`carpedeam ancient_reads_loop --num-iter-reads-only 5 --num-iterations 12` on it: python scripts/loop_scale.py <reads> [threads]"""
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carpedeam_amd import build, capi, mmdb, synth  # noqa: E402

n = int(sys.argv[1])
threads = sys.argv[2] if len(sys.argv) > 2 else "16"
exe = os.path.join(os.path.dirname(build.build()), "carpedeam")
with tempfile.TemporaryDirectory() as d:
    synth.write_dhigh_profiles(d + "/dhigh")
    t0 = time.time()
    ctx = capi.Ctx(0)
    db = ctx.synth(n, 60, 150, 1)
    seqs, keys, _ = db.download()
    del db, ctx
    mmdb.write_seqdb(d + "/reads", seqs)
    del seqs
    print("reads written in %.1f s" % (time.time() - t0), flush=True)
    t0 = time.time()
    os.environ["CDM_TIMING"] = "1"
    r = subprocess.run([exe, "ancient_reads_loop", d + "/reads", d + "/out", "--ancient-damage", d + "/dhigh", "--num-iter-reads-only", "5", "--num-iterations", "12",
                        "--threads", threads], capture_output=True, text=True)
    print(r.stderr[-9000:])
    print("exit %d, %.1f s" % (r.returncode, time.time() - t0))
    if r.returncode == 0:
        lens = sorted((int(l.split()[2]) - 2 for l in open(d + "/out.index")), reverse=True)
        print("result: %d sequences, longest %s, N50-ish %d" % (len(lens), lens[:5], lens[len(lens) // 2]))
    sys.exit(r.returncode)
