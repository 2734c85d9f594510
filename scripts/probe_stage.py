"""Times ancient_correction / ancient_read_assemble alone on a resident alignment set under several environments:
python scripts/probe_stage.py <reads> <env1,k=v;...> ...   ('-' = no extra env).  The environment is read per call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carpedeam_amd import capi, synth  # noqa: E402

n = int(sys.argv[1])
ctx = capi.Ctx(0)
import tempfile
d = tempfile.mkdtemp()
synth.write_dhigh_profiles(d + "/dhigh")
ctx.damage_load(d + "/dhigh")
db = ctx.synth(n, 100, 100, 1)
hits = ctx.kmermatch(db, capi.KmerParams.reads_default())
alns = ctx.rescore(db, hits)
del hits
print("reads %d alignments %d" % (n, alns.count), flush=True)
for spec in sys.argv[2:]:
    keys = []
    if spec != "-":
        for kv in spec.split(";"):
            k, v = kv.split("=")
            os.environ[k] = v
            keys.append(k)
    tc, te = [], []
    for rep in range(3):
        corr = ctx.correct(db, alns)
        tc.append(ctx.last_kernel_ms(0))
        nxt = ctx.extend(corr, alns)
        te.append(ctx.last_kernel_ms(4))
        del corr, nxt
    print("%-40s correct %8.3f ms   extend %8.3f ms" % (spec, min(tc), min(te)), flush=True)
    for k in keys:
        del os.environ[k]
