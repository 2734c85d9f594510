import os, sys, time, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from carpedeam_amd import capi, synth
n = int(sys.argv[1])
ctx = capi.Ctx(0)
with tempfile.TemporaryDirectory() as d:
    synth.write_dhigh_profiles(os.path.join(d, "dhigh")); ctx.damage_load(os.path.join(d, "dhigh"))
db = ctx.synth(n, 100, 100, 1)
comm = capi.Comm.rccl(ctx, 0, 1, capi.Comm.unique_id())
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, 1e3 * (time.perf_counter() - t0)
for it in range(4):
    h, t1 = timed(lambda: comm.kmermatch(db))
    a, t2 = timed(lambda: ctx.rescore(db, h))
    c, t3 = timed(lambda: ctx.correct(db, a))
    cg, t4 = timed(lambda: comm.allgather_owned(c))
    e, t5 = timed(lambda: ctx.extend(cg, a))
    eg, t6 = timed(lambda: comm.allgather_owned(e))
    print("kmermatch_dist %.1f  rescore %.1f  correct %.1f  allgather %.1f  extend %.1f  allgather %.1f  = %.1f ms" % (t1, t2, t3, t4, t5, t6, t1 + t2 + t3 + t4 + t5 + t6), flush=True)
    del h, a, c, cg, e, eg
