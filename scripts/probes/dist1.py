import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from carpedeam_amd import capi
n = int(sys.argv[1])
ctx = capi.Ctx(0)
db = ctx.synth(n, 100, 100, 1)
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, time.perf_counter() - t0
h0, t = timed(lambda: ctx.kmermatch(db)); print("single device: %.3f s, %d hits" % (t, h0.count), flush=True)
comm = capi.Comm.rccl(ctx, 0, 1, capi.Comm.unique_id())
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    st0 = capi.pool_stats()
    h1, t = timed(lambda: comm.kmermatch(db)); st1 = capi.pool_stats()
    print("one-rank communicator: %.3f s, %d hits; driver calls %d, %.1f GB, %.2f s" % (t, h1.count, st1["driver_calls"] - st0["driver_calls"], (st1["driver_bytes"] - st0["driver_bytes"]) / 1e9, st1["driver_seconds"] - st0["driver_seconds"]), flush=True)
a, b = h0.download(), h1.download()
print("equal:", np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]))
