import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
from carpedeam_amd import capi, shard
from test_gpu_shards import run_native_ranks, run_ranks, merged_hits
n, lo, hi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ref = capi.Ctx(0)
want = ref.kmermatch(ref.synth(n, lo, hi, 1)).download()
print("single:", len(want[1]))
for W in (1, 2):
    def rank_fn(rank, comm, c):
        h = comm.kmermatch(c.synth(n, lo, hi, 1))
        return h.download(), comm.owned(n)
    t0 = time.time(); res = run_native_ranks(W, rank_fn); dt = time.time() - t0
    off, rec = merged_hits([r[0] for r in res], n, res[0][1])
    print("native W=%d: %.2f s, %d hits, equal %s, bounds %s" % (W, dt, len(rec), np.array_equal(off, want[0]) and np.array_equal(rec, want[1]), res[0][1]))
    def rank_py(rank, comm):
        c = capi.Ctx(0)
        return shard.kmermatch_exact(c, c.synth(n, lo, hi, 1), comm).download()
    t0 = time.time(); res = run_ranks(W, rank_py); dt = time.time() - t0
    off, rec = merged_hits(res, n)
    print("python W=%d: %.2f s, %d hits, equal %s" % (W, dt, len(rec), np.array_equal(off, want[0]) and np.array_equal(rec, want[1])))
