import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from carpedeam_amd import capi
ctx = capi.Ctx(0)
db = ctx.synth(int(sys.argv[1]), 100, 100, 5)
t0 = time.perf_counter(); a = ctx.kmermatch(db).download(); t1 = time.perf_counter()
os.environ["CDM_KMER_PASSES"] = sys.argv[2]; os.environ["CDM_BUCKET_STATS"] = "1"
b = ctx.kmermatch(db).download(); t2 = time.perf_counter()
print("one pass %.3f s, passes %s: %.3f s, equal: %s, hits %d" % (t1 - t0, sys.argv[2], t2 - t1, np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), len(a[1])))
