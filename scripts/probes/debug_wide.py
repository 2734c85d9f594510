import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from carpedeam_amd import capi
from test_gpu_widekey import databases, READS, CONTIGS
name = sys.argv[1] if len(sys.argv) > 1 else "mixed"
envs = [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[2:]] or [{}]
ctx = capi.Ctx(0)
db = ctx.upload_seqs([s.encode() for s in databases()[name]])
for par in (READS, CONTIGS):
    want = ctx.kmermatch(db, par).download()
    for env in envs:
        os.environ["CDM_FORCE_WIDE_KEY"] = "1"; os.environ["CDM_BUCKET_STATS"] = "1"
        os.environ.update(env)
        got = ctx.kmermatch(db, par).download()
        for k in list(env) + ["CDM_FORCE_WIDE_KEY", "CDM_BUCKET_STATS"]:
            del os.environ[k]
        same = np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        print("k", par.kmer_size, env, "same" if same else "DIFFERENT", len(want[1]), len(got[1]))
        if not same:
            cw = np.diff(want[0]); cg = np.diff(got[0])
            bad = np.nonzero(cw != cg)[0]
            print("  queries with another number of hits:", len(bad), bad[:10], cw[bad[:10]], cg[bad[:10]])
            for q in bad[:3]:
                print("  want", want[1][want[0][q]:want[0][q + 1]][:8]); print("  got ", got[1][got[0][q]:got[0][q + 1]][:8])
