"""debug: slot layout against packed on a uniform DB; prints the first differing records"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from carpedeam_amd import capi
from test_gpu_slotlayout import uniform_reads, text
ctx = capi.Ctx(0)
kw = eval(sys.argv[3]) if len(sys.argv) > 3 else dict(dup=0.05, lowc=30, with_n=0.02)
L, n = int(sys.argv[1]), int(sys.argv[2])
seqs = uniform_reads(n, L, seed=L * 7 + n, **kw)
keyed = {i: (s.encode() + b"\n", 0) for i, s in enumerate(seqs)}
os.environ["CDM_KMER_LAYOUT"] = "packed"; a = text(ctx, keyed)
os.environ["CDM_KMER_LAYOUT"] = "slot"; b = text(ctx, keyed)
bad = [k for k in a if a[k] != b.get(k)]
print(len(bad), "of", len(a), "keys differ")
for k in bad[:6]:
    la, lb = a[k][0].decode().split("\n"), b[k][0].decode().split("\n")
    print("key", k, seqs[k])
    print("  packed only:", [x for x in la if x not in lb][:8])
    print("  slot only:  ", [x for x in lb if x not in la][:8])
