"""Debug helper: one small database of tests/test_gpu_shards.py::test_exact_kmermatcher_on_small_databases, split into k-mer
ranges on one GPU; prints the records that differ from the single-device result and the per-range bookkeeping."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from carpedeam_amd import capi, shard  # noqa: E402
from test_gpu_shards import merged_hits, run_ranks  # noqa: E402

want_case, world = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(77)
letters = np.frombuffer(b"ACGT", np.uint8)
for case in range(want_case + 1):
    genome = rng.integers(0, 4, 260)
    seqs = []
    for _ in range(int(rng.integers(3, 50))):
        L = int(rng.integers(18, 110)); st = int(rng.integers(0, 260 - L))
        c = genome[st:st + L].copy()
        if rng.random() < 0.5:
            c = (3 - c)[::-1]
        seqs.append(letters[c].tobytes())
    seqs += [seqs[0]] * int(rng.integers(0, 3)) + [b"ACG", b""][: int(rng.integers(0, 3))]
ref = capi.Ctx(0)
db = ref.upload_seqs(seqs)
want = ref.kmermatch(db).download()
print("sequences", len(seqs), "lengths", [len(s) for s in seqs])
one = ref.kmermatch_part(db, 0, 1)
print("single range:", one.info(), "stale at kept:", list(one.stale(one.info()["kept"])[:6]))
for j in range(one.info()["kept"] - 3, one.info()["kept"] + 4):
    print("   single j", j, list(one.stale(j)[:5]))
del one
parts = [ref.kmermatch_part(db, p, world) for p in range(world)]
for j in range(190, 200):
    print("   part1 j", j, list(parts[1].stale(j)[:5]))
infos = [p.info() for p in parts]
print("infos", infos)
holder, jl = shard.stale_plan(infos)
print("stale plan", holder, jl)
for p in parts:
    print("  head", list(p.stale(0)[:8]), "end", p.stale(0)[66])
if holder is not None:
    print("  holder list", list(parts[holder].stale(jl)[:10]), parts[holder].stale(jl)[66])
del parts


def rank_fn(rank, comm):
    c = capi.Ctx(0)
    return shard.kmermatch_exact(c, c.upload_seqs(seqs), comm).download()


off, rec = merged_hits(run_ranks(world, rank_fn), len(seqs))
print("offsets equal", np.array_equal(off, want[0]))
for q in range(len(seqs)):
    a = rec[int(off[q]):int(off[q + 1])]; b = want[1][int(want[0][q]):int(want[0][q + 1])]
    if not np.array_equal(a, b):
        print("query", q, "got", a.tolist(), "want", b.tolist())
