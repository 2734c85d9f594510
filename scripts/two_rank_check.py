"""Two real PROCESSES (torch.distributed, gloo, device tensors staged through the host) on ONE GPU: the exact multi-GPU scheme's
iteration and the contig all-gather of the "reads" scheme, each rank checking its result against the single-device calls it makes
itself.  What the threads-as-ranks tests cannot see - torch's stream against the library's stream across processes, every rank with
its own context and allocator - runs here.  Launched by tests/test_gpu_two_ranks.py:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P scripts/two_rank_check.py <dhigh prefix>"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carpedeam_amd import capi, shard  # noqa: E402
from carpedeam_amd import dist as cd  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ctx = capi.Ctx(0)
ctx.damage_load(sys.argv[1])
same = lambda x, y: [bytes(a) for a in x[0]] == [bytes(a) for a in y[0]] and np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2])

# ---- exact scheme: every rank holds the corpus, kmermatcher by k-mer range, one exchange of group keys, query-sharded stages
db = ctx.synth(60_000, 60, 150, 7)
h, a, co, nx = shard.exact_iteration(ctx, db, shard.TorchComm(dist, rank, world, dev))
h0 = ctx.kmermatch(db); a0 = ctx.rescore(db, h0); c0 = ctx.correct(db, a0); n0 = ctx.extend(c0, a0)
(off, rec), (off0, rec0) = h.download(), h0.download()
lo, hi = shard.owned_range(rank, world, db.n)          # a rank's prefilter result holds the rows of the representatives it owns
for q in range(lo, hi, 37):
    assert np.array_equal(rec[int(off[q]):int(off[q + 1])], rec0[int(off0[q]):int(off0[q + 1])]), "hits of query %d differ on rank %d" % (q, rank)
assert int(off[hi] - off[lo]) == int(off0[hi] - off0[lo]), "hit count differs on rank %d" % rank
assert same(co.download(), c0.download()), "corrected DB differs on rank %d" % rank
assert same(nx.download(), n0.download()), "next DB differs on rank %d" % rank

# ---- the same through the LIBRARY's own calling sequence (csrc/dist.hip, what a deployment runs over RCCL), here over a gloo transport
comm = capi.Comm.from_transport(ctx, rank, world, shard.GlooTransport(dist, rank, world))
h2, a2, co2, nx2 = comm.reads_iteration(db)
(off2, rec2) = h2.download()
own = comm.owned(db.n)                                   # the library cuts the owners' id ranges by group keys, not by ids
lo2, hi2 = int(own[rank]), int(own[rank + 1])
for q in range(lo2, hi2, 37):
    assert np.array_equal(rec2[int(off2[q]):int(off2[q + 1])], rec0[int(off0[q]):int(off0[q + 1])]), "native hits of query %d differ on rank %d" % (q, rank)
assert int(off2[hi2] - off2[lo2]) == int(off0[hi2] - off0[lo2]), "native hit count differs on rank %d" % rank
assert same(co2.download(), c0.download()) and same(nx2.download(), n0.download()), "native DBs differ on rank %d" % rank
del comm, h2, a2, co2, nx2

# ---- reads scheme: each rank its own shard of ONE corpus, then the all-gather of the contigs; every rank must end up holding the
# contigs of both shards, in shard order
plan = cd.shard_plan(rank, world, 40_000, 3, "strong")
mine = ctx.synth(plan["n"], 100, 100, plan["seed"], n_total=plan["n_total"], first=plan["first"])
hm = ctx.kmermatch(mine); am = ctx.rescore(mine, hm); cm = ctx.correct(mine, am); asm = ctx.extend(cm, am)
allc = cd.allgather_contigs(dist, ctx, asm, world, key_base=plan["first"])
got = allc.download()
want_seqs, want_keys = [], []
for r in range(world):
    p = cd.shard_plan(r, world, 40_000, 3, "strong")
    d = ctx.synth(p["n"], 100, 100, p["seed"], n_total=p["n_total"], first=p["first"])
    hh = ctx.kmermatch(d); aa = ctx.rescore(d, hh); cc = ctx.correct(d, aa)
    s, k, e = ctx.extend(cc, aa).download()
    want_seqs += [bytes(x) for x, f in zip(s, e) if f == 1]
    want_keys += [int(kk) + p["first"] for kk, f in zip(k, e) if f == 1]
assert [bytes(x) for x in got[0]] == want_seqs and [int(k) for k in got[1]] == want_keys, "gathered contigs differ on rank %d" % rank
dist.barrier()
print("rank %d of %d ok: %d hits, %d gathered contigs" % (rank, world, h.count, allc.n), flush=True)
dist.destroy_process_group()
