"""The N > 1 path of bench.py on CPU: two gloo ranks exercise the shard plan, the max-over-ranks timing and the
variable-length all-gather that carries the per-shard contigs (on GPUs the same code runs over RCCL)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_contigs(rank):
    """deterministic per-rank 'packed contigs': rank r has 3 + 2r contigs of lengths 101.. (codes words = ceil(len/16))"""
    rng = np.random.default_rng(100 + rank)
    lens = np.array([101 + 7 * i + rank for i in range(3 + 2 * rank)], np.int32)
    words = int(((lens + 15) // 16).sum())
    codes = rng.integers(-2 ** 31, 2 ** 31 - 1, words, dtype=np.int64).astype(np.int32)
    nmask = rng.integers(0, 2 ** 15, words, dtype=np.int64).astype(np.int16)
    keys = np.arange(len(lens), dtype=np.int32) * 5 + rank
    return codes, nmask, lens, keys


def _pack(rank):
    from carpedeam_amd import dist as cd
    codes, nmask, lens, keys = _fake_contigs(rank)
    n, words = len(lens), len(codes)
    oc, om, ol, ok, total = cd.packed_layout(n, words)
    buf = np.zeros(total, np.int32)
    buf[oc:oc + words] = codes
    buf[om:ol].view(np.int16)[:words] = nmask
    buf[ol:ol + n] = lens
    buf[ok:ok + n] = keys
    return torch.from_numpy(buf), n, words


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from carpedeam_amd import dist as cd
    plan = cd.shard_plan(rank, world, 1001, 1)
    assert plan["seed"] == 1 and plan["n_total"] == 1001 and plan["first"] == (0 if rank == 0 else 500) and plan["n"] == (500 if rank == 0 else 501)
    weak = cd.shard_plan(rank, world, 1000, 1, scaling="weak")
    assert weak["seed"] == 1 + rank and weak["n"] == 1000 and weak["first"] == 0
    t = cd.max_over_ranks(dist, 0.5 + rank)
    buf, n, words = _pack(rank)
    parts = cd.allgather_packed(dist, buf, n, words, plan["first"], world)     # sizes + ONE data all-gather
    c, m, l, k = cd.merge_packed(parts)
    # the generic variable-length helper (one all_gather per tensor)
    codes, nmask, lens, keys = (torch.from_numpy(a) for a in _fake_contigs(rank))
    g = cd.allgather_variable(dist, (codes, lens), world)
    out[rank] = (t, c.numpy().copy(), m.numpy().copy(), l.numpy().copy(), k.numpy().copy(), torch.cat(g[0]).numpy().copy(), torch.cat(g[1]).numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allgather_of_contigs():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    exp_c = np.concatenate([_fake_contigs(r)[0] for r in range(world)])
    exp_m = np.concatenate([_fake_contigs(r)[1] for r in range(world)])
    exp_l = np.concatenate([_fake_contigs(r)[2] for r in range(world)])
    exp_k = np.concatenate([_fake_contigs(r)[3].astype(np.int64) + (0, 500)[r] for r in range(world)])
    for r in range(world):
        t, c, m, l, k, gc, gl = out[r]
        assert t == 1.5                       # max over ranks
        assert (c == exp_c).all() and (m == exp_m).all() and (l == exp_l).all() and (k == exp_k).all()
        assert (gc == exp_c).all() and (gl == exp_l).all()
