"""CPU side of the exact multi-GPU scheme (carpedeam_amd/shard.py): the pure logic (who holds the reference's run-past-the-end
scan, how the per-range lists chain) and the collectives on two gloo ranks."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from carpedeam_amd import shard  # noqa: E402


def _lst(cnt, target, pos, end):
    a = np.zeros(shard.STALE_LEN, np.uint32)
    a[0], a[1], a[66] = cnt, target, end
    a[2:2 + cnt] = pos
    return a


def test_stale_plan_finds_the_range_that_holds_index_j():
    infos = [{"real": 10, "kept": 6}, {"real": 0, "kept": 0}, {"real": 7, "kept": 5}]
    assert shard.stale_plan(infos) == (2, 1)                       # J = 11 -> second tuple of the third range
    assert shard.stale_plan([{"real": 5, "kept": 5}, {"real": 4, "kept": 4}]) == (None, 0)     # J behind every tuple
    assert shard.stale_plan([{"real": 5, "kept": 2}, {"real": 4, "kept": 1}]) == (0, 3)


def test_combine_stale_chains_over_range_ends():
    z = _lst(0, 0, [], 1)
    # the holder's run ends inside its range
    assert list(shard.combine_stale([_lst(2, 9, [4, 7], 0), _lst(3, 9, [1, 2, 3], 0)], 0)[:4]) == [2, 9, 4, 7]
    # it consumes the range and goes on while the sequence id stays the same, across an empty range
    out = shard.combine_stale([z, _lst(2, 9, [4, 7], 1), z, _lst(1, 9, [5], 0), _lst(1, 9, [6], 0)], 1)
    assert list(out[:5]) == [3, 9, 4, 7, 5]
    # another sequence in the next range stops it; an empty holder list takes the next range's id
    assert list(shard.combine_stale([_lst(1, 9, [4], 1), _lst(2, 8, [1, 2], 0)], 0)[:3]) == [1, 9, 4]
    assert list(shard.combine_stale([_lst(0, 0, [], 1), _lst(2, 8, [1, 2], 0)], 0)[:4]) == [2, 8, 1, 2]
    assert shard.combine_stale([z], None)[0] == 0


def test_owned_ranges_tile_the_sequences():
    for n in (0, 1, 7, 1000):
        for w in (1, 2, 3, 8):
            r = [shard.owned_range(i, w, n) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = shard.TorchComm(dist, rank, world, torch.device("cpu"))
    # rank p sends [100 p + r] * (p + r + 1) to rank r
    pieces = [torch.full((rank + r + 1,), 100 * rank + r, dtype=torch.int64) for r in range(world)]
    send = torch.cat(pieces)
    off = np.concatenate([[0], np.cumsum([p.numel() for p in pieces])])
    recv = comm.exchange(send, off)
    g = comm.all_gather_tensor(torch.arange(3 + rank, dtype=torch.int32) + 10 * rank)
    a = comm.all_gather_array(np.array([rank, 2 ** 40 + rank], np.uint64))
    out[rank] = (recv.numpy().copy(), [x.numpy().copy() for x in g], [x.copy() for x in a])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_keeps_rank_order():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        recv, g, a = out[r]
        exp = np.concatenate([np.full(p + r + 1, 100 * p + r, np.int64) for p in range(world)])       # slices in rank order
        assert (recv == exp).all()
        assert [list(x) for x in g] == [[0, 1, 2], [10, 11, 12, 13]]
        assert [list(x) for x in a] == [[0, 2 ** 40], [1, 2 ** 40 + 1]]


def test_build_cont_follows_the_scan_across_ranks():
    cap = 8
    def head(cnt, target, whole, entries):
        a = np.zeros(cap + 3, np.uint32); a[0], a[1], a[2] = cnt, target, whole; a[3:3 + len(entries)] = entries
        return a
    heads = [head(1, 3, 0, [7]), head(2, 5, 1, [10, 11]), head(0, 0, 1, []), head(1, 5, 0, [12])]
    counts, last = [4, 2, 0, 9], [5, 5, 0, 2]
    # rank 0's last target 5: all of rank 1 (two tuples of target 5, the whole array), rank 2 is empty, one tuple of rank 3, then another target
    assert list(shard.build_cont(0, counts, last, heads)) == [3, 5, 0, 10, 11, 12]
    # rank 1's last target 5: rank 3's head, which is not its whole array -> the left-over list is not reached
    assert list(shard.build_cont(1, counts, last, heads)) == [1, 5, 0, 12]
    assert shard.build_cont(2, counts, last, heads) is None
    assert list(shard.build_cont(3, counts, last, heads)) == [0, 2, 1]                 # nothing behind the last rank: on into the left-overs
    assert list(shard.build_cont(0, [4, 0, 0, 9], [6, 0, 0, 2], heads)) == [0, 6, 0]   # next tuples have another target: the scan stops
