"""One-process-per-GPU scaling of the hot path (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).

The read corpus partitions into independent shards (north star: "reads shard trivially"): every rank runs the four stages
on its own shard with no data-path collective, then ONE all-gather hands every rank the contigs of all shards.
"""
import numpy as np


def shard_plan(rank, world, reads_per_gpu, seed):
    """Weak-scaling plan: rank r works on its own corpus of reads_per_gpu reads (generator seed + r)."""
    return {"rank": rank, "world": world, "n": int(reads_per_gpu), "seed": int(seed) + int(rank)}


def max_over_ranks(dist, seconds, device="cpu"):
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allgather_variable(dist, tensors, world):
    """All-gather a tuple of 1-D tensors whose lengths differ between ranks (padding to the maximum, ONE size exchange and
    one all_gather per tensor).  Returns, per tensor, the list of the ranks' un-padded parts."""
    import torch
    sizes = torch.tensor([int(t.numel()) for t in tensors], dtype=torch.int64, device=tensors[0].device)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    out = []
    for j, t in enumerate(tensors):
        mx = max(int(s[j]) for s in all_sizes)
        pad = torch.zeros(max(mx, 1), dtype=t.dtype, device=t.device)
        pad[: t.numel()] = t
        parts = [torch.zeros_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        out.append([p[: int(all_sizes[r][j])] for r, p in enumerate(parts)])
    return out


def merge_contig_parts(codes, nmask, lens, keys, key_stride):
    """Concatenate the ranks' packed contig sets; keys become rank * key_stride + key so they stay unique and ordered."""
    import torch
    k = [kk.to(torch.int64) + r * int(key_stride) for r, kk in enumerate(keys)]
    return torch.cat(codes), torch.cat(nmask), torch.cat(lens), torch.cat(k)


def allgather_contigs(dist, ctx, asm_db, world, key_stride):
    """GPU path: contigs (wasExtended == 1) of this rank's assembled DB -> packed device tensors -> RCCL all-gather -> one
    device DB holding the contigs of every shard (on every rank)."""
    import torch
    contigs = asm_db.select_ext()
    n, words = contigs.n, contigs.words
    dev = torch.device("cuda", torch.cuda.current_device())
    codes = torch.zeros(max(words, 1), dtype=torch.int32, device=dev)
    nmask = torch.zeros(max(words, 1), dtype=torch.int16, device=dev)
    lens = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    keys = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    contigs.copy_packed(codes.data_ptr(), nmask.data_ptr(), lens.data_ptr(), keys.data_ptr())
    # the N bit planes travel widened to int32 (gloo, used by the CPU tests of this path, has no 16-bit integer type)
    g = allgather_variable(dist, (codes[:words], nmask[:words].to(torch.int32), lens[:n], keys[:n]), world)
    c, m, l, k = merge_contig_parts(*g, key_stride=key_stride)
    m = m.to(torch.int16)
    if int(k.max().item()) >= 2 ** 32 if k.numel() else False:
        raise ValueError("contig keys overflow 32 bits")
    k32 = k.to(torch.int32).contiguous()
    c, m, l = c.contiguous(), m.contiguous(), l.contiguous()
    torch.cuda.synchronize()
    return ctx.from_packed(c.data_ptr(), m.data_ptr(), l.data_ptr(), k32.data_ptr(), int(l.numel()), int(c.numel()), 1)
