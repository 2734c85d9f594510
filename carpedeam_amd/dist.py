"""One-process-per-GPU scaling of the hot path (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).

Two ways to spread ONE read corpus over W ranks (BASELINE.json configs[3]):

* scheme "reads" (the north star's "reads shard trivially"): rank r owns the reads [r n/W, (r+1) n/W) and runs the four
  stages on them alone - no data-path collective - then ONE all-gather hands every rank the contigs of all shards.  Overlaps
  between reads of different shards are not seen, so the result differs from the single-device run (tests/test_gpu_shards.py
  measures by how much); it equals the reference run shard by shard.
* scheme "exact" (SURVEY.md 8(e)): every rank holds the whole packed read DB; kmermatcher is split by k-mer range (the
  reference's MPI split, lib/mmseqs/src/linclust/kmermatcher.cpp:634-663), the (rep, id, diagonal) group tuples go to the
  owner of their representative in ONE all-to-all, rescore / correction / extension run on the rank's query range, and the
  new sequences are all-gathered.  Bit-identical to the single-device result (carpedeam_amd/shard.py).

`scaling` says what N ranks share: "strong" = one corpus of n reads split over the ranks (default, config 4), "weak" = every
rank gets its own n-read corpus (generator seed + rank).
"""
import numpy as np


def shard_plan(rank, world, n_reads, seed, scaling="strong"):
    """Which reads rank `rank` of `world` generates / owns: {n_total, first, n, seed} for cdm_seqdb_synth."""
    n_reads, rank, world = int(n_reads), int(rank), int(world)
    if scaling == "weak":
        return {"rank": rank, "world": world, "n_total": n_reads, "first": 0, "n": n_reads, "seed": int(seed) + rank, "scaling": "weak"}
    if scaling != "strong":
        raise ValueError("scaling must be strong or weak")
    first = rank * n_reads // world
    last = (rank + 1) * n_reads // world
    return {"rank": rank, "world": world, "n_total": n_reads, "first": first, "n": last - first, "seed": int(seed), "scaling": "strong"}


def max_over_ranks(dist, seconds, device="cpu"):
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ------------------------------------------------------------------------------------------------ one buffer, one all-gather
# A rank's contig set travels as ONE int32 buffer: [codes: words][N planes, 16 bit per code word: ceil(words/2)][lengths: n]
# [keys: n].  The sizes (n, words, key base) go ahead in a 3-number all-gather; then a single all_gather moves the padded
# buffers - the "single RCCL all-gather of per-shard contigs" of the north star.
def packed_layout(n, words):
    """offsets (in int32 elements) of the four sections and the total"""
    o_codes = 0
    o_mask = o_codes + words
    o_len = o_mask + (words + 1) // 2
    o_key = o_len + n
    return o_codes, o_mask, o_len, o_key, o_key + n


def raw_layout(n, words):
    """a contig set with letters beyond ACGTN (lower case, IUPAC codes) carries two more sections behind packed_layout's:
    [original bytes, 16 per code word: 4 * words][one byte per sequence, 1 = its row counts: ceil(n/4)] -> (o_raw, o_flags, total)"""
    o_raw = packed_layout(n, words)[4]
    o_flags = o_raw + 4 * words
    return o_raw, o_flags, o_flags + (n + 3) // 4


def allgather_packed(dist, buf, n, words, key_base, world):
    """buf: this rank's packed int32 buffer (packed_layout(n, words), or raw_layout's longer form).  Returns [(buf_r, n_r, words_r,
    key_base_r)] for all ranks.  Two collectives: the sizes (4 int64) and the data."""
    import torch
    home = buf.device
    if buf.is_cuda and dist.get_backend() == "gloo":       # gloo moves host memory: stage the buffer (tests: two ranks on one GPU)
        buf = buf.cpu()
    meta = torch.tensor([int(n), int(words), int(key_base), int(buf.numel())], dtype=torch.int64, device=buf.device)
    metas = torch.zeros(world * 4, dtype=torch.int64, device=buf.device)
    dist.all_gather_into_tensor(metas, meta)
    metas = [[int(v) for v in m] for m in metas.view(world, 4).tolist()]
    mx = max(max(m[3] for m in metas), 1)
    if buf.numel() == mx:
        pad = buf.contiguous()
    else:
        pad = torch.zeros(mx, dtype=torch.int32, device=buf.device)
        pad[: buf.numel()] = buf
    flat = torch.empty(world * mx, dtype=torch.int32, device=buf.device)      # ONE receive buffer, no per-rank tensor list
    dist.all_gather_into_tensor(flat, pad)
    if flat.device != home:
        flat = flat.to(home)
    return [(flat[r * mx: r * mx + m[3]], m[0], m[1], m[2]) for r, m in enumerate(metas)]


def merge_packed(parts):
    """Concatenate the ranks' packed sets into (codes int32[words], nmask int16[words], lens int32[n], keys int64[n]);
    keys become key_base_r + key so that they stay unique and ordered across the shards."""
    import torch
    codes, masks, lens, keys = [], [], [], []
    for buf, n, words, base in parts:
        oc, om, ol, ok, _ = packed_layout(n, words)
        codes.append(buf[oc: oc + words])
        masks.append(buf[om: ol].view(torch.int16)[:words])
        lens.append(buf[ol: ol + n])
        keys.append(buf[ok: ok + n].to(torch.int64) + int(base))
    return torch.cat(codes), torch.cat(masks), torch.cat(lens), torch.cat(keys)


def merge_raw(parts):
    """(original bytes uint8[16 * words], row flags uint8[n]) of the merged set, or None when no rank's contigs carry letters
    beyond ACGTN (a rank without such letters contributes rows that do not count)"""
    import torch
    if not any(buf.numel() > packed_layout(n, words)[4] for buf, n, words, _ in parts):
        return None
    raws, flags = [], []
    for buf, n, words, _ in parts:
        if buf.numel() > packed_layout(n, words)[4]:
            o_raw, o_flags, total = raw_layout(n, words)
            raws.append(buf[o_raw: o_flags].view(torch.uint8)[: 16 * words])
            flags.append(buf[o_flags: total].view(torch.uint8)[:n])
        else:
            raws.append(torch.zeros(16 * words, dtype=torch.uint8, device=buf.device))
            flags.append(torch.zeros(n, dtype=torch.uint8, device=buf.device))
    return torch.cat(raws), torch.cat(flags)


def allgather_variable(dist, tensors, world):
    """All-gather a tuple of 1-D tensors whose lengths differ between ranks (padding to the maximum; one size exchange and
    one all_gather per tensor).  Generic helper (shard.py's group-tuple exchange on gloo uses it); the contig hand-off uses
    the single-buffer form above."""
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


def pack_contigs(ctx, asm_db):
    """The contigs (wasExtended == 1) of an assembled DB as one packed int32 device tensor -> (buf, n, words)."""
    import torch
    contigs = asm_db.select_ext()
    n, words = contigs.n, contigs.words
    dev = torch.device("cuda", torch.cuda.current_device())
    oc, om, ol, ok, total = packed_layout(n, words)
    o_raw, o_flags, total_raw = raw_layout(n, words)
    if contigs.has_raw:
        total = total_raw
    buf = torch.zeros(max(total, 1), dtype=torch.int32, device=dev)
    # the library copies on its own (non-blocking) stream: torch's zero fill has to be complete before it starts
    torch.cuda.synchronize()
    base = buf.data_ptr()
    contigs.copy_packed(base + 4 * oc, base + 4 * om, base + 4 * ol, base + 4 * ok)     # (synchronises the library's stream)
    if contigs.has_raw:
        contigs.copy_raw(base + 4 * o_raw, base + 4 * o_flags)
    return buf[:total], n, words


def unpack_to_db(ctx, parts, ext_value=1):
    """merged device DB from allgather_packed's parts"""
    import torch
    c, m, l, k = merge_packed(parts)
    if k.numel() and int(k.max().item()) >= 2 ** 32:
        raise ValueError("contig keys overflow 32 bits")
    k32 = k.to(torch.int32).contiguous()
    c, m, l = c.contiguous(), m.contiguous(), l.contiguous()
    raw = merge_raw(parts)
    if raw is not None:
        raw = (raw[0].contiguous(), raw[1].contiguous())
    torch.cuda.synchronize()     # torch's stream wrote the merged tensors; the library reads them on its own stream
    db = ctx.from_packed(c.data_ptr(), m.data_ptr(), l.data_ptr(), k32.data_ptr(), int(l.numel()), int(c.numel()), ext_value)
    if raw is not None and int(c.numel()):
        db.attach_raw(raw[0].data_ptr(), raw[1].data_ptr())
    return db


def allgather_contigs(dist, ctx, asm_db, world, key_base):
    """GPU path of scheme "reads": contigs of this rank's assembled DB -> one packed device buffer -> ONE RCCL all-gather ->
    a device DB holding the contigs of every shard (on every rank)."""
    buf, n, words = pack_contigs(ctx, asm_db)
    parts = allgather_packed(dist, buf, n, words, key_base, world)
    return unpack_to_db(ctx, parts)
