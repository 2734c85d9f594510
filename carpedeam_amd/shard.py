"""Scheme "exact" of the multi-GPU run: one reads-loop iteration spread over W ranks with results that are
bit-identical to the single-device run.

Every rank holds the whole packed sequence DB.  Per iteration:
  1. kmermatcher, first half, on the rank's k-mer RANGE (cdm_kmermatch_part: the reference's MPI split of the k-mer space,
     lib/mmseqs/src/linclust/kmermatcher.cpp:634-663, by value so that rank order is k-mer order)
  2. a few integers per rank are all-gathered (tuple counts; the candidates of the reference's run-past-the-end scan)
  3. ONE all-to-all moves every (rep, id, diagonal, strand) group key to the rank that owns its representative
     (representatives [r n/W, (r+1) n/W) belong to rank r); received slices are concatenated in rank = k-mer order, which is
     what keeps the reference's tie order
  4. kmermatcher, second half: sort 2 (cdm_kpart_sort), one more small all-gather (the heads of the sorted arrays: the
     reference's per-target scan runs on into the next representative's tuples, i.e. across rank boundaries), then the vote
     (cdm_kpart_vote): hits of the owned representatives
  5. rescorediagonal, ancient_correction on the owned queries (the other queries carry their self hit only); the corrected
     owned ranges are all-gathered so that every rank holds the corrected DB the extension stage looks targets up in
  6. ancient_read_assemble on the owned queries; the new owned ranges are all-gathered: every rank holds the next iteration's DB

The collectives go through a small `Comm` interface: TorchComm (torch.distributed: RCCL on GPUs, gloo in the CPU tests) and
ThreadComm (W ranks as threads of one process on one GPU: what tests/test_gpu_shards.py uses to check the scheme against the
single-device result without a multi-GPU box).
"""
import threading

import numpy as np

STALE_MAX = 62          # csrc/kmermatch.hip
STALE_LEN = 67          # [0] count, [1] sequence id, [2..63] positions, [66] the scan reached the end of the range


# ------------------------------------------------------------------------------------------------ pure logic (CPU-testable)
def owned_range(rank, world, n):
    return rank * n // world, (rank + 1) * n // world


def stale_plan(infos):
    """infos: per range, in k-mer order, {"real": real tuples, "kept": kept group keys}.  The reference's last per-target scan
    runs from index J = (all kept group keys) of the k-mer-ordered tuple array on.  -> (holder range or None, local index)"""
    j = sum(i["kept"] for i in infos)
    base = 0
    for h, i in enumerate(infos):
        if j < base + i["real"]:
            return h, j - base
        base += i["real"]
    return None, 0


def combine_stale(lists, holder):
    """lists[p]: range p's 67-value list (the holder's from its local index, the later ranges' from their first tuple).
    The scan collects tuples while they belong to one sequence; it goes on into the next range when it consumed the current one."""
    out = np.zeros(STALE_LEN, np.uint32)
    if holder is None:
        return out
    cnt, target, pos = 0, None, []
    for p in range(holder, len(lists)):
        l = lists[p]
        c = int(l[0])
        if c:
            if target is None:
                target = int(l[1])
            elif int(l[1]) != target:
                break
            pos.extend(int(x) for x in l[2:2 + c])
            cnt += c
        if not int(l[66]):
            break
    if cnt >= STALE_MAX:
        raise RuntimeError("the reference's last per-target scan would run over %d or more left-over tuples; not reproduced" % STALE_MAX)
    out[0] = cnt
    out[1] = target if target is not None else 0
    out[2:2 + cnt] = pos
    return out


def build_cont(rank, counts, last_targets, heads):
    """What the per-target scan of rank `rank`'s last segment runs into (VoteArgs::cont of csrc/kmermatch.hip): the heads of the
    later ranks' sorted arrays while they carry the same target id and are consumed completely; [0] entries, [1] the target id,
    [2] = 1 if the scan then reaches the left-over list, [3..] entries.  None: no tuples on this rank."""
    if counts[rank] == 0:
        return None
    t = int(last_targets[rank])
    entries, then_stale = [], 1
    for p in range(rank + 1, len(counts)):
        if counts[p] == 0:
            continue
        h = heads[p]
        if int(h[1]) != t:
            then_stale = 0
            break
        c = int(h[0])
        if c > len(h) - 3:
            raise RuntimeError("the scan of a rank's last target runs over more than %d tuples of the next rank; not reproduced" % (len(h) - 3))
        entries.extend(int(x) for x in h[3:3 + c])
        if not int(h[2]):
            then_stale = 0
            break
    return np.array([len(entries), t, then_stale] + entries, np.uint32)


def seq_section_layout(n, words):
    """one int32 buffer for a range of sequences: [codes: words][N planes: ceil(words/2)][lengths: n][keys: n][ext: ceil(n/4)]"""
    o_codes = 0
    o_mask = o_codes + words
    o_len = o_mask + (words + 1) // 2
    o_key = o_len + n
    o_ext = o_key + n
    return o_codes, o_mask, o_len, o_key, o_ext, o_ext + (n + 3) // 4


# ------------------------------------------------------------------------------------------------ communicators
class TorchComm:
    def __init__(self, dist, rank, world, device):
        self.dist, self.rank, self.world, self.device = dist, rank, world, device
        self.backend = dist.get_backend()
        # gloo moves host memory: device tensors are staged through the host around every collective (two ranks sharing ONE GPU in
        # the tests - RCCL wants a device per rank - and any deployment without RCCL)
        self.stage = self.backend == "gloo" and getattr(device, "type", str(device)) == "cuda"

    def _out(self, t):
        return t.cpu() if self.stage and t.is_cuda else t

    def _back(self, t):
        return t.to(self.device) if self.stage else t

    def all_gather_array(self, a):
        """a: 1-D numpy array of the same length and dtype (<= 8 bytes per element) on every rank -> list of arrays"""
        import torch
        a = np.ascontiguousarray(a)
        t = torch.from_numpy(a.astype(np.int64))
        if not self.stage:
            t = t.to(self.device)
        parts = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t)
        return [p.cpu().numpy().astype(a.dtype) for p in parts]

    def exchange(self, send, offsets):
        """send: 1-D tensor, offsets[r]..offsets[r+1] goes to rank r -> what this rank receives, concatenated in rank order.
        One all_to_all_single on RCCL; gloo (CPU tests) has no all-to-all: an all-gather of the padded buffers stands in."""
        import torch
        offsets = [int(x) for x in offsets]
        counts = np.array([offsets[r + 1] - offsets[r] for r in range(self.world)], np.int64)
        all_counts = self.all_gather_array(counts)                 # all_counts[p][r] = what p sends to r
        recv_counts = [int(all_counts[p][self.rank]) for p in range(self.world)]
        if self.backend == "nccl":
            recv = torch.empty(sum(recv_counts), dtype=send.dtype, device=send.device)
            self.dist.all_to_all_single(recv, send[: offsets[-1]].contiguous(), output_split_sizes=recv_counts, input_split_sizes=[int(c) for c in counts])
            return recv
        mx = max(1, max(int(c.sum()) for c in all_counts))
        src = self._out(send[: offsets[-1]])
        pad = torch.zeros(mx, dtype=send.dtype, device=src.device)
        pad[: offsets[-1]] = src
        bufs = [torch.zeros_like(pad) for _ in range(self.world)]
        self.dist.all_gather(bufs, pad)
        out = []
        for p in range(self.world):
            o = np.concatenate([[0], np.cumsum(all_counts[p])])
            out.append(bufs[p][int(o[self.rank]): int(o[self.rank + 1])])
        return self._back(torch.cat(out))

    def all_gather_tensor(self, t):
        """variable-length 1-D tensors -> list of the ranks' tensors (sizes first, then ONE padded all_gather)"""
        import torch
        sizes = self.all_gather_array(np.array([t.numel()], np.int64))
        mx = max(1, max(int(s[0]) for s in sizes))
        src = self._out(t)
        pad = torch.zeros(mx, dtype=t.dtype, device=src.device)
        pad[: t.numel()] = src
        bufs = [torch.empty_like(pad) for _ in range(self.world)]
        self.dist.all_gather(bufs, pad)
        return [self._back(b[: int(s[0])]) for b, s in zip(bufs, sizes)]


class ThreadComm:
    """W ranks as threads of one process (one GPU): collectives are slots of a shared list between two barriers."""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared, rank):
        self.sh, self.rank, self.world = shared, rank, shared.world

    def _gather(self, x):
        self.sh.slots[self.rank] = x
        self.sh.barrier.wait()
        out = list(self.sh.slots)
        self.sh.barrier.wait()
        return out

    def all_gather_array(self, a):
        return [np.array(x) for x in self._gather(np.ascontiguousarray(a).copy())]

    def exchange(self, send, offsets):
        import torch
        offsets = [int(x) for x in offsets]
        parts = self._gather((send, offsets))
        torch.cuda.synchronize()
        out = torch.cat([s[o[self.rank]: o[self.rank + 1]] for s, o in parts]).clone()
        torch.cuda.synchronize()
        self.sh.barrier.wait()          # nobody frees its send buffer before everybody has copied
        return out

    def all_gather_tensor(self, t):
        import torch
        parts = self._gather(t)
        torch.cuda.synchronize()
        out = [p.clone() for p in parts]
        torch.cuda.synchronize()
        self.sh.barrier.wait()
        return out


# ------------------------------------------------------------------------------------------------ the stages
def kmermatch_exact(ctx, db, comm, par=None):
    """kmermatcher over comm.world ranks: the hits of the representatives this rank owns (self hits for all other sequences);
    the union over the ranks is the single-device result."""
    import torch
    part = ctx.kmermatch_part(db, comm.rank, comm.world, par)
    info = part.info()
    infos = [{"real": int(a[0]), "kept": int(a[1])} for a in comm.all_gather_array(np.array([info["real"], info["kept"]], np.uint64))]
    holder, j_local = stale_plan(infos)
    mine = np.zeros(STALE_LEN, np.uint32)
    if holder is not None and comm.rank >= holder:
        mine = part.stale(j_local if comm.rank == holder else 0)
    stale = combine_stale(comm.all_gather_array(mine), holder)
    off, ptr = part.gather(comm.world)
    kept = int(off[-1])
    send = torch.empty(max(kept, 1), dtype=torch.int64, device=torch.device("cuda", torch.cuda.current_device()))
    torch.cuda.synchronize()
    if kept:
        ctx.dev_copy(send.data_ptr(), ptr, kept * 8)
    recv = comm.exchange(send[:kept], off)                         # the all-to-all of the group keys
    torch.cuda.synchronize()
    head, count, last_target = part.sort(recv.data_ptr() if recv.numel() else None, recv.numel())
    # the reference's per-target scan does not stop at a representative's last tuple (kmermatcher.cpp:875-887): a rank's last scan
    # runs into the next rank's first tuples, so the heads of the sorted arrays go round once more (a few KB per rank)
    heads = comm.all_gather_array(np.concatenate([np.array([count, last_target], np.uint64), head.astype(np.uint64)]))
    cont = build_cont(comm.rank, [int(h[0]) for h in heads], [int(h[1]) for h in heads], [h[2:].astype(np.uint32) for h in heads])
    return part.vote(cont, stale)


def pack_owned(ctx, db, lo, hi):
    """sequences [lo, hi) of a device DB as one int32 device tensor (seq_section_layout) -> (buf, n, words)"""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    n, words = db.n, db.words
    codes = torch.empty(max(words, 1), dtype=torch.int32, device=dev)
    mask = torch.empty(max(words, 1), dtype=torch.int16, device=dev)
    lens = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    keys = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    ext = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    db.copy_packed(codes.data_ptr(), mask.data_ptr(), lens.data_ptr(), keys.data_ptr())
    db.copy_ext(ext.data_ptr())
    woff = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    woff[1:] = torch.cumsum((lens[:n].to(torch.int64) + 15) // 16, 0)
    w0, w1 = int(woff[lo].item()), int(woff[hi].item())
    m, w = hi - lo, w1 - w0
    oc, om, ol, ok, oe, total = seq_section_layout(m, w)
    has_raw = db.has_raw
    if has_raw:         # letters beyond ACGTN: [original bytes, 16 per code word: 4 * w][row flags: ceil(m/4)] behind the rest
        o_raw, o_flags, total = total, total + 4 * w, total + 4 * w + (m + 3) // 4
    buf = torch.zeros(max(total, 1), dtype=torch.int32, device=dev)
    if has_raw:
        raw = torch.empty(max(16 * words, 1), dtype=torch.uint8, device=dev)
        flags = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        db.copy_raw(raw.data_ptr(), flags.data_ptr())
        buf[o_raw: o_flags].view(torch.uint8)[: 16 * w] = raw[16 * w0: 16 * w1]
        buf[o_flags: total].view(torch.uint8)[:m] = flags[lo:hi]
    buf[oc: oc + w] = codes[w0:w1]
    buf[om: ol].view(torch.int16)[:w] = mask[w0:w1]
    buf[ol: ol + m] = lens[lo:hi]
    buf[ok: ok + m] = keys[lo:hi]
    buf[oe: total].view(torch.uint8)[:m] = ext[lo:hi]
    return buf[:total], m, w


def merge_owned(ctx, db_local, comm):
    """All-gather the owned sequence ranges of the ranks' result DBs (same number of sequences everywhere) into the full DB."""
    import torch
    n = db_local.n
    lo, hi = owned_range(comm.rank, comm.world, n)
    buf, m, w = pack_owned(ctx, db_local, lo, hi)
    metas = comm.all_gather_array(np.array([m, w, int(buf.numel())], np.int64))
    bufs = comm.all_gather_tensor(buf)
    codes, masks, lens, keys, exts, raws, flags = [], [], [], [], [], [], []
    any_raw = any(int(sz) > seq_section_layout(int(mm), int(ww))[5] for mm, ww, sz in metas)
    for b, (mm, ww, sz) in zip(bufs, metas):
        mm, ww = int(mm), int(ww)
        oc, om, ol, ok, oe, total = seq_section_layout(mm, ww)
        codes.append(b[oc: oc + ww]); masks.append(b[om: ol].view(torch.int16)[:ww]); lens.append(b[ol: ol + mm]); keys.append(b[ok: ok + mm])
        exts.append(b[oe: total].view(torch.uint8)[:mm])
        if int(sz) > total:
            raws.append(b[total: total + 4 * ww].view(torch.uint8)[: 16 * ww]); flags.append(b[total + 4 * ww: total + 4 * ww + (mm + 3) // 4].view(torch.uint8)[:mm])
        elif any_raw:
            raws.append(torch.zeros(16 * ww, dtype=torch.uint8, device=b.device)); flags.append(torch.zeros(mm, dtype=torch.uint8, device=b.device))
    c, k16, l, k, e = (torch.cat(x).contiguous() for x in (codes, masks, lens, keys, exts))
    if any_raw:
        r8, f8 = torch.cat(raws).contiguous(), torch.cat(flags).contiguous()
    torch.cuda.synchronize()
    out = ctx.from_packed_ext(c.data_ptr(), k16.data_ptr(), l.data_ptr(), k.data_ptr(), e.data_ptr(), int(l.numel()), int(c.numel()))
    if any_raw and int(c.numel()):
        out.attach_raw(r8.data_ptr(), f8.data_ptr())
    return out


def exact_iteration(ctx, db, comm, kpar=None, rpar=None, apar=None):
    """One iteration of the reads loop (data/nuclassemble.sh:100-146) over comm.world ranks -> (hits, alns, corrected DB, next DB),
    the two DBs complete on every rank and identical to the single-device ones."""
    hits = kmermatch_exact(ctx, db, comm, kpar)
    alns = ctx.rescore(db, hits, rpar)
    corr = merge_owned(ctx, ctx.correct(db, alns, apar), comm)
    asm = merge_owned(ctx, ctx.extend(corr, alns, apar), comm)
    return hits, alns, corr, asm


# ------------------------------------------------------------------------------------------------ transports for the library's own multi-GPU calls
# carpedeam_amd.capi.Comm.from_transport binds these to cdm_comm_create_ops (csrc/dist.hip): the C++ calling sequence then runs with
# several ranks on ONE device, where RCCL - the transport of a deployment, Comm.rccl - wants a device per rank.
class ThreadTransport:
    """ranks = threads of one process on one device (ThreadComm.Shared): peers read each other's device memory directly"""

    def __init__(self, shared, rank, ctx):
        self.sh, self.rank, self.world, self.ctx, self.error = shared, rank, shared.world, ctx, None

    def _gather(self, x):
        self.sh.slots[self.rank] = x
        self.sh.barrier.wait()
        out = list(self.sh.slots)
        self.sh.barrier.wait()
        return out

    def all_gather_host(self, b):
        return self._gather(bytes(b))

    def all_to_all_dev(self, send, soff, recv, roff):
        parts = self._gather((send, soff))
        for p, (ptr, off) in enumerate(parts):
            n = off[self.rank + 1] - off[self.rank]
            assert n == roff[p + 1] - roff[p]
            if n:
                self.ctx.dev_copy(recv + roff[p], ptr + off[self.rank], n)
        self.sh.barrier.wait()          # nobody frees its send buffer before everybody has copied

    def all_gather_dev(self, send, nbytes, recv, roff):
        parts = self._gather((send, nbytes))
        for p, (ptr, n) in enumerate(parts):
            assert n == roff[p + 1] - roff[p]
            if n:
                self.ctx.dev_copy(recv + roff[p], ptr, n)
        self.sh.barrier.wait()


class GlooTransport:
    """ranks = processes (torch.distributed, gloo), device memory staged through the host with hipMemcpy"""

    def __init__(self, dist, rank, world):
        import ctypes
        self.dist, self.rank, self.world, self.error = dist, rank, world, None
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]

    def _down(self, ptr, n):
        a = np.empty(max(n, 1), np.uint8)
        if n:
            assert self.hip.hipMemcpy(a.ctypes.data, ptr, n, 2) == 0        # hipMemcpyDeviceToHost
        return a[:n]

    def _up(self, ptr, a):
        a = np.ascontiguousarray(a, np.uint8)
        if a.size:
            assert self.hip.hipMemcpy(ptr, a.ctypes.data, a.size, 1) == 0    # hipMemcpyHostToDevice

    def _gather_objects(self, x):
        out = [None] * self.world
        self.dist.all_gather_object(out, x)
        return out

    def all_gather_host(self, b):
        return self._gather_objects(bytes(b))

    def all_to_all_dev(self, send, soff, recv, roff):
        mine = self._down(send, soff[-1])
        parts = self._gather_objects((mine.tobytes(), soff))
        for p, (buf, off) in enumerate(parts):
            piece = np.frombuffer(buf, np.uint8)[off[self.rank]: off[self.rank + 1]]
            assert piece.size == roff[p + 1] - roff[p]
            self._up(recv + roff[p], piece)

    def all_gather_dev(self, send, nbytes, recv, roff):
        parts = self._gather_objects(self._down(send, nbytes).tobytes())
        for p, buf in enumerate(parts):
            assert len(buf) == roff[p + 1] - roff[p]
            self._up(recv + roff[p], np.frombuffer(buf, np.uint8))
