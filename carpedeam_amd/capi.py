"""ctypes binding of the C ABI (include/carpedeam_hip.h) -- used by tests and bench.py.

There is no CPU fallback: importing works anywhere (so that `-m "not gpu"` tests can check that the library loads and
exports its symbols), but every compute entry point needs a gfx950 device and raises CdmError otherwise.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CDM_LIB", os.path.join(HERE, "libcarpedeam_hip.so"))      # CDM_LIB: another build of the library (bisecting)

HIT_DTYPE = np.dtype([("target", "<u4"), ("score", "<i4"), ("diagonal", "<i4")])
ALN_DTYPE = np.dtype([("target", "<u4"), ("raw_score", "<i4"), ("ident", "<i4"), ("q_start", "<i4"), ("q_end", "<i4"),
                      ("db_start", "<i4"), ("db_end", "<i4"), ("seq_id", "<f4")])


class KmerParams(C.Structure):
    _fields_ = [("kmer_size", C.c_int32), ("kmers_per_seq", C.c_int32), ("kmers_per_seq_scale", C.c_float), ("hash_shift", C.c_uint64),
                ("ignore_multi_kmer", C.c_int32), ("include_only_extendable", C.c_int32), ("cov_mode", C.c_int32), ("cov_thr", C.c_float)]

    @classmethod
    def reads_default(cls):
        return cls(20, 200, 0.2, 67, 1, 0, 1, 0.0)


class RescoreParams(C.Structure):
    _fields_ = [("seq_id_thr", C.c_float), ("eval_thr", C.c_double), ("cov_mode", C.c_int32), ("cov_thr", C.c_float),
                ("seq_id_mode", C.c_int32), ("min_aln_len", C.c_int32)]

    @classmethod
    def default(cls):
        return cls(0.9, 0.001, 1, 0.0, 0, 0)


class HammingParams(C.Structure):
    _fields_ = [("seq_id_thr", C.c_float), ("eval_thr", C.c_double), ("cov_mode", C.c_int32), ("cov_thr", C.c_float),
                ("seq_id_mode", C.c_int32), ("min_aln_len", C.c_int32), ("reverse_prefilter", C.c_int32)]

    @classmethod
    def linclust(cls):
        """what `ancient_assemble` passes to linclust's pre-clustering (GuidedNuclassembler.cpp:176-181, Linclust.cpp:103-112)"""
        return cls(0.97, 0.001, 1, 0.99, 0, 0, 1)


class AncientParams(C.Structure):
    _fields_ = [("seq_id_thr", C.c_float), ("corr_reads_ry_seq_id", C.c_float), ("ry_seq_id_thr", C.c_float), ("rand_align_penal", C.c_float),
                ("excess_penal", C.c_float), ("likelihood_threshold", C.c_float), ("unsafe", C.c_int32), ("min_cov_safe", C.c_int32),
                ("max_seq_len", C.c_uint64)]

    @classmethod
    def default(cls):
        return cls(0.9, 0.99, 0.99, 0.85, 0.0625, 0.5, 0, 5, 200000)


EXPORTS = [
    "cdm_last_error", "cdm_ctx_create", "cdm_ctx_destroy", "cdm_ctx_sync", "cdm_ctx_stream", "cdm_ctx_last_kernel_ms",
    "cdm_seqdb_upload", "cdm_seqdb_synth", "cdm_seqdb_size", "cdm_seqdb_residues", "cdm_seqdb_max_len", "cdm_seqdb_meta",
    "cdm_seqdb_download", "cdm_seqdb_download_stream", "cdm_seqdb_free", "cdm_seqdb_select_ext", "cdm_seqdb_words", "cdm_seqdb_copy_packed", "cdm_seqdb_from_packed", "cdm_damage_load", "cdm_damage_get", "cdm_kmermatch", "cdm_hits_upload", "cdm_hits_count",
    "cdm_hits_download", "cdm_hits_free", "cdm_rescore", "cdm_alns_upload", "cdm_alns_count", "cdm_alns_download", "cdm_alns_free",
    "cdm_evalue", "cdm_bit_score", "cdm_gapped_evalue", "cdm_correct", "cdm_extend",
    "cdm_kmermatch_part", "cdm_kmermatch_split_begin", "cdm_kpart_outgoing", "cdm_kmermatch_split_finish", "cdm_kpart_info", "cdm_kpart_stale", "cdm_kpart_gather", "cdm_kpart_sort", "cdm_kpart_vote", "cdm_kpart_cont_cap", "cdm_kpart_free", "cdm_dev_copy",
    "cdm_seqdb_from_packed_ext", "cdm_seqdb_copy_ext", "cdm_seqdb_export_packed", "cdm_seqdb_import_packed", "cdm_contig_merge", "cdm_cyclecheck", "cdm_seqdb_has_raw", "cdm_seqdb_copy_raw", "cdm_seqdb_attach_raw",
    "cdm_rescore_hamming", "cdm_pool_headroom", "cdm_pool_stats", "cdm_env_refresh",
    "cdm_comm_unique_id", "cdm_comm_create_rccl", "cdm_comm_create_ops", "cdm_comm_free", "cdm_comm_rank", "cdm_comm_world", "cdm_kmermatch_dist",
    "cdm_seqdb_allgather_owned", "cdm_reads_iteration_dist", "cdm_contig_iteration_dist", "cdm_comm_owned", "cdm_comm_last_path", "cdm_kpart_gather_at", "cdm_comm_standin_group", "cdm_comm_create_standin", "cdm_kpart_set_range",
]


AG_HOST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)
A2A_DEV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p)
AG_DEV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p)


class CommOps(C.Structure):
    """cdm_comm_ops (include/carpedeam_hip.h): the collectives of a caller-supplied transport"""
    _fields_ = [("user", C.c_void_p), ("all_gather_host", AG_HOST_FN), ("all_to_all_dev", A2A_DEV_FN), ("all_gather_dev", AG_DEV_FN)]


class CdmError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libcarpedeam_hip.so (built in-tree by carpedeam_amd.build); fail loudly when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CdmError("libcarpedeam_hip.so not built: run `python -m carpedeam_amd.build` (no CPU fallback exists)")
        l = C.CDLL(LIB_PATH)
        vp, u64p, u32p, u8p = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
        l.cdm_last_error.restype = C.c_char_p
        l.cdm_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
        l.cdm_ctx_destroy.argtypes = [vp]
        l.cdm_ctx_destroy.restype = None
        l.cdm_ctx_sync.argtypes = [vp]
        l.cdm_ctx_stream.argtypes = [vp]
        l.cdm_ctx_stream.restype = vp
        l.cdm_ctx_last_kernel_ms.argtypes = [vp, C.c_int]
        l.cdm_ctx_last_kernel_ms.restype = C.c_float
        l.cdm_seqdb_upload.argtypes = [vp, vp, vp, vp, vp, vp, C.c_uint64, C.POINTER(vp)]
        l.cdm_seqdb_synth.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(vp)]
        for f in (l.cdm_seqdb_size, l.cdm_seqdb_residues, l.cdm_hits_count, l.cdm_alns_count):
            f.argtypes = [vp]
            f.restype = C.c_uint64
        l.cdm_seqdb_max_len.argtypes = [vp]
        l.cdm_seqdb_max_len.restype = C.c_uint32
        l.cdm_seqdb_meta.argtypes = [vp, vp, vp, vp, vp]
        l.cdm_seqdb_download.argtypes = [vp, vp, vp, vp]
        for f in (l.cdm_seqdb_free, l.cdm_hits_free, l.cdm_alns_free):
            f.argtypes = [vp]
            f.restype = None
        l.cdm_seqdb_select_ext.argtypes = [vp, vp, C.POINTER(vp)]
        l.cdm_seqdb_words.argtypes = [vp]
        l.cdm_seqdb_words.restype = C.c_uint64
        l.cdm_seqdb_copy_packed.argtypes = [vp, vp, vp, vp, vp, vp]
        l.cdm_seqdb_has_raw.argtypes = [vp]
        l.cdm_seqdb_copy_raw.argtypes = [vp, vp, vp, vp]
        l.cdm_seqdb_attach_raw.argtypes = [vp, vp, vp, vp]
        l.cdm_seqdb_from_packed.argtypes = [vp, vp, vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint8, C.POINTER(vp)]
        l.cdm_damage_load.argtypes = [vp, C.c_char_p]
        l.cdm_damage_get.argtypes = [vp, vp]
        l.cdm_kmermatch.argtypes = [vp, vp, C.POINTER(KmerParams), C.POINTER(vp)]
        l.cdm_hits_upload.argtypes = [vp, vp, vp, vp, C.POINTER(vp)]
        l.cdm_hits_download.argtypes = [vp, vp, vp, vp]
        l.cdm_rescore.argtypes = [vp, vp, vp, C.POINTER(RescoreParams), C.POINTER(vp)]
        l.cdm_rescore_hamming.argtypes = [vp, vp, vp, C.POINTER(HammingParams), C.POINTER(vp)]
        l.cdm_pool_headroom.argtypes = [C.c_float]
        l.cdm_pool_headroom.restype = None
        l.cdm_alns_upload.argtypes = [vp, vp, vp, vp, C.POINTER(vp)]
        l.cdm_alns_download.argtypes = [vp, vp, vp, vp]
        l.cdm_evalue.argtypes = [C.c_double, C.c_double, C.c_uint64]
        l.cdm_evalue.restype = C.c_double
        l.cdm_bit_score.argtypes = [C.c_double]
        l.cdm_correct.argtypes = [vp, vp, vp, C.POINTER(AncientParams), C.POINTER(vp)]
        l.cdm_extend.argtypes = [vp, vp, vp, C.POINTER(AncientParams), C.POINTER(vp), vp]
        l.cdm_kmermatch_part.argtypes = [vp, vp, C.POINTER(KmerParams), C.c_int, C.c_int, C.POINTER(vp)]
        if hasattr(l, "cdm_kmermatch_split_begin"):
            l.cdm_kmermatch_split_begin.argtypes = [vp, vp, C.POINTER(KmerParams), C.c_int, C.c_int, C.POINTER(vp)]
            l.cdm_kpart_outgoing.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint64)]
            l.cdm_kmermatch_split_finish.argtypes = [vp, vp, vp, vp, C.c_uint64, vp, vp, C.c_uint64, C.c_int]
        l.cdm_kpart_info.argtypes = [vp, vp]
        l.cdm_kpart_stale.argtypes = [vp, vp, C.c_uint64, vp]
        l.cdm_kpart_gather.argtypes = [vp, vp, C.c_int, vp, C.POINTER(vp)]
        l.cdm_kpart_sort.argtypes = [vp, vp, vp, C.c_uint64, vp, vp]
        l.cdm_kpart_vote.argtypes = [vp, vp, vp, vp, C.POINTER(vp)]
        l.cdm_kpart_free.argtypes = [vp]
        l.cdm_kpart_free.restype = None
        l.cdm_dev_copy.argtypes = [vp, vp, vp, C.c_uint64]
        l.cdm_seqdb_from_packed_ext.argtypes = [vp, vp, vp, vp, vp, vp, C.c_uint64, C.c_uint64, C.POINTER(vp)]
        l.cdm_seqdb_copy_ext.argtypes = [vp, vp, vp]
        l.cdm_contig_merge.argtypes = [vp, vp, vp, C.POINTER(AncientParams), C.c_float, C.POINTER(vp)]
        l.cdm_cyclecheck.argtypes = [vp, vp, C.c_uint32, C.c_int, C.POINTER(vp), C.POINTER(vp), vp]
        if hasattr(l, "cdm_comm_create_ops"):
            l.cdm_comm_unique_id.argtypes = [vp]
            l.cdm_comm_create_rccl.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
            l.cdm_comm_create_ops.argtypes = [vp, C.c_int, C.c_int, C.POINTER(CommOps), C.POINTER(vp)]
            l.cdm_comm_free.argtypes = [vp]
            l.cdm_comm_free.restype = None
            l.cdm_kmermatch_dist.argtypes = [vp, vp, vp, C.POINTER(KmerParams), C.POINTER(vp)]
            l.cdm_seqdb_allgather_owned.argtypes = [vp, vp, vp, C.POINTER(vp)]
            if hasattr(l, "cdm_comm_owned"):
                l.cdm_comm_owned.argtypes = [vp, C.c_uint64, vp]
                l.cdm_comm_world.argtypes = [vp]
            l.cdm_reads_iteration_dist.argtypes = [vp, vp, vp, C.POINTER(KmerParams), C.POINTER(RescoreParams), C.POINTER(AncientParams), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
            l.cdm_contig_iteration_dist.argtypes = [vp, vp, vp, C.POINTER(KmerParams), C.POINTER(RescoreParams), C.POINTER(AncientParams), C.c_float, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
        try:
            l.cdm_env_refresh.restype = None
        except AttributeError:          # (CDM_LIB names an older build of the library - bisecting: it reads its switches with getenv)
            pass
        _lib = l
    # The library snapshots its CDM_* switches once per process (no getenv on its call paths, csrc/pool.h).  Tests and A/B runs change
    # them between calls of one process: when os.environ's CDM_* entries differ from what the library last saw, it reads them again.
    global _env_seen
    env = {k: v for k, v in os.environ.items() if k.startswith("CDM_")}
    if env != _env_seen:
        _env_seen = env
        if hasattr(_lib, "cdm_env_refresh"):
            _lib.cdm_env_refresh()
    return _lib


_env_seen = None


def pool_stats():
    """the device-memory allocator's counters (cdm_pool_stats)"""
    a = np.zeros(8, np.uint64)
    lib().cdm_pool_stats(a.ctypes.data_as(C.c_void_p))
    return {"requests": int(a[0]), "served_without_driver": int(a[1]), "driver_calls": int(a[2]), "driver_bytes": int(a[3]), "driver_seconds": a[4] / 1e9, "trims": int(a[5]),
            "mapped_bytes": int(a[6]), "in_use_bytes": int(a[7])}


def _check(rc):
    if rc != 0:
        raise CdmError("cdm error %d: %s" % (rc, lib().cdm_last_error().decode()))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class SeqDb:
    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:      # (at interpreter shutdown the module's globals are gone already)
            lib().cdm_seqdb_free(self.h)
            self.h = None

    @property
    def n(self):
        return int(lib().cdm_seqdb_size(self.h))

    @property
    def residues(self):
        return int(lib().cdm_seqdb_residues(self.h))

    @property
    def words(self):
        return int(lib().cdm_seqdb_words(self.h))

    def select_ext(self):
        """the contigs (wasExtended == 1) as a new device DB"""
        h = C.c_void_p()
        _check(lib().cdm_seqdb_select_ext(self.ctx.h, self.h, C.byref(h)))
        return SeqDb(self.ctx, h)

    def copy_ext(self, ext_ptr):
        _check(lib().cdm_seqdb_copy_ext(self.ctx.h, self.h, ext_ptr))

    def copy_packed(self, codes_ptr, nmask_ptr, len_ptr, key_ptr):
        """copy the packed form into DEVICE buffers (raw pointers, e.g. torch tensor .data_ptr())"""
        _check(lib().cdm_seqdb_copy_packed(self.ctx.h, self.h, codes_ptr, nmask_ptr, len_ptr, key_ptr))

    @property
    def has_raw(self):
        """the DB carries letters beyond ACGTN (their original bytes travel beside the packed form: copy_raw / attach_raw)"""
        return bool(lib().cdm_seqdb_has_raw(self.h))

    def copy_raw(self, raw_ptr, flags_ptr):
        _check(lib().cdm_seqdb_copy_raw(self.ctx.h, self.h, raw_ptr, flags_ptr))

    def attach_raw(self, raw_ptr, flags_ptr):
        _check(lib().cdm_seqdb_attach_raw(self.ctx.h, self.h, raw_ptr, flags_ptr))

    def meta(self):
        n = self.n
        lens, keys, ext = np.empty(n, np.uint32), np.empty(n, np.uint32), np.empty(n, np.uint8)
        _check(lib().cdm_seqdb_meta(self.ctx.h, self.h, _ptr(lens), _ptr(keys), _ptr(ext)))
        return lens, keys, ext

    def download_into(self, buf, offs):
        """sequence i as "SEQ\\n" at buf[offs[i]:] (uint8 / uint64 numpy arrays owned by the caller; the library writes the whole range up to the last entry's end: bytes between entries come out as NUL)"""
        _check(lib().cdm_seqdb_download(self.ctx.h, self.h, _ptr(buf), _ptr(offs)))

    def download(self):
        """-> (list of bytes sequences, keys, ext)"""
        if self.n == 0:
            return [], np.zeros(0, np.uint32), np.zeros(0, np.uint8)
        lens, keys, ext = self.meta()
        offs = np.zeros(self.n, np.uint64)
        offs[1:] = np.cumsum(lens[:-1].astype(np.uint64) + 1)
        total = int(lens.astype(np.uint64).sum() + self.n)
        buf = np.empty(total, np.uint8)
        _check(lib().cdm_seqdb_download(self.ctx.h, self.h, _ptr(buf), _ptr(offs)))
        raw = buf.tobytes()
        seqs = [raw[int(o):int(o) + int(l)] for o, l in zip(offs, lens)]
        return seqs, keys, ext


class _Csr:
    free = None
    count_fn = None
    download_fn = None
    dtype = None

    def __init__(self, ctx, handle, n):
        self.ctx, self.h, self.n = ctx, handle, n

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            getattr(lib(), self.free)(self.h)
            self.h = None

    @property
    def count(self):
        return int(getattr(lib(), self.count_fn)(self.h))

    def download(self):
        off = np.empty(self.n + 1, np.uint64)
        rec = np.empty(self.count, self.dtype)
        _check(getattr(lib(), self.download_fn)(self.ctx.h, self.h, _ptr(off), _ptr(rec)))
        return off, rec


class Hits(_Csr):
    free, count_fn, download_fn, dtype = "cdm_hits_free", "cdm_hits_count", "cdm_hits_download", HIT_DTYPE


class Alns(_Csr):
    free, count_fn, download_fn, dtype = "cdm_alns_free", "cdm_alns_count", "cdm_alns_download", ALN_DTYPE


KPART_SLICES = 256          # CDM_KPART_SLICES


class KPart:
    """One k-mer range of a split kmermatcher run (cdm_kpart)."""

    def __init__(self, ctx, handle, db):
        self.ctx, self.h, self.db = ctx, handle, db

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib().cdm_kpart_free(self.h)
            self.h = None

    def info(self):
        a = np.zeros(4, np.uint64)
        _check(lib().cdm_kpart_info(self.h, _ptr(a)))
        return {"real": int(a[0]), "kept": int(a[1]), "any_below": bool(a[2]), "n": int(a[3])}

    def outgoing(self):
        """after Ctx.kmermatch_split_begin -> (offsets[KPART_SLICES + 1], keys ptr, values ptr, bytes per value, hash keys ptr, hash values
        ptr, hash tuples): this rank's tuples ordered by fine slices of the k-mer space (the ranks' ranges are runs of slices)"""
        off = np.zeros(KPART_SLICES + 1, np.uint64)
        k, v, hk, hv = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        vb, nh = C.c_int(), C.c_uint64()
        _check(lib().cdm_kpart_outgoing(self.h, _ptr(off), C.byref(k), C.byref(v), C.byref(vb), C.byref(hk), C.byref(hv), C.byref(nh)))
        return off, k.value or 0, v.value or 0, vb.value, hk.value or 0, hv.value or 0, nh.value

    def split_finish(self, keys_ptr, vals_ptr, m, hash_keys_ptr, hash_vals_ptr, n_hash, below):
        _check(lib().cdm_kmermatch_split_finish(self.ctx.h, self.h, keys_ptr, vals_ptr, int(m), hash_keys_ptr, hash_vals_ptr, int(n_hash), int(bool(below))))

    def stale(self, j):
        a = np.zeros(67, np.uint32)
        _check(lib().cdm_kpart_stale(self.ctx.h, self.h, int(j), _ptr(a)))
        return a

    def gather(self, nranks):
        """-> (offsets[nranks + 1], device pointer of the group keys grouped by representative)"""
        off = np.zeros(nranks + 1, np.uint64)
        p = C.c_void_p()
        _check(lib().cdm_kpart_gather(self.ctx.h, self.h, nranks, _ptr(off), C.byref(p)))
        return off, p.value or 0

    def sort(self, keys_ptr, n_keys):
        """sort 2 on the received keys -> (head list, sorted tuples, target id of the last one)"""
        head = np.zeros(lib().cdm_kpart_cont_cap() + 3, np.uint32)
        info = np.zeros(2, np.uint64)
        _check(lib().cdm_kpart_sort(self.ctx.h, self.h, keys_ptr, int(n_keys), _ptr(head), _ptr(info)))
        return head, int(info[0]), int(info[1])

    def vote(self, cont, stale):
        st = np.ascontiguousarray(stale[:65], np.uint32)
        ct = None if cont is None else np.ascontiguousarray(cont, np.uint32)
        h = C.c_void_p()
        _check(lib().cdm_kpart_vote(self.ctx.h, self.h, _ptr(ct), _ptr(st), C.byref(h)))
        return Hits(self.ctx, h, self.db.n)


class Ctx:
    def __init__(self, device=0):
        h = C.c_void_p()
        _check(lib().cdm_ctx_create(device, C.byref(h)))
        self.h = h

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib().cdm_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        _check(lib().cdm_ctx_sync(self.h))

    def last_kernel_ms(self, which):
        return float(lib().cdm_ctx_last_kernel_ms(self.h, which))

    def damage_load(self, prefix):
        _check(lib().cdm_damage_load(self.h, prefix.encode()))

    def damage_get(self):
        out = np.empty(352, np.longdouble)
        _check(lib().cdm_damage_get(self.h, _ptr(out)))
        return out.reshape(2, 11, 4, 4)

    # ---- sequence DB
    def upload_seqs(self, seqs, keys=None, ext=None):
        """seqs: list of bytes/str; builds the 'SEQ\\n\\0' data blob the way the DB data file holds it."""
        bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
        n = len(bs)
        lens = np.array([len(b) for b in bs], np.uint32)
        offs = np.zeros(n, np.uint64)
        offs[1:] = np.cumsum(lens[:-1].astype(np.uint64) + 2)
        data = np.frombuffer(b"".join(b + b"\n\0" for b in bs), np.uint8)
        keys = np.arange(n, dtype=np.uint32) if keys is None else np.asarray(keys, np.uint32)
        ext = None if ext is None else np.asarray(ext, np.uint8)
        h = C.c_void_p()
        _check(lib().cdm_seqdb_upload(self.h, _ptr(data), _ptr(offs), _ptr(lens), _ptr(keys), _ptr(ext), n, C.byref(h)))
        return SeqDb(self, h)

    def upload_keyed_seqdb(self, keyed):
        """keyed: dict key -> (payload incl. newline, ext) as mmdb.read_db / load_keyed give it."""
        ks = sorted(keyed)
        return self.upload_seqs([keyed[k][0].rstrip(b"\n") for k in ks], ks, [keyed[k][1] for k in ks])

    def synth(self, n, lo, hi, seed, n_total=None, first=0):
        h = C.c_void_p()
        _check(lib().cdm_seqdb_synth(self.h, n if n_total is None else n_total, first, n, lo, hi, seed, C.byref(h)))
        return SeqDb(self, h)

    def from_packed(self, codes_ptr, nmask_ptr, len_ptr, key_ptr, n, words, ext_value=1):
        h = C.c_void_p()
        _check(lib().cdm_seqdb_from_packed(self.h, codes_ptr, nmask_ptr, len_ptr, key_ptr, n, words, ext_value, C.byref(h)))
        return SeqDb(self, h)

    def contig_merge(self, db, alns, par=None, merge_seq_id=0.99):
        par = par or AncientParams.default()
        h = C.c_void_p()
        _check(lib().cdm_contig_merge(self.h, db.h, alns.h, C.byref(par), merge_seq_id, C.byref(h)))
        return SeqDb(self, h)

    def cyclecheck(self, db, max_seq_len=65535, chop_cycle=False):
        """(cyclic contigs [cut at the split diagonal if chop_cycle], the other contigs, split diagonal per contig)"""
        import numpy as np
        c, r = C.c_void_p(), C.c_void_p()
        split = np.zeros(max(db.n, 1), dtype=np.uint32)
        _check(lib().cdm_cyclecheck(self.h, db.h, max_seq_len, 1 if chop_cycle else 0, C.byref(c), C.byref(r), split.ctypes.data))
        return SeqDb(self, c), SeqDb(self, r), split[:db.n]

    def from_packed_ext(self, codes_ptr, nmask_ptr, len_ptr, key_ptr, ext_ptr, n, words):
        h = C.c_void_p()
        _check(lib().cdm_seqdb_from_packed_ext(self.h, codes_ptr, nmask_ptr, len_ptr, key_ptr, ext_ptr, n, words, C.byref(h)))
        return SeqDb(self, h)

    def dev_copy(self, dst_ptr, src_ptr, nbytes):
        _check(lib().cdm_dev_copy(self.h, dst_ptr, src_ptr, nbytes))

    def kmermatch_part(self, db, part, nparts, par=None):
        """phase A of kmermatcher on k-mer range part/nparts (multi-GPU runs, carpedeam_amd/shard.py)"""
        par = par or KmerParams.reads_default()
        h = C.c_void_p()
        _check(lib().cdm_kmermatch_part(self.h, db.h, C.byref(par), part, nparts, C.byref(h)))
        return KPart(self, h, db)

    def kmermatch_split_begin(self, db, rank, nranks, par=None):
        """the same first half with the extraction split by reads: this rank's block of the sequences, the tuples ordered by the k-mer
        range they go to (KPart.outgoing, then KPart.split_finish on what arrived)"""
        par = par or KmerParams.reads_default()
        h = C.c_void_p()
        _check(lib().cdm_kmermatch_split_begin(self.h, db.h, C.byref(par), rank, nranks, C.byref(h)))
        return KPart(self, h, db)

    # ---- containers
    def upload_hits(self, db, off, rec):
        off = np.ascontiguousarray(off, np.uint64)
        rec = np.ascontiguousarray(rec, HIT_DTYPE)
        h = C.c_void_p()
        _check(lib().cdm_hits_upload(self.h, db.h, _ptr(off), _ptr(rec), C.byref(h)))
        return Hits(self, h, db.n)

    def upload_alns(self, db, off, rec):
        off = np.ascontiguousarray(off, np.uint64)
        rec = np.ascontiguousarray(rec, ALN_DTYPE)
        h = C.c_void_p()
        _check(lib().cdm_alns_upload(self.h, db.h, _ptr(off), _ptr(rec), C.byref(h)))
        return Alns(self, h, db.n)

    # ---- stages
    def kmermatch(self, db, par=None):
        par = par or KmerParams.reads_default()
        h = C.c_void_p()
        _check(lib().cdm_kmermatch(self.h, db.h, C.byref(par), C.byref(h)))
        return Hits(self, h, db.n)

    def rescore(self, db, hits, par=None):
        par = par or RescoreParams.default()
        h = C.c_void_p()
        _check(lib().cdm_rescore(self.h, db.h, hits.h, C.byref(par), C.byref(h)))
        return Alns(self, h, db.n)

    def rescore_hamming(self, db, hits, par=None):
        par = par or HammingParams.linclust()
        h = C.c_void_p()
        _check(lib().cdm_rescore_hamming(self.h, db.h, hits.h, C.byref(par), C.byref(h)))
        return Hits(self, h, db.n)

    def correct(self, db, alns, par=None):
        par = par or AncientParams.default()
        h = C.c_void_p()
        _check(lib().cdm_correct(self.h, db.h, alns.h, C.byref(par), C.byref(h)))
        return SeqDb(self, h)

    def extend(self, db, alns, par=None, want_scores=False):
        par = par or AncientParams.default()
        h = C.c_void_p()
        scores = np.full(alns.count, np.nan, np.float64) if want_scores else None
        _check(lib().cdm_extend(self.h, db.h, alns.h, C.byref(par), C.byref(h), _ptr(scores)))
        out = SeqDb(self, h)
        return (out, scores) if want_scores else out


class Comm:
    """cdm_comm: a rank's communicator for the library's own multi-GPU calls (csrc/dist.hip)."""

    def __init__(self, ctx, handle, rank, world, keep=None):
        self.ctx, self.h, self.rank, self.world, self._keep = ctx, handle, rank, world, keep

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib().cdm_comm_free(self.h)
            self.h = None

    @staticmethod
    def unique_id():
        """rank 0: the 128 bytes every rank's Comm.rccl() needs (ncclGetUniqueId)"""
        b = C.create_string_buffer(128)
        _check(lib().cdm_comm_unique_id(b))
        return b.raw

    @classmethod
    def rccl(cls, ctx, rank, world, unique_id):
        """RCCL, called by the library itself (one device per rank)"""
        h = C.c_void_p()
        _check(lib().cdm_comm_create_rccl(ctx.h, rank, world, C.create_string_buffer(bytes(unique_id), 128), C.byref(h)))
        return cls(ctx, h, rank, world)

    @staticmethod
    def standin_group(world):
        """shared by the ranks (threads) of one test: the in-process stand-in for RCCL's point-to-point calls"""
        lib().cdm_comm_standin_group.restype = C.c_void_p
        lib().cdm_comm_standin_group.argtypes = [C.c_int]
        return lib().cdm_comm_standin_group(world)

    @classmethod
    def standin(cls, ctx, group, rank, world):
        """the RCCL transport's code over the stand-in (several ranks as threads on ONE device)"""
        lib().cdm_comm_create_standin.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        h = C.c_void_p()
        _check(lib().cdm_comm_create_standin(ctx.h, group, rank, C.byref(h)))
        return cls(ctx, h, rank, world)

    @classmethod
    def from_transport(cls, ctx, rank, world, t):
        """A transport written in Python (tests: several ranks on ONE device): t.all_gather_host(bytes) -> list of the ranks' bytes;
        t.all_to_all_dev(send_ptr, send_off, recv_ptr, recv_off) and t.all_gather_dev(send_ptr, send_bytes, recv_ptr, recv_off) move
        device memory (byte offsets, world + 1 of them) and return when the data has arrived."""
        def ag_host(user, send, recv, nbytes):
            try:
                parts = t.all_gather_host(C.string_at(send, nbytes))
                for p, b in enumerate(parts):
                    C.memmove(recv + p * nbytes, b, nbytes)
                return 0
            except BaseException as e:      # noqa: BLE001 - an exception must not cross the C frames
                t.error = e
                return -1

        def a2a(user, send, soff, recv, roff, stream):
            try:
                ctx.sync()
                t.all_to_all_dev(send or 0, [int(soff[i]) for i in range(world + 1)], recv or 0, [int(roff[i]) for i in range(world + 1)])
                return 0
            except BaseException as e:      # noqa: BLE001
                t.error = e
                return -1

        def ag_dev(user, send, nbytes, recv, roff, stream):
            try:
                ctx.sync()
                t.all_gather_dev(send or 0, int(nbytes), recv or 0, [int(roff[i]) for i in range(world + 1)])
                return 0
            except BaseException as e:      # noqa: BLE001
                t.error = e
                return -1

        ops = CommOps(None, AG_HOST_FN(ag_host), A2A_DEV_FN(a2a), AG_DEV_FN(ag_dev))
        h = C.c_void_p()
        _check(lib().cdm_comm_create_ops(ctx.h, rank, world, C.byref(ops), C.byref(h)))
        return cls(ctx, h, rank, world, keep=(ops, t))

    def kmermatch(self, db, par=None):
        """kmermatcher over the ranks: this rank's share of the hits (its representatives' rows; self hits elsewhere)"""
        par = par or KmerParams.reads_default()
        h = C.c_void_p()
        _check(lib().cdm_kmermatch_dist(self.ctx.h, self.h, db.h, C.byref(par), C.byref(h)))
        return Hits(self.ctx, h, db.n)

    def allgather_owned(self, db_local):
        h = C.c_void_p()
        _check(lib().cdm_seqdb_allgather_owned(self.ctx.h, self.h, db_local.h, C.byref(h)))
        return SeqDb(self.ctx, h)

    def owned(self, n):
        """bounds[world + 1]: rank r owns the sequences [bounds[r], bounds[r + 1]) of a DB of n sequences (cdm_comm_owned: as this
        communicator's last kmermatch on such a DB cut them - equal shares of the group keys -, equal id ranges before any)"""
        b = np.zeros(lib().cdm_comm_world(self.h) + 1, np.uint64)
        _check(lib().cdm_comm_owned(self.h, int(n), _ptr(b)))
        return b

    def last_path(self):
        """what the last kmermatch over the ranks did (cdm_comm_last_path)"""
        lib().cdm_comm_last_path.argtypes = [C.c_void_p]
        return {0: None, 1: "replicate", 2: "all", 3: "split", 4: "part", 5: "ranges"}[int(lib().cdm_comm_last_path(self.h))]

    def reads_iteration(self, db, kpar=None, rpar=None, apar=None):
        """one iteration of the reads loop over the ranks -> (hits, alns, corrected DB, next DB); the DBs are complete on every rank"""
        kpar, rpar, apar = kpar or KmerParams.reads_default(), rpar or RescoreParams.default(), apar or AncientParams.default()
        hh, ah, ch, nh = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().cdm_reads_iteration_dist(self.ctx.h, self.h, db.h, C.byref(kpar), C.byref(rpar), C.byref(apar), C.byref(hh), C.byref(ah), C.byref(ch), C.byref(nh)))
        return Hits(self.ctx, hh, db.n), Alns(self.ctx, ah, db.n), SeqDb(self.ctx, ch), SeqDb(self.ctx, nh)

    def contig_iteration(self, db, kpar, rpar=None, apar=None, merge_seq_id=0.99):
        """one iteration of the contig loop over the ranks, up to ancient_contig_merge -> (alns, corrected DB, merged DB); the DBs are complete on every rank"""
        rpar, apar = rpar or RescoreParams.default(), apar or AncientParams.default()
        ah, ch, nh = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().cdm_contig_iteration_dist(self.ctx.h, self.h, db.h, C.byref(kpar), C.byref(rpar), C.byref(apar), merge_seq_id, C.byref(ah), C.byref(ch), C.byref(nh)))
        return Alns(self.ctx, ah, db.n), SeqDb(self.ctx, ch), SeqDb(self.ctx, nh)


# ------------------------------------------------------------------------------------------------ text codecs (tests)
def parse_pref_db(keyed, keys):
    """Prefilter DB text (QueryMatcher.h:81-126) -> CSR (offsets, HIT_DTYPE) over the key-sorted sequence index."""
    idx = {int(k): i for i, k in enumerate(keys)}
    off = np.zeros(len(keys) + 1, np.uint64)
    recs = []
    for i, k in enumerate(keys):
        payload = keyed.get(int(k), (b"", 0))[0]
        for line in payload.decode().split("\n"):
            if not line:
                continue
            t, s, d = line.split("\t")
            recs.append((idx[int(t)], int(s), int(d)))
        off[i + 1] = len(recs)
    return off, np.array(recs, HIT_DTYPE) if recs else np.empty(0, HIT_DTYPE)


def hits_to_text(off, rec, keys):
    out = {}
    for i, k in enumerate(keys):
        lines = ["%d\t%d\t%d" % (keys[r["target"]], r["score"], r["diagonal"]) for r in rec[int(off[i]):int(off[i + 1])]]
        out[int(k)] = ("\n".join(lines) + "\n").encode() if lines else b""
    return out


def parse_aln_db(keyed, keys):
    """Alignment DB text (Matcher.cpp:274-404) -> CSR (offsets, ALN_DTYPE).  raw_score is re-derived from the bit score
    the way the reference's consumers do (correction.cpp:222); ident is not in the text (-1)."""
    idx = {int(k): i for i, k in enumerate(keys)}
    off = np.zeros(len(keys) + 1, np.uint64)
    recs = []
    ln2 = float(np.log(2.0))
    lam, logk = float.fromhex("0x1.4478764a1b24ap-1"), float(np.log(float.fromhex("0x1.a1c1e68ea2ab1p-2")))
    for i, k in enumerate(keys):
        payload = keyed.get(int(k), (b"", 0))[0]
        for line in payload.decode().split("\n"):
            if not line:
                continue
            f = line.split("\t")
            raw = int((logk + int(f[1]) * ln2) / lam + 0.5)
            recs.append((idx[int(f[0])], raw, -1, int(f[4]), int(f[5]), int(f[7]), int(f[8]), np.float32(float(f[2]))))
        off[i + 1] = len(recs)
    return off, np.array(recs, ALN_DTYPE) if recs else np.empty(0, ALN_DTYPE)


def fast_seqid_text(seq_id):
    """Util::fastSeqIdToBuffer + the tab that eats the last written char (Util.cpp:278-307, Matcher.cpp:362-363)."""
    s = np.float32(seq_id)
    if s == np.float32(1.0):
        return "1.00"
    out = "0."
    if s < np.float32(0.10):
        out += "0"
    if s < np.float32(0.01):
        out += "0"
    return out + str(int(np.float32(s * np.float32(1000))))


def alns_to_text(off, rec, keys, lens, db_residues):
    """CSR alignments -> alignment DB text as Matcher::resultToBuffer writes it."""
    l = lib()
    out = {}
    for i, k in enumerate(keys):
        lines = []
        for r in rec[int(off[i]):int(off[i + 1])]:
            qs, qe, ds, de = int(r["q_start"]), int(r["q_end"]), int(r["db_start"]), int(r["db_end"])
            aln_len = max(abs(qe - qs), abs(de - ds)) + 1
            sid = np.float32(np.float32(int(r["ident"])) / np.float32(aln_len)) if r["ident"] >= 0 else np.float32(r["seq_id"])
            ev = l.cdm_evalue(float(r["raw_score"]), float(lens[i]), db_residues)
            lines.append("%d\t%d\t%s\t%.3E\t%d\t%d\t%d\t%d\t%d\t%d" % (keys[r["target"]], l.cdm_bit_score(float(r["raw_score"])), fast_seqid_text(sid), ev,
                                                                    qs, qe, lens[i], ds, de, lens[r["target"]]))
        out[int(k)] = ("\n".join(lines) + "\n").encode() if lines else b""
    return out
