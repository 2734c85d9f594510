"""GPU parity of kmermatcher's 8-byte slot layout (round 5: DBs whose sequences all have one length go through sort 1 as slot keys /
slot tuples - radix.h sortSlotKeys, kmermatch.hip LayoutSlot): identical to the 12-byte layout and to the oracle, on every path the
layout touches - the three extraction kernels, the head and segment passes, the grouping kernel's decode, big buckets, the left-over scan."""
import numpy as np
import pytest

from carpedeam_amd import capi, mmdb, synth
from gpuutil import diff_keys, run_oracle
from stageflags import K_FLAGS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return capi.Ctx(0)


def text(ctx, seq_keyed, par=None):
    db = ctx.upload_keyed_seqdb(seq_keyed)
    _, keys, _ = db.meta()
    off, rec = ctx.kmermatch(db, par).download()
    return {k: (v, 0) for k, v in capi.hits_to_text(off, rec, keys).items()}


def strip_ext(db):
    return {k: (v[0], 0) for k, v in db.items()}


def uniform_reads(n, L, seed, genome_len=None, dup=0.0, lowc=0, with_n=0.0):
    """n reads of L letters from a random genome (both strands), some repeated verbatim, `lowc` low-complexity ones (tandem repeats:
    the repeated-k-mer path of the extractor), a fraction with N letters"""
    rng = np.random.default_rng(seed)
    letters = np.frombuffer(b"ACGT", np.uint8)
    G = genome_len or max(4 * L, n * L // 20)
    genome = rng.integers(0, 4, G + L)
    out = []
    for _ in range(n):
        st = int(rng.integers(0, G))
        c = genome[st:st + L].copy()
        if rng.random() < 0.5:
            c = (3 - c)[::-1]
        s = bytearray(letters[c].tobytes())
        if rng.random() < with_n:
            s[int(rng.integers(0, L))] = ord("N")
        out.append(s.decode())
    for i in range(int(dup * n)):
        out.append(out[int(rng.integers(0, n))])
    units = ["ACGTTGCA", "AC", "A", "ACGTACGTAC", "TTGCAACG", "ACGGT"]
    for i in range(lowc):
        u = units[i % len(units)]
        out.append((u * (L // len(u) + 1))[:L])
    order = rng.permutation(len(out))
    return [out[i] for i in order]


@pytest.mark.parametrize("L,n,kw", [(100, 4000, dict(dup=0.05, lowc=30, with_n=0.02)), (36, 3000, dict(dup=0.1, lowc=12)), (151, 2500, dict(lowc=10, with_n=0.05)),
                                     (20, 500, dict(dup=0.3)), (21, 800, dict(dup=0.2, lowc=6)), (250, 1200, dict(lowc=8))])
def test_slot_layout_equals_packed_layout_and_oracle(ctx, oracle_bin, tmp_path, monkeypatch, L, n, kw):
    seqs = uniform_reads(n, L, seed=L * 7 + n, **kw)
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in"), seqs)
    run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "4")
    want = strip_ext(mmdb.read_db(t("pref")))
    keyed = mmdb.read_db(t("in"))
    for layout in ("slot", "packed"):
        monkeypatch.setenv("CDM_KMER_LAYOUT", layout)
        assert not diff_keys(text(ctx, keyed), want), layout
    # the grouping kernel's capacities lowered: big buckets (gathered as pairs, sorted, scattered back as slot tuples), small owned ranges
    monkeypatch.setenv("CDM_KMER_LAYOUT", "slot")
    for cap in ("64", "5", "3,17", "1", "512,40"):
        monkeypatch.setenv("CDM_BUCKET_CAP", cap)
        assert not diff_keys(text(ctx, keyed), want), cap
    monkeypatch.delenv("CDM_BUCKET_CAP")
    # sort 2 and the vote on their other paths behind the slot layout
    # ... and the head pass's histogram: counted by the extraction kernels (default; "check" compares it with a count of the keys), or by sort 1
    for env in ({"CDM_KMER_VOTE": "tuples"}, {"CDM_KMER_SORT2": "check"}, {"CDM_AGG_D": "3"}, {"CDM_UNIT_CAP": "5", "CDM_BLOCK_CAP": "8"}, {"CDM_FORCE_WIDE_KEY": "1"},
                {"CDM_SLOT_HIST": "check"}, {"CDM_SLOT_HIST": "kernel"},
                # the run records: from the key array instead of the grouping kernel's stage; a stage of 2 records per wave (overflow: back to the key array)
                {"CDM_RUN_RECORDS": "kernel"}, {"CDM_REC_LIMIT": "2"}, {"CDM_REC_LIMIT": "2", "CDM_KMER_LAYOUT": "packed"}, {"CDM_BUCKET_CAP": "512,40", "CDM_REC_LIMIT": "9"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        assert not diff_keys(text(ctx, keyed), want), env
        for k in env:
            monkeypatch.delenv(k)
        monkeypatch.setenv("CDM_KMER_LAYOUT", "slot")
    monkeypatch.delenv("CDM_KMER_LAYOUT")


def test_slot_layout_contig_parameters_and_other_k(ctx, oracle_bin, tmp_path, monkeypatch):
    """-k 22 is beyond the slot tuple's 31 + 9 k-mer bits (the 12-byte layout takes the call), k = 14 .. 20 are inside; only extendable
    overlaps drop members in the middle of a k-mer run"""
    seqs = uniform_reads(3000, 90, seed=5, dup=0.05, lowc=10)
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in"), seqs)
    keyed = mmdb.read_db(t("in"))
    for k in (14, 17, 20, 22):
        flags = " ".join(K_FLAGS).replace("-k 20", "-k %d" % k).replace("--include-only-extendable 0", "--include-only-extendable 1").split()
        run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *flags, "--threads", "4")
        par = capi.KmerParams(k, 200, 0.2, 67, 1, 1, 1, 0.0)
        assert not diff_keys(text(ctx, keyed, par), strip_ext(mmdb.read_db(t("pref")))), k
        monkeypatch.setenv("CDM_KMER_LAYOUT", "slot")
        if k == 22:
            with pytest.raises(capi.CdmError):
                text(ctx, keyed, par)
        else:
            assert not diff_keys(text(ctx, keyed, par), strip_ext(mmdb.read_db(t("pref")))), k
        monkeypatch.delenv("CDM_KMER_LAYOUT")


def test_slot_layout_is_refused_for_mixed_lengths(ctx, monkeypatch):
    monkeypatch.setenv("CDM_KMER_LAYOUT", "slot")
    with pytest.raises(capi.CdmError):
        text(ctx, {0: (b"ACGTTGCAAGGCTTAACGGATCCGATTACAGGCATCGA\n", 0), 1: (b"ACGTTGCAAGGCTTAACGGATCCGATTACAGGCATC\n", 0)})
    monkeypatch.delenv("CDM_KMER_LAYOUT")


def test_slot_layout_tiny_uniform_databases_against_the_oracle(ctx, oracle_bin, tmp_path, monkeypatch):
    """the reference's quirks decide most records of a handful of reads (first-group strand rule, the scan that runs past the end into the
    left-over tuples): 80 random DBs of 2 .. 14 reads of one length, both layouts"""
    rng = np.random.default_rng(314)
    letters = np.frombuffer(b"ACGT", np.uint8)
    t = lambda s: str(tmp_path / s)
    for case in range(80):
        L = int(rng.integers(20, 60))
        genome = rng.integers(0, 4, 3 * L)
        seqs = []
        for _ in range(int(rng.integers(2, 15))):
            st = int(rng.integers(0, 2 * L))
            c = genome[st:st + L].copy()
            if rng.random() < 0.5:
                c = (3 - c)[::-1]
            seqs.append(letters[c].tobytes().decode())
        if rng.random() < 0.3:
            seqs.append(seqs[0])
        mmdb.write_seqdb(t("in"), seqs)
        run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "1")
        want = strip_ext(mmdb.read_db(t("pref")))
        monkeypatch.setenv("CDM_SLOT_HIST", "check")
        for layout in ("slot", "packed"):
            monkeypatch.setenv("CDM_KMER_LAYOUT", layout)
            bad = diff_keys(text(ctx, mmdb.read_db(t("in"))), want)
            assert not bad, (case, layout, seqs, bad)
        monkeypatch.delenv("CDM_KMER_LAYOUT"); monkeypatch.delenv("CDM_SLOT_HIST")


def test_slot_layout_at_scale_equals_packed(ctx, monkeypatch):
    """2 M synthetic reads of 100 letters (the bench's corpus in small): identical hit arrays from both layouts, and with the grouping
    kernel's capacity lowered so that thousands of buckets take the big-bucket path"""
    db = ctx.synth(2_000_000, 100, 100, 7)
    ref = None
    for env in ({"CDM_KMER_LAYOUT": "packed"}, {"CDM_KMER_LAYOUT": "slot"}, {}, {"CDM_KMER_LAYOUT": "slot", "CDM_BUCKET_CAP": "40"}, {"CDM_SLOT_HIST": "check"}, {"CDM_SLOT_HIST": "kernel"},
                # run records: a wave's stage of 7 (a few go to the overflow list, merged by start), of 0 (the list overflows: from the key array)
                {"CDM_REC_LIMIT": "7"}, {"CDM_REC_LIMIT": "8", "CDM_BUCKET_CAP": "40"}, {"CDM_REC_LIMIT": "0"}, {"CDM_RUN_RECORDS": "kernel"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        off, rec = ctx.kmermatch(db).download()
        for k in env:
            monkeypatch.delenv(k)
        if ref is None:
            ref = (off, rec)
            assert len(rec) > 6_000_000
        else:
            assert np.array_equal(off, ref[0]) and np.array_equal(rec, ref[1]), env
