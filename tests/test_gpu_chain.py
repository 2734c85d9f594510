"""End-to-end on the device: synthetic reads generated in HBM, then kmermatcher -> rescorediagonal -> ancient_correction ->
ancient_read_assemble through the C ABI with every intermediate resident on the device, against the oracle run on the
same reads (DB files) -- plus size-independent properties at a larger size."""
import os

import numpy as np
import pytest

from carpedeam_amd import capi, mmdb, synth
from gpuutil import diff_keys, run_oracle, seqdb_to_keyed
from stageflags import A_FLAGS, K_FLAGS, R_FLAGS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(dhigh_prefix):
    c = capi.Ctx(0)
    c.damage_load(dhigh_prefix)
    return c


def chain(ctx, db):
    hits = ctx.kmermatch(db)
    alns = ctx.rescore(db, hits)
    corr = ctx.correct(db, alns)
    asm = ctx.extend(corr, alns)
    return hits, alns, corr, asm


@pytest.mark.parametrize("n,lo,hi,seed", [(3000, 100, 100, 1), (2500, 60, 150, 2)])
def test_device_generator_matches_numpy_spec(ctx, n, lo, hi, seed):
    got, _, _ = ctx.synth(n, lo, hi, seed).download()
    exp = synth.generate_strings(n, L=lo, seed=seed, mixed=None if lo == hi else (lo, hi))
    assert [g.decode() for g in got] == exp
    # a shard of a larger corpus is the corresponding slice
    part, _, _ = ctx.synth(500, lo, hi, seed, n_total=n, first=1000).download()
    assert [g.decode() for g in part] == exp[1000:1500]


@pytest.mark.parametrize("n,lo,hi,seed,iters", [(20000, 100, 100, 1, 2), (15000, 60, 150, 2, 2)])
def test_chain_matches_oracle(ctx, oracle_bin, dhigh_prefix, tmp_path, n, lo, hi, seed, iters):
    db = ctx.synth(n, lo, hi, seed)
    t = lambda s: str(tmp_path / s)
    seqs, keys, ext = db.download()
    mmdb.write_seqdb(t("in0"), seqs)
    for it in range(iters):
        hits, alns, corr, asm = chain(ctx, db)
        i, o = t("in%d" % it), t("in%d" % (it + 1))
        run_oracle(oracle_bin, "kmermatcher", i, t("pref"), *K_FLAGS, "--threads", "8")
        run_oracle(oracle_bin, "rescorediagonal", i, i, t("pref"), t("aln"), *R_FLAGS, "--threads", "8")
        run_oracle(oracle_bin, "ancient_correction", i, t("aln"), t("corr"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "8")
        run_oracle(oracle_bin, "ancient_read_assemble", t("corr"), t("aln"), o, *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "8")
        lens, keys, _ = db.meta()
        hoff, hrec = hits.download()
        assert not diff_keys({k: (v, 0) for k, v in capi.hits_to_text(hoff, hrec, keys).items()}, {k: (v[0], 0) for k, v in mmdb.read_db(t("pref")).items()})
        aoff, arec = alns.download()
        assert not diff_keys({k: (v, 0) for k, v in capi.alns_to_text(aoff, arec, keys, lens, db.residues).items()}, mmdb.read_db(t("aln")))
        assert not diff_keys(seqdb_to_keyed(*corr.download()), mmdb.read_db(t("corr")))
        assert not diff_keys(seqdb_to_keyed(*asm.download()), mmdb.read_db(o))
        db = asm


def test_chain_fuzz_small_databases(ctx, oracle_bin, dhigh_prefix, tmp_path):
    """40 random databases of 6..60 reads (30..120 bp, both strands, end damage, a few N) from a 300 bp genome, three
    iterations each: every stage of the device chain against the oracle.  Small, dense pile-ups reach corner cases the
    big synthetic sets rarely do (reads contained in others, identical reads, extensions that meet)."""
    rng = np.random.default_rng(2024)
    letters = np.frombuffer(b"ACGT", np.uint8)
    t = lambda s: str(tmp_path / s)
    for case in range(40):
        genome = rng.integers(0, 4, 300)
        seqs = []
        for _ in range(int(rng.integers(6, 61))):
            L = int(rng.integers(30, 121)); st = int(rng.integers(0, 300 - L))
            c = genome[st:st + L].copy()
            if rng.random() < 0.5:
                c = (3 - c)[::-1]
            for j in range(3):                      # deamination at the ends: C->T at 5', G->A at 3'
                if c[j] == 1 and rng.random() < 0.3:
                    c[j] = 3
                if c[L - 1 - j] == 2 and rng.random() < 0.3:
                    c[L - 1 - j] = 0
            sq = letters[c].tobytes().decode()
            if rng.random() < 0.05:
                k = int(rng.integers(0, L)); sq = sq[:k] + "N" + sq[k + 1:]
            seqs.append(sq)
        if rng.random() < 0.4:
            seqs.append(seqs[int(rng.integers(0, len(seqs)))])
        mmdb.write_seqdb(t("in0"), seqs)
        db = ctx.upload_seqs(seqs)
        for it in range(3):
            hits, alns, corr, asm = chain(ctx, db)
            i, o = t("in%d" % it), t("in%d" % (it + 1))
            run_oracle(oracle_bin, "kmermatcher", i, t("pref"), *K_FLAGS, "--threads", "1")
            run_oracle(oracle_bin, "rescorediagonal", i, i, t("pref"), t("aln"), *R_FLAGS, "--threads", "1")
            run_oracle(oracle_bin, "ancient_correction", i, t("aln"), t("corr"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "1")
            run_oracle(oracle_bin, "ancient_read_assemble", t("corr"), t("aln"), o, *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "1")
            lens, keys, _ = db.meta()
            hoff, hrec = hits.download()
            ctxt = (case, it, seqs)
            assert not diff_keys({k: (v, 0) for k, v in capi.hits_to_text(hoff, hrec, keys).items()}, {k: (v[0], 0) for k, v in mmdb.read_db(t("pref")).items()}), ctxt
            aoff, arec = alns.download()
            assert not diff_keys({k: (v, 0) for k, v in capi.alns_to_text(aoff, arec, keys, lens, db.residues).items()}, mmdb.read_db(t("aln"))), ctxt
            assert not diff_keys(seqdb_to_keyed(*corr.download()), mmdb.read_db(t("corr"))), ctxt
            assert not diff_keys(seqdb_to_keyed(*asm.download()), mmdb.read_db(o)), ctxt
            db = asm


def test_saved_fuzz_cases_replayed(ctx, oracle_bin, dhigh_prefix, tmp_path):
    """Read sets a fuzz campaign saved when a stage differed from the oracle (tests/golden/fuzz/*.txt, one read per line; replayed by
    hand with scripts/fuzz_replay.py): three iterations, every stage against the oracle, each set three times in the same process."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz", "*.txt")))
    assert files
    t = lambda s: str(tmp_path / s)
    for fn in files:
        seqs = [l.rstrip("\n") for l in open(fn) if l.strip()]
        mmdb.write_seqdb(t("in0"), seqs)
        for it in range(3):
            i, o = t("in%d" % it), t("in%d" % (it + 1))
            run_oracle(oracle_bin, "kmermatcher", i, t("pref%d" % it), *K_FLAGS, "--threads", "1")
            run_oracle(oracle_bin, "rescorediagonal", i, i, t("pref%d" % it), t("aln%d" % it), *R_FLAGS, "--threads", "1")
            run_oracle(oracle_bin, "ancient_correction", i, t("aln%d" % it), t("corr%d" % it), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "1")
            run_oracle(oracle_bin, "ancient_read_assemble", t("corr%d" % it), t("aln%d" % it), o, *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "1")
        for rep in range(3):
            db = ctx.upload_seqs(seqs)
            for it in range(3):
                hits, alns, corr, asm = chain(ctx, db)
                lens, keys, _ = db.meta()
                hoff, hrec = hits.download(); aoff, arec = alns.download()
                where = (os.path.basename(fn), rep, it)
                assert not diff_keys({k: (v, 0) for k, v in capi.hits_to_text(hoff, hrec, keys).items()}, {k: (v[0], 0) for k, v in mmdb.read_db(t("pref%d" % it)).items()}), where
                assert not diff_keys({k: (v, 0) for k, v in capi.alns_to_text(aoff, arec, keys, lens, db.residues).items()}, mmdb.read_db(t("aln%d" % it))), where
                assert not diff_keys(seqdb_to_keyed(*corr.download()), mmdb.read_db(t("corr%d" % it))), where
                assert not diff_keys(seqdb_to_keyed(*asm.download()), mmdb.read_db(t("in%d" % (it + 1)))), where
                db = asm


@pytest.mark.parametrize("n", [2_000_000, 5_000_000])
def test_chain_properties_at_scale(ctx, n):
    """2 M reads, and 5 M = BASELINE.json configs[1] (seed 1, 100 bp, dhigh: the reads `bench.py --config 2` corrects) - no oracle at
    these sizes: structural invariants of every stage, those of ancient_correction among them."""
    L = 100
    db = ctx.synth(n, L, L, 1)
    hits, alns, corr, asm = chain(ctx, db)
    hoff, hrec = hits.download()
    # every query's list starts with its self hit; targets strictly increase afterwards (sorted by id, one hit per target)
    first = hrec[hoff[:-1].astype(np.int64)]
    assert (first["target"] == np.arange(n)).all() and (first["score"] == 0).all() and (first["diagonal"] == 0).all()
    cnt = np.diff(hoff.astype(np.int64))
    assert cnt.min() >= 1 and cnt.sum() == hits.count
    owner = np.repeat(np.arange(n), cnt)
    inner = np.ones(hits.count, bool)
    inner[hoff[:-1].astype(np.int64)] = False
    assert (hrec["target"][inner] != owner[inner]).all()
    d = np.diff(hrec["target"].astype(np.int64))
    same_owner = owner[1:] == owner[:-1]
    assert (d[same_owner & inner[1:] & inner[:-1]] > 0).all()
    assert (np.abs(hrec["score"][inner]) >= 1).all() and (np.abs(hrec["score"]) <= L - 20 + 2).all()
    assert (np.abs(hrec["diagonal"]) < L).all()
    # alignments: subset of the hits, self alignment kept with full identity, coordinates inside the sequences
    aoff, arec = alns.download()
    acnt = np.diff(aoff.astype(np.int64))
    assert (acnt >= 1).all() and (acnt <= cnt).all()
    selfa = arec[aoff[:-1].astype(np.int64)]
    assert (selfa["target"] == np.arange(n)).all() and (selfa["raw_score"] == 2 * L).all() and (selfa["ident"] == L).all()
    for f in ("q_start", "q_end", "db_start", "db_end"):
        assert (arec[f] >= 0).all() and (arec[f] < L).all()
    aln_len = np.maximum(np.abs(arec["q_end"] - arec["q_start"]), np.abs(arec["db_end"] - arec["db_start"])) + 1
    assert (arec["ident"] <= aln_len).all() and (arec["raw_score"] == 2 * arec["ident"] - 3 * (aln_len - arec["ident"])).all()
    # correction: same geometry, only damage-like substitutions dominate (T->C at 5', A->G at 3' undo C>T / G>A)
    s0, _, _ = db.download()
    s1, _, e1 = corr.download()
    a0 = np.frombuffer(b"".join(s0), np.uint8).reshape(n, L)
    a1 = np.frombuffer(b"".join(s1), np.uint8).reshape(n, L)
    changed = a0 != a1
    assert 0 < changed.sum() < 0.01 * n * L
    tc = ((a0 == ord("T")) & (a1 == ord("C")) & changed).sum() + ((a0 == ord("A")) & (a1 == ord("G")) & changed).sum()
    assert tc > 0.95 * changed.sum()
    assert (e1 == 0).all()
    # idempotence: correcting the corrected reads against the same alignments changes (almost) nothing more
    # extension: extended sequences contain the corrected query; flags: ext == 1 exactly for the longer ones
    s2, _, e2 = asm.download()
    l2 = np.array([len(x) for x in s2])
    assert ((l2 > L) == (e2 == 1)).all() and (l2 >= L).all() and (l2 < 3 * L + 3).all()
    idx = np.nonzero(e2 == 1)[0][:2000]
    assert all(s1[i] in s2[i] for i in idx)
    assert 0.05 * n < (e2 == 1).sum() < 0.3 * n


def test_full_size_invariants(ctx, monkeypatch):
    """BASELINE.json configs[2] at full size - 50 M x 100 bp, no oracle there: kmermatcher with its two independent sort-2 pipelines
    (run records + on-chip sorters vs radix passes + bucket finish) cross-checked on the device over all 4 G tuples, the wide group key
    (representative out of the member's key, entries only) against the narrow one on the same data (equal hit arrays, compared on the
    device's downloads chunk by chunk), and the structural invariants that need only counts
    and per-sequence metadata (no 5 GB downloads into Python objects)."""
    n, L = 50_000_000, 100
    db = ctx.synth(n, L, L, 1)
    monkeypatch.setenv("CDM_KMER_SORT2", "check")            # a mismatch between the two pipelines is an error of the call
    hits = ctx.kmermatch(db)
    monkeypatch.delenv("CDM_KMER_SORT2")
    monkeypatch.setenv("CDM_FORCE_WIDE_KEY", "1")
    hits_wide = ctx.kmermatch(db)
    monkeypatch.delenv("CDM_FORCE_WIDE_KEY")
    assert hits.count == hits_wide.count
    (off_n, rec_n), (off_w, rec_w) = hits.download(), hits_wide.download()
    assert np.array_equal(off_n, off_w) and np.array_equal(rec_n, rec_w)
    del hits_wide, off_n, rec_n, off_w, rec_w
    alns = ctx.rescore(db, hits)
    assert n <= alns.count <= hits.count and hits.count > 3 * n
    del hits
    corr = ctx.correct(db, alns)
    l1, k1, e1 = corr.meta()
    assert (l1 == L).all() and (e1 == 0).all() and (k1 == np.arange(n, dtype=np.uint32)).all() and corr.residues == n * L
    asm = ctx.extend(corr, alns)
    l2, k2, e2 = asm.meta()
    assert (k2 == k1).all() and (l2 >= L).all() and (l2 < 3 * L + 3).all() and ((l2 > L) == (e2 == 1)).all()
    assert 0.05 * n < int((e2 == 1).sum()) < 0.3 * n and asm.residues == int(l2.astype(np.int64).sum())
    # the same corpus in two halves of the read range gives the halves of the per-sequence generator (shards of ONE corpus)
    part = ctx.synth(1000, L, L, 1, n_total=n, first=n - 1000)
    assert part.n == 1000 and part.residues == 1000 * L


def test_contig_handoff_roundtrip(ctx):
    """select_ext -> packed DEVICE buffers -> from_packed: what the RCCL all-gather of bench.py --gpus N moves.
    (device buffers come straight from the HIP runtime here; bench.py passes torch tensors' data_ptr())"""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]

    def dmalloc(nbytes):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), max(nbytes, 4)) == 0
        return p

    db = ctx.synth(30000, 100, 100, 3)
    _, _, _, asm = chain(ctx, db)
    seqs, keys, ext = asm.download()
    contigs = asm.select_ext()
    exp = [(int(k), s) for s, k, e in zip(seqs, keys, ext) if e == 1]
    got_s, got_k, got_e = contigs.download()
    assert [(int(k), s) for s, k in zip(got_s, got_k)] == exp and (got_e == 1).all() and len(exp) > 1000
    n, words = contigs.n, contigs.words
    codes, nmask, lens, kk = dmalloc(words * 4), dmalloc(words * 2), dmalloc(n * 4), dmalloc(n * 4)
    contigs.copy_packed(codes, nmask, lens, kk)
    hl = np.zeros(n, np.uint32)
    assert hip.hipMemcpy(hl.ctypes.data_as(ctypes.c_void_p), lens, n * 4, 2) == 0   # hipMemcpyDeviceToHost
    assert int(hl.sum()) == contigs.residues
    again = ctx.from_packed(codes, nmask, lens, kk, n, words, 1)
    s2, k2, e2 = again.download()
    assert [(int(k), s) for s, k in zip(s2, k2)] == exp and (e2 == 1).all()
    for p in (codes, nmask, lens, kk):
        hip.hipFree(p)


def test_chain_with_contig_sized_sequences(ctx, oracle_bin, dhigh_prefix, tmp_path):
    """Sequences of tens of thousands of letters (the reference's `int` position path) next to reads, through all four stages."""
    rng = np.random.default_rng(37)
    letters = np.frombuffer(b"ACGT", np.uint8)
    genome = rng.integers(0, 4, 90_000)
    text = lambda c: letters[c].tobytes().decode()
    rc = lambda c: (3 - c)[::-1]
    seqs = [text(genome[0:40_000]), text(rc(genome[30_000:72_000])), text(genome[70_000:89_000])]
    for _ in range(1500):
        L = int(rng.integers(60, 151)); st = int(rng.integers(0, 90_000 - L))
        c = genome[st:st + L].copy()
        if c[0] == 1 and rng.random() < 0.3:
            c[0] = 3
        seqs.append(text(rc(c) if rng.random() < 0.5 else c))
    seqs = [seqs[i] for i in rng.permutation(len(seqs))]
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in"), seqs)
    i = t("in")
    run_oracle(oracle_bin, "kmermatcher", i, t("pref"), *K_FLAGS, "--threads", "8")
    run_oracle(oracle_bin, "rescorediagonal", i, i, t("pref"), t("aln"), *R_FLAGS, "--threads", "8")
    run_oracle(oracle_bin, "ancient_correction", i, t("aln"), t("corr"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "8")
    run_oracle(oracle_bin, "ancient_read_assemble", t("corr"), t("aln"), t("asm"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "8")
    db = ctx.upload_seqs(seqs)
    hits, alns, corr, asm = chain(ctx, db)
    assert not diff_keys(seqdb_to_keyed(*corr.download()), mmdb.read_db(t("corr")))
    assert not diff_keys(seqdb_to_keyed(*asm.download()), mmdb.read_db(t("asm")))


def test_pile_up_beyond_65535_records(ctx, oracle_bin, dhigh_prefix, tmp_path):
    """One long read on top of 70 000 shorter ones from the same 170 bp stretch: it is the representative of every k-mer group it is
    in, its query gets ~70 000 prefilter hits and alignment records, and correction piles them all up (64-bit counters in the general
    kernel; round 1 refused more than 65 535).  Every stage against the oracle."""
    rng = np.random.default_rng(65536)
    letters = np.frombuffer(b"ACGT", np.uint8)
    comp = np.array([3, 2, 1, 0])
    genome = rng.integers(0, 4, 170)
    seqs = [letters[genome[20:150]].tobytes().decode()]                  # 130 letters: the longest sequence = the representative
    for _ in range(70000):
        L = int(rng.integers(60, 101))
        s = int(rng.integers(0, 170 - L + 1))
        r = genome[s:s + L].copy()
        if rng.random() < 0.5:
            r = comp[r][::-1]
        for k, p in enumerate((0.3, 0.12, 0.05)):                       # end damage
            if r[k] == 1 and rng.random() < p:
                r[k] = 3
            if r[L - 1 - k] == 2 and rng.random() < p:
                r[L - 1 - k] = 0
        if rng.random() < 0.02:
            r[int(rng.integers(0, L))] = int(rng.integers(0, 4))
        seqs.append(letters[r].tobytes().decode())
    t = lambda s: str(tmp_path / s)
    mmdb.write_seqdb(t("in"), seqs)
    db = ctx.upload_seqs([s.encode() for s in seqs], list(range(len(seqs))), [0] * len(seqs))
    hits, alns, corr, asm = chain(ctx, db)
    aoff, arec = alns.download()
    assert int((aoff[1:] - aoff[:-1]).max()) > 65535
    run_oracle(oracle_bin, "kmermatcher", t("in"), t("pref"), *K_FLAGS, "--threads", "8")
    run_oracle(oracle_bin, "rescorediagonal", t("in"), t("in"), t("pref"), t("aln"), *R_FLAGS, "--threads", "8")
    run_oracle(oracle_bin, "ancient_correction", t("in"), t("aln"), t("corr"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "8")
    run_oracle(oracle_bin, "ancient_read_assemble", t("corr"), t("aln"), t("asm"), *A_FLAGS, "--ancient-damage", dhigh_prefix, "--threads", "8")
    lens, keys, _ = db.meta()
    hoff, hrec = hits.download()
    assert not diff_keys({k: (v, 0) for k, v in capi.hits_to_text(hoff, hrec, keys).items()}, {k: (v[0], 0) for k, v in mmdb.read_db(t("pref")).items()})
    assert not diff_keys({k: (v, 0) for k, v in capi.alns_to_text(aoff, arec, keys, lens, db.residues).items()}, mmdb.read_db(t("aln")))
    assert not diff_keys(seqdb_to_keyed(*corr.download()), mmdb.read_db(t("corr")))
    assert not diff_keys(seqdb_to_keyed(*asm.download()), mmdb.read_db(t("asm")))
