"""BASELINE.json configs[3] on one device: ONE corpus split into W read shards (scheme "reads" of carpedeam_amd/dist.py), the
four stages per shard on the device, the per-shard contigs merged through the very hand-off the multi-GPU run uses
(select_ext -> copy_packed -> [all-gather] -> from_packed) - against the oracle run on the same shards, zero tolerance.  The
price of read sharding (contigs that differ from the un-sharded run) is measured and bounded; DESIGN.md quotes it."""
import numpy as np
import pytest
import torch

from carpedeam_amd import capi, dist as cd, mmdb
from gpuutil import run_oracle
from stageflags import A_FLAGS, K_FLAGS, R_FLAGS

pytestmark = pytest.mark.gpu
N_READS = 200_000


@pytest.fixture(scope="module")
def ctx(dhigh_prefix):
    c = capi.Ctx(0)
    c.damage_load(dhigh_prefix)
    return c


def oracle_chain(oracle_bin, dhigh_prefix, t, tag, seqs):
    i = t(tag + "_in")
    mmdb.write_seqdb(i, seqs)
    dmg = ["--ancient-damage", dhigh_prefix, "--threads", "8"]
    run_oracle(oracle_bin, "kmermatcher", i, t(tag + "_pref"), *K_FLAGS, "--threads", "8")
    run_oracle(oracle_bin, "rescorediagonal", i, i, t(tag + "_pref"), t(tag + "_aln"), *R_FLAGS, "--threads", "8")
    run_oracle(oracle_bin, "ancient_correction", i, t(tag + "_aln"), t(tag + "_corr"), *A_FLAGS, *dmg)
    run_oracle(oracle_bin, "ancient_read_assemble", t(tag + "_corr"), t(tag + "_aln"), t(tag + "_asm"), *A_FLAGS, *dmg)
    return mmdb.read_db(t(tag + "_asm"))


@pytest.mark.parametrize("world", [2, 4])
def test_read_shards_merge_equals_oracle_on_the_same_shards(ctx, oracle_bin, dhigh_prefix, tmp_path, world):
    t = lambda s: str(tmp_path / s)
    parts, want = [], {}
    for rank in range(world):
        plan = cd.shard_plan(rank, world, N_READS, 1)
        db = ctx.synth(plan["n"], 100, 100, plan["seed"], n_total=plan["n_total"], first=plan["first"])
        hits = ctx.kmermatch(db)
        alns = ctx.rescore(db, hits)
        corr = ctx.correct(db, alns)
        asm = ctx.extend(corr, alns)
        buf, n, words = cd.pack_contigs(ctx, asm)                # what rank `rank` would hand to the all-gather
        parts.append((buf.clone(), n, words, plan["first"]))
        seqs, _, _ = db.download()
        for k, (payload, ext) in oracle_chain(oracle_bin, dhigh_prefix, t, "s%d" % rank, seqs).items():
            if ext == 1:
                want[k + plan["first"]] = payload
    merged = cd.unpack_to_db(ctx, parts)                          # what every rank holds after it
    seqs, keys, ext = merged.download()
    got = {int(k): bytes(s) + b"\n" for s, k in zip(seqs, keys)}
    assert set(ext.tolist()) <= {1}
    assert len(want) > 1000
    assert got == want


def test_price_of_read_sharding(ctx, oracle_bin, dhigh_prefix, tmp_path):
    """Contigs of the 2- and 4-shard runs that the un-sharded oracle run does not have (and vice versa): reads whose overlap
    partners live in another shard extend differently.  With uniformly placed reads a shard sees 1/W of every pile-up."""
    t = lambda s: str(tmp_path / s)
    whole, _, _ = ctx.synth(N_READS, 100, 100, 1).download()
    full = {p for p, e in oracle_chain(oracle_bin, dhigh_prefix, t, "full", whole).values() if e == 1}
    report = {}
    for world in (2, 4):
        sharded = set()
        for rank in range(world):
            plan = cd.shard_plan(rank, world, N_READS, 1)
            db = ctx.synth(plan["n"], 100, 100, 1, n_total=N_READS, first=plan["first"])
            hits = ctx.kmermatch(db); alns = ctx.rescore(db, hits); corr = ctx.correct(db, alns); asm = ctx.extend(corr, alns)
            seqs, _, ext = asm.download()
            sharded |= {bytes(s) + b"\n" for s, e in zip(seqs, ext) if e == 1}
        report[world] = (len(sharded), len(sharded - full), len(full - sharded))
    print("read sharding, %d reads, 1 iteration: un-sharded %d contigs; W=2: %d contigs, %d not in the un-sharded set, %d missing; W=4: %d / %d / %d"
          % ((N_READS, len(full)) + report[2] + report[4]))
    # sharding is NOT equivalent: the test documents the size of the effect and guards against it being mistaken for exact
    assert report[2][1] > 0 and report[4][1] > report[2][1] * 0.5
