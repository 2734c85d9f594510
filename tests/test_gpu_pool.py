"""Device memory from arenas (csrc/pool.h) on the device itself: the CPU build runs the bookkeeping under sanitizers
(tests/test_pool_stress.py); here - that a steady-state call asks the driver for nothing, that buffers which grow from call to call
(the contig iterations of the workflow loop: 1.02-1.5x per iteration) stop asking once head room is on, that what is freed is reused
(mapped memory stays near the largest call's need, where the exact-size block cache of rounds 1-3 kept every size ever asked for), and
that CDM_POOL=blocks still works (a subprocess: the scheme is chosen once per process)."""
import subprocess
import sys

import numpy as np
import pytest

from carpedeam_amd import capi

pytestmark = pytest.mark.gpu


def test_steady_state_asks_the_driver_for_nothing():
    ctx = capi.Ctx(0)
    db = ctx.synth(400_000, 100, 100, 3)
    want = ctx.kmermatch(db).download()
    before = capi.pool_stats()
    for _ in range(3):
        got = ctx.kmermatch(db).download()
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    after = capi.pool_stats()
    assert after["driver_calls"] == before["driver_calls"] and after["requests"] > before["requests"] + 100
    assert after["in_use_bytes"] < after["mapped_bytes"]


def test_growing_buffers_with_head_room():
    ctx = capi.Ctx(0)
    capi.lib().cdm_pool_headroom(1.6)
    try:
        sizes = [300_000]
        while len(sizes) < 9:
            sizes.append(int(sizes[-1] * 1.12))                    # 2.5x over the run
        asked = []
        for n in sizes:
            db = ctx.synth(n, 100, 100, 3)
            before = capi.pool_stats()["driver_bytes"]
            hits = ctx.kmermatch(db)
            assert hits.count > n
            del hits, db
            asked.append(capi.pool_stats()["driver_bytes"] - before)
        # the first calls map memory; once blocks carry head room every other call or so maps nothing (a chunk is 256 MB), and what a growth of 2.5x maps in all stays far
        # below the sum of the calls' needs (every call needs ~40 bytes per k-mer slot, 82 slots per read)
        assert sum(1 for a in asked[2:] if a == 0) >= 3, asked
        assert sum(asked) < 0.5 * sum(n * 82 * 40 for n in sizes), asked
    finally:
        capi.lib().cdm_pool_headroom(1.0)


def test_block_cache_scheme_still_runs(tmp_path):
    code = ("import numpy as np\nfrom carpedeam_amd import capi\nctx = capi.Ctx(0)\ndb = ctx.synth(200000, 100, 100, 3)\n"
            "a = ctx.kmermatch(db).download(); b = ctx.kmermatch(db).download()\nassert np.array_equal(a[1], b[1])\n"
            "print('mapped', capi.pool_stats()['mapped_bytes'], 'hits', len(a[1]))\n")
    import os
    env = dict(os.environ, CDM_POOL="blocks")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.startswith("mapped 0 hits "), r.stdout          # (no arena in that scheme)
