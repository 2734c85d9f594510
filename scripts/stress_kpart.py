#!/usr/bin/env python3
"""Stress of the ranks-as-threads harness (tests/test_gpu_shards.py) around cdm_kmermatch_part, for the one host segmentation fault
round 3 saw there:   python scripts/stress_kpart.py [seconds] [getenv-race]

W rank threads of one process, each with its own context, run the exact scheme's kmermatcher on small random databases over and
over for `seconds`; faulthandler prints the Python stacks of all threads on a fatal signal and CDM_SEGV_BACKTRACE=1 makes the
library print the native stack of the faulting thread.  With `getenv-race` a further thread keeps adding NEW variables to the
process environment through libc's setenv (what grows and reallocates `environ`): a library that calls getenv on its call paths dies
of that within seconds; one that snapshots its switches does not care."""
import ctypes
import faulthandler
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["CDM_SEGV_BACKTRACE"] = "1"
faulthandler.enable(all_threads=True)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    race = len(sys.argv) > 2 and sys.argv[2] == "getenv-race"
    from carpedeam_amd import capi, shard
    from test_gpu_shards import merged_hits, run_ranks
    stop = threading.Event()

    def start_churn():
        # (started after the first databases: the HIP runtime itself calls getenv while it initialises - the first run of this script,
        # with the setenv thread up from the start, died inside libamdhip64's getenv under cdm_ctx_create)
        libc = ctypes.CDLL(None)
        libc.setenv.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]

        def churn():
            i = 0
            while not stop.is_set():
                libc.setenv(b"STRESS_VAR_%d" % i, b"x" * 64, 1)
                i += 1
                if i % 64 == 0:
                    time.sleep(0.0005)
        threading.Thread(target=churn, daemon=True).start()
    rng = np.random.default_rng(1234)
    letters = np.frombuffer(b"ACGT", np.uint8)
    ref = capi.Ctx(0)
    import torch
    torch.zeros(4, device="cuda").sum().item()
    t0, cases, calls = time.time(), 0, 0
    while time.time() - t0 < seconds:
        if race and cases == 3:
            start_churn()
            print("stress_kpart: setenv thread started", flush=True)
        genome = rng.integers(0, 4, 400)
        seqs = []
        for _ in range(int(rng.integers(3, 80))):
            L = int(rng.integers(18, 150)); st = int(rng.integers(0, 400 - L))
            c = genome[st:st + L].copy()
            if rng.random() < 0.5:
                c = (3 - c)[::-1]
            seqs.append(letters[c].tobytes())
        db = ref.upload_seqs(seqs)
        want = ref.kmermatch(db).download()
        for world in (2, 3, 5):
            def rank_fn(rank, comm, seqs=seqs):
                c = capi.Ctx(0)
                return shard.kmermatch_exact(c, c.upload_seqs(seqs), comm).download()
            off, rec = merged_hits(run_ranks(world, rank_fn), len(seqs))
            assert np.array_equal(off, want[0]) and np.array_equal(rec, want[1]), (cases, world)
            calls += world
        cases += 1
        if cases % 50 == 0:
            print("stress_kpart: %d databases, %d cdm_kmermatch_part calls, %.0f s" % (cases, calls, time.time() - t0), flush=True)
    stop.set()
    print("stress_kpart: done, %d databases, %d cdm_kmermatch_part calls in rank threads, %.0f s%s, no fault" % (
        cases, calls, time.time() - t0, ", with a setenv thread beside them" if race else ""))


if __name__ == "__main__":
    main()
