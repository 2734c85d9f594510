"""Shared helpers of the GPU parity tests: load golden DBs, run oracle stages, compare keyed DBs."""
import os
import subprocess

from carpedeam_amd import mmdb

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DATASETS = [("synth2k", 2), ("mixed3k", 3), ("example", 1), ("letters", 3)]


def gold(name, stage, it=None):
    fn = "reads.keyed.gz" if stage == "reads" else "%s_%d.keyed.gz" % (stage, it)
    return mmdb.load_keyed(os.path.join(GOLD, name, fn))


def stage_input(name, it):
    """sequence DB that iteration `it` of the read loop starts from"""
    return gold(name, "reads") if it == 0 else gold(name, "asm", it - 1)


class OracleCrash(AssertionError):
    """the oracle (or the reference binary) ended with a signal: an input on which the reference's behaviour is undefined"""


def run_oracle(exe, *args):
    r = subprocess.run([exe] + list(args), capture_output=True, text=True)
    if r.returncode < 0:
        raise OracleCrash("%s %s ended with signal %d" % (os.path.basename(exe), args[0], -r.returncode))
    assert r.returncode == 0, r.stderr[-2000:]


def diff_keys(got, exp):
    got, exp = mmdb.canon(got), mmdb.canon(exp)
    return [k for k in sorted(set(got) | set(exp)) if got.get(k) != exp.get(k)]


def seqdb_to_keyed(seqs, keys, ext):
    return {int(k): (bytes(s) + b"\n", int(e)) for s, k, e in zip(seqs, keys, ext)}
