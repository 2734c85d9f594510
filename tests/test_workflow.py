"""BASELINE config 1: `carpedeam ancient_assemble` on the reference's example reads, through the front end (carpedeam_amd/carpedeam,
csrc/host/front.c) that takes the reference binary's place.  The reference's own workflow drivers and scripts run
(oracle/_ref/carpedeam_full = the reference's whole program compiled in place by oracle/Makefile.ref), and because the front end
starts it with argv[0] = itself, every "$MMSEQS" <module> call of those scripts (Application.cpp:198, data/nuclassemble.sh:105-136,
data/guidedNuclAssemble.sh, linclust.sh) comes back through it: the modules of the hot path go to the device binary, the rest to
the reference.  Golden: tests/golden/example/ancient_assemble.fasta (make_golden.py workflow).

CPU part: the routing alone, with the reference's own modules (oracle/_ref/carpedeam_ref) standing in for the device binary.
GPU part: the real thing."""
import collections
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRONT = os.path.join(ROOT, "carpedeam_amd", "carpedeam")
MODULES = os.path.join(ROOT, "carpedeam_amd", "carpedeam_mi355x")
REF_MODULES = os.path.join(ROOT, "oracle", "_ref", "carpedeam_ref")
REF_FULL = os.path.join(ROOT, "oracle", "_ref", "carpedeam_full")
EXAMPLE = os.path.join(ROOT, "tests", "golden", "example")
OWNED = ["kmermatcher", "rescorediagonal", "ancient_correction", "ancient_read_assemble", "ancient_contig_merge", "cyclecheck", "createdb", "createhdb",
         "convert2fasta", "clust", "createsubdb", "filterdb", "mergeclusters", "result2repseq", "rmdb", "mvdb", "align"]
HOST_ONLY = ["clust", "createsubdb", "filterdb", "mergeclusters", "result2repseq", "rmdb", "mvdb", "align"]      # csrc/host/cluster.cpp: no device needed
# what is left on the reference binary: its three workflow drivers (they parse the workflow's flags, write the shell scripts and run them)
ON_REFERENCE = {"ancient_assemble": 1, "nuclassemble": 1, "linclust": 1}


def fasta_records(path):
    recs, name = [], None
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            name = line
            recs.append([name, ""])
        elif recs:
            recs[-1][1] += line
    return recs


def assemble(tmp_path, dhigh_prefix, modules):
    from carpedeam_amd import build
    build.build()
    log = str(tmp_path / "dispatch.log")
    env = dict(os.environ, CARPEDEAM_GPU_BIN=modules, CARPEDEAM_REF_BIN=REF_FULL, CARPEDEAM_DISPATCH_LOG=log)
    out = str(tmp_path / "out.fasta")
    r = subprocess.run([FRONT, "ancient_assemble", os.path.join(EXAMPLE, "test_data.fq.gz"), out, str(tmp_path / "tmp"), "--ancient-damage", dhigh_prefix,
                        "--threads", "8"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    calls = collections.Counter(tuple(l.split()) for l in open(log))
    return fasta_records(out), calls


def check_routing(calls, fallbacks):
    # the workflow's defaults: 5 read iterations + 5 contig iterations (log of the reference run: STEP 0..9), then linclust
    assert {m: c for (where, m), c in calls.items() if where == "ref"} == ON_REFERENCE          # nothing else runs on the reference
    for m in OWNED:
        assert calls[("ref", m)] == 0, m                      # not one call of an owned module went around the front end
    gpu = {m: calls[("gpu", m)] for m in OWNED}
    assert gpu["ancient_correction"] == 10 and gpu["ancient_read_assemble"] == 5 and gpu["ancient_contig_merge"] == 5 and gpu["cyclecheck"] == 5
    assert gpu["createdb"] == 1 and gpu["createhdb"] == 1 and gpu["convert2fasta"] == 1
    assert gpu["kmermatcher"] == 11                            # 10 of the loop + linclust's
    # linclust's tail (linclust.sh:33-87, guidedNuclAssemble.sh:190-195) and the scripts' housekeeping
    assert gpu["clust"] == 2 and gpu["createsubdb"] == 3 and gpu["filterdb"] == 1 and gpu["mergeclusters"] == 1 and gpu["result2repseq"] == 1
    assert gpu["rmdb"] > 30 and gpu["mvdb"] == 1 and gpu["align"] == 1
    # (a module call the device path refuses - status 77 before any work - is REFUSED by the front end, never handed to the reference,
    # none in this workflow - linclust's Hamming-distance pre-clustering pass, linclust.sh:27-31,
    # is a mode of the device module)
    assert gpu["rescorediagonal"] == 11
    assert sum(n for (where, _), n in calls.items() if where in ("fallback", "refused")) == fallbacks == 0


@pytest.mark.skipif(not (os.path.exists(REF_FULL) and os.path.exists(REF_MODULES)), reason="oracle/_ref (the reference's object code) is not built here")
def test_front_end_routes_every_module_call_of_the_reference_workflow(tmp_path, dhigh_prefix):
    # no device here: the reference's own modules stand in for the device modules; the product's host-only modules run as they are
    stand_in = str(tmp_path / "modules.sh")
    open(stand_in, "w").write('#!/bin/sh\ncase "$1" in\n  %s) exec %s "$@";;\n  *) exec %s "$@";;\nesac\n' % ("|".join(HOST_ONLY), MODULES, REF_MODULES))
    os.chmod(stand_in, 0o755)
    recs, calls = assemble(tmp_path, dhigh_prefix, stand_in)
    check_routing(calls, fallbacks=0)
    assert recs == fasta_records(os.path.join(EXAMPLE, "ancient_assemble.fasta"))


def test_front_end_without_a_reference_binary(tmp_path):
    from carpedeam_amd import build
    build.build()
    env = {k: v for k, v in os.environ.items() if k != "CARPEDEAM_REF_BIN"}
    r = subprocess.run([FRONT, "ancient_assemble", "a", "b", "c"], capture_output=True, text=True, env=env)
    assert r.returncode == 1 and "Invalid Command: ancient_assemble" in r.stderr and "CARPEDEAM_REF_BIN" in r.stderr
    # a refusal of the device binary (status 77 inside) stays a plain failure when nothing can take the call
    r = subprocess.run([FRONT, "rescorediagonal", "a", "a", "p", "o", "--rescore-mode", "0"], capture_output=True, text=True, env=env)
    assert r.returncode == 1 and "not supported by the MI355X path" in r.stderr
    r = subprocess.run([MODULES, "rescorediagonal", "a", "a", "p", "o", "--rescore-mode", "0"], capture_output=True, text=True, env=env)
    assert r.returncode == 77
    # ... and ALSO when there is a reference binary (/bin/echo stands in: it would print its arguments): an owned module is never
    # computed by the reference behind the caller's back
    log = str(tmp_path / "dispatch.log")
    env["CARPEDEAM_REF_BIN"] = "/bin/echo"
    env["CARPEDEAM_DISPATCH_LOG"] = log
    r = subprocess.run([FRONT, "rescorediagonal", "a", "a", "p", "o", "--rescore-mode", "0"], capture_output=True, text=True, env=env)
    assert r.returncode == 1 and r.stdout == "" and "not handed to the reference binary" in r.stderr
    assert open(log).read().split() == ["refused", "rescorediagonal"]
    # round 4's opt-in hand-over is gone from the shipped binary: the variable changes nothing
    env["CARPEDEAM_ALLOW_REF_FALLBACK"] = "1"
    r = subprocess.run([FRONT, "rescorediagonal", "a", "a", "p", "o", "--rescore-mode", "0"], capture_output=True, text=True, env=env)
    assert r.returncode == 1 and r.stdout == "" and "not handed to the reference binary" in r.stderr
    assert open(log).read().split() == ["refused", "rescorediagonal", "refused", "rescorediagonal"]
    assert b"ALLOW_REF_FALLBACK" not in open(FRONT, "rb").read()


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_FULL), reason="oracle/_ref (the reference's object code) is not built here")
def test_ancient_assemble_on_the_example_reads(tmp_path, dhigh_prefix):
    recs, calls = assemble(tmp_path, dhigh_prefix, MODULES)
    check_routing(calls, fallbacks=0)      # (linclust's Hamming-distance rescorediagonal runs on the device too)
    exp = fasta_records(os.path.join(EXAMPLE, "ancient_assemble.fasta"))
    assert sorted(s for _, s in recs) == sorted(s for _, s in exp)
    assert recs == exp
