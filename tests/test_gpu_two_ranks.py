"""Two ranks as two PROCESSES on the one GPU of the box (scripts/two_rank_check.py): torch.distributed over gloo with the device
tensors staged through the host (RCCL wants a device per rank), every rank with its own library context, allocator and streams."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_processes_share_one_gpu(dhigh_prefix):
    from carpedeam_amd import build
    build.build()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
                        os.path.join(ROOT, "scripts", "two_rank_check.py"), dhigh_prefix], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    assert "rank 0 of 2 ok" in r.stdout and "rank 1 of 2 ok" in r.stdout


def test_bench_line_from_two_ranks():
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one process per rank, --gpus 2, the exact scheme) - with
    CDM_BENCH_BACKEND=gloo, so that the two ranks can share this box's one device (RCCL wants a device per rank; the library's calls go
    over the gloo transport).  The line must say n_gpus 2 and carry the single-device hit and alignment counts."""
    import json
    from carpedeam_amd import capi
    n = 1_500_000
    ctx = capi.Ctx(0)
    db = ctx.synth(n, 100, 100, 1)
    hits = ctx.kmermatch(db)
    want = (hits.count, ctx.rescore(db, hits).count)
    del hits, db, ctx
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CDM_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29543",
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", str(n), "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong" and d["config"]["equivalent_to_single_device"] is True
    assert (d["config"]["prefilter_hits"], d["config"]["alignments"]) == want, (d["config"], want)
    assert d["value"] > 0 and d["roofline"]["frac"] > 0


def test_config5_line_from_two_ranks_equals_one_rank():
    """`bench.py --config 5 --gpus 2` (the 12-iteration loop, every iteration split over the ranks by the library - read iterations and
    contig iterations alike) ends in the DB of the single-process run: same final sequences and letters"""
    import json
    n = 150_000
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CDM_BENCH_BACKEND="gloo")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "5", "--reads", str(n), "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert one.returncode == 0, (one.stdout[-1500:], one.stderr[-2500:])
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29545",
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "5", "--reads", str(n), "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert two.returncode == 0, (two.stdout[-1500:], two.stderr[-2500:])
    a, b = (json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][0]) for r in (one, two))
    assert b["n_gpus"] == 2 and b["config"]["equivalent_to_single_device"] is True and "cdm_contig_iteration_dist" in b["config"]["multi_gpu_scheme"]
    for k in ("final_sequences_rank0", "final_residues_rank0", "circular_contigs_set_aside_rank0"):
        assert a["config"][k] == b["config"][k], k
    assert a["config"]["final_residues_rank0"] > n * 110          # the contigs grew (the reads hold 105 letters on average)
