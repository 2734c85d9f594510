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
