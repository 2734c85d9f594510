"""The library's device-memory cache and environment snapshot (carpedeam_amd/csrc/pool.h) - the code every context of every host
thread goes through - compiled against stand-ins for hipMalloc / hipFree and driven by 8 threads under ThreadSanitizer and
AddressSanitizer (sanitizers run on the CPU build only: tests/cpu/pool_stress.cpp).  Short-lived threads, blocks freed by another
thread than their allocator's (and after that thread ended), head room changed while others allocate, out-of-memory with blocks
parked in other threads' caches, the environment snapshot refreshed while others read it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpu", "pool_stress.cpp")


@pytest.mark.parametrize("sanitizer,scheme", [("thread", "arenas"), ("address", "arenas"), ("thread", "blocks")])
def test_pool_under_sanitizers(tmp_path, sanitizer, scheme):
    exe = str(tmp_path / ("pool_stress_" + sanitizer))
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + sanitizer, "-pthread", SRC, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, "8", "30", scheme], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    assert ", 0 errors" in r.stdout and " 0 bytes / 0 blocks left on the device, 0 registered" in r.stdout, r.stdout
