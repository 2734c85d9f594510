import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_bin():
    """CPU oracle (test infrastructure): built on demand from oracle/cdm_oracle.cpp."""
    exe = os.path.join(ROOT, "oracle", "_build", "cdm_oracle")
    src = os.path.join(ROOT, "oracle", "cdm_oracle.cpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    return exe


@pytest.fixture(scope="session")
def dhigh_prefix(tmp_path_factory):
    from carpedeam_amd import synth
    d = tmp_path_factory.mktemp("prof")
    prefix = str(d / "dhigh")
    synth.write_dhigh_profiles(prefix)
    return prefix
