import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
HOSTSIM = os.path.join(ROOT, "tests", "hostsim", "_build", "hostsim")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "yart_ref")
ORACLE_BIN = os.path.join(ROOT, "oracle", "_build", "yart_oracle")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Native targets exist (build() is idempotent and cheap when up to date)."""
    if not (os.path.exists(os.path.join(ROOT, "yart_amd", "libyart_hip.so")) and os.path.exists(HOSTSIM)):
        import __graft_entry__
        __graft_entry__.build()
    return True


@pytest.fixture(scope="session")
def hostsim(built):
    return HOSTSIM


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, capture_output=True, text=True, **kw)
