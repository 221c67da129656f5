import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """make sure the checker library exists (plain C, seconds); the HIP library is built by
    __graft_entry__.build() / hmp3_amd/build.sh and must already be present for GPU tests"""
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    yield


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
