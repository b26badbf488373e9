import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "my-lidar-graph-slam-v2_amd"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu_ctx():
    """One csm_ctx for the whole GPU session. Fails loudly (no fallback) when
    the HIP library or the device is missing."""
    from csm_hip import api
    ctx = api.Context(0)
    yield ctx
    ctx.close()
