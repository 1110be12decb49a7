import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def built():
    """Build (or reuse) the in-tree native libraries once per session."""
    import __graft_entry__ as g
    g.build()
    return g


@pytest.fixture(scope="session")
def engine_cls(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cranesched_amd.engine import GpuNodeSelector
    return GpuNodeSelector
