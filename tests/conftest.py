import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(autouse=True)
def _no_silent_retry(monkeypatch):
    """cns_run_resident re-runs a cycle on k_pipe / k_select when k_wide ends in a fault >= 20 (fail-soft for production).  In the
    parity tests that would paper over a k_wide that breaks — a full-size C3 once passed its digest that way — so every test
    runs with the retry off; the one test of the retry itself turns it back on."""
    monkeypatch.setenv("CNS_WIDE_NO_RETRY", "1")


@pytest.fixture(scope="session")
def built():
    """Build (or reuse) the in-tree native libraries once per session."""
    import __graft_entry__ as g
    g.build()
    return g


@pytest.fixture
def gpu(built):
    """For GPU tests that build their engine themselves: skips on a machine without a GPU instead of failing with CNS_ERR_NO_DEVICE
    when the whole directory is run without `-m "not gpu"`."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return built


@pytest.fixture(params=["legacy", "pipe", "wide", "wide32"])
def engine_cls(built, request, monkeypatch):
    """The engine class, once per selection kernel: k_select (one worker wave carries test + commit), k_pipe
    (decoupled test / commit pipeline on one CU) and k_wide (k_pipe's protocol with the scanners of a partition spread over
    16 more workgroups — "wide32": 8, the build that serves 9..24 partitions; the 4- and 2-workgroup builds for 25..80 partitions: engine_cls_narrow —, exchange through HBM granules; clusters it does not cover fall back to k_pipe / k_select).  The library reads CNS_SELECT_KERNEL at every run."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    monkeypatch.setenv("CNS_SELECT_KERNEL", request.param)
    from cranesched_amd.engine import GpuNodeSelector
    return GpuNodeSelector


@pytest.fixture(params=["wide16", "wide8"])
def engine_cls_narrow(built, request, monkeypatch):
    """k_wide's builds for clusters of MANY partitions: 16 scanner waves on 4 workgroups (25..48 partitions) and 8 scanner waves on
    2 workgroups (49..80) next to the home workgroup.  CNS_SELECT_KERNEL=wide16 / wide8 caps the build, so that small test clusters
    run on them too (tests/test_gpu_wide_narrow.py: a cross-section of the suite, not all of it again)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    monkeypatch.setenv("CNS_SELECT_KERNEL", request.param)
    from cranesched_amd.engine import GpuNodeSelector
    return GpuNodeSelector


@pytest.fixture
def engine_default(built):
    """The engine class on its default selection kernel: for the kernels that never read CNS_SELECT_KERNEL (run limits,
    MultiFactorPriority, step scheduler) — running those under four settings only repeated the same case four times."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cranesched_amd.engine import GpuNodeSelector
    return GpuNodeSelector
