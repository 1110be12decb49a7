"""k_wide with SEVERAL JOBS PER EXCHANGE (wide_kernel.inc, "A WINDOW OF JOBS PER EXCHANGE": the scanner waves exchange a pool of their
cheapest rows and decide up to 16 one-node jobs on it) — opt-in (CNS_WIDE_WINDOW), so the suite's other tests run without it: here the
full-size digests of the configurations whose queues open windows, and the small contended cases, with it on.  The window rule itself
against the sequential rule: tests/test_exchange_pairs_model.py."""
import os

import numpy as np
import pytest

from tests import test_gpu_fullrun as fr

pytestmark = pytest.mark.gpu


@pytest.fixture
def wide_windows(built, monkeypatch):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    monkeypatch.setenv("CNS_SELECT_KERNEL", "wide")
    monkeypatch.setenv("CNS_WIDE_WINDOW", "16")
    from cranesched_amd.engine import GpuNodeSelector
    return GpuNodeSelector


@pytest.mark.parametrize("tag", ["c2", "c4", "c5", "c5deep", "c4r", "c4v", "tile1", "tile3"])
def test_full_run_with_windows_matches_oracle_digest(wide_windows, tag, capsys):
    fr.test_full_run_matches_oracle_digest(wide_windows, tag)
    if tag in ("c2", "c5"):   # these queues do open windows (and decide most of their start-now jobs in them)
        out = capsys.readouterr().out
        dec = int(out.split("decided ")[1].split()[0])
        assert dec > 30000, out


@pytest.mark.parametrize("limit", [2, 3, 5])
def test_window_length_limits(built, monkeypatch, limit):
    """a window closes when it is full: every length gives the same placements"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    monkeypatch.setenv("CNS_SELECT_KERNEL", "wide")
    monkeypatch.setenv("CNS_WIDE_WINDOW", str(limit))
    from cranesched_amd.engine import GpuNodeSelector
    fr.test_full_run_matches_oracle_digest(GpuNodeSelector, "c2")
