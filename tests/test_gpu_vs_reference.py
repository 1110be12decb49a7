"""The HIP engine through the C ABI against THE REFERENCE'S OWN CODE (oracle/_ref/libcrane_ref.so, built from slices of
/root/reference's JobScheduler.{h,cpp} / PublicHeader.{h,cpp}; tests/test_ref_pin.py, oracle/ref_build/) — no
restatement in between.  The prebuilt library travels to the GPU box with the snapshot (/root/reference itself does not
exist there and is not read).  Exact: placements, fp64 cost bit patterns, final time maps, preempted lists."""
import numpy as np
import pytest

from cranesched_amd import abi, synth
from oracle import pyoracle
from tests import helpers, kat, kat_preempt

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not pyoracle.ref_available(), reason="oracle/_ref/*.so was not shipped and /root/reference is absent")]


def _engine(c, run=None, rv=None):
    from cranesched_amd.engine import GpuNodeSelector
    eng = GpuNodeSelector(device=0)
    eng.set_nodes(c)
    if rv is not None:
        eng.set_reservations(rv)
    if run is not None:
        eng.set_running(run)
    return eng


def _same(tag, c, j, eng, got, ref):
    d = got.diff(ref.placements)
    assert d is None, f"{tag}: engine placements differ from the reference's own code at {d}"
    assert np.array_equal(eng.costs().view(np.uint64), ref.costs().view(np.uint64)), f"{tag}: fp64 costs differ"
    for n in range(c.num_nodes):
        b = ref.timeline(n)
        if len(b["t"]) == 0:      # no NodeState in the reference (node of no partition with pending jobs)
            continue
        a = eng.timeline(n)
        for f in ("t", "cpu_raw", "mem", "core_lo", "core_hi", "gres"):
            assert np.array_equal(a[f], b[f]), f"{tag}: time map of node {n} differs in {f}"


@pytest.mark.parametrize("kernel", ["wide", "pipe", "legacy"])
@pytest.mark.parametrize("scn", [s for s in kat.scenarios() if not s[3]], ids=lambda s: s[0])
def test_engine_vs_reference_code_hand_derived(built, monkeypatch, scn, kernel):
    monkeypatch.setenv("CNS_SELECT_KERNEL", kernel)
    name, c, j, cfg, expect = scn
    eng = _engine(c)
    try:
        got = eng.node_select(kat.NOW, j)
        _same(name, c, j, eng, got, pyoracle.select(c, j, kat.NOW, backend="ref"))
    finally:
        eng.close()


@pytest.mark.parametrize("kernel", ["wide", "pipe", "legacy"])
@pytest.mark.parametrize("seed", range(6))
def test_engine_vs_reference_code_random(built, monkeypatch, seed, kernel):
    monkeypatch.setenv("CNS_SELECT_KERNEL", kernel)
    c, j, now, run = helpers.random_case(40 + seed)
    eng = _engine(c, run)
    try:
        got = eng.node_select(now, j)
        _same(f"random {seed}", c, j, eng, got, pyoracle.select(c, j, now, running=run, backend="ref"))
    finally:
        eng.close()


@pytest.mark.parametrize("name,J,N,P", [("C2", 6000, 512, 1), ("C3", 5000, 640, 1), ("C4", 8000, 1024, 8), ("C5", 8000, 512, 8)])
def test_engine_vs_reference_code_scaled_configs(built, name, J, N, P):
    c, j, now = synth.make_config(name, J=J, N=N, P=P)
    eng = _engine(c)
    try:
        got = eng.node_select(now, j)
        _same(name, c, j, eng, got, pyoracle.select(c, j, now, backend="ref"))
        assert eng.last_kernel().startswith("k_wide"), eng.last_kernel()   # the default selection kernel served it
    finally:
        eng.close()


@pytest.mark.parametrize("seed", range(4))
def test_engine_vs_reference_code_reservations(built, seed):
    from tests.test_reservations import random_resv_case
    c, j, now, run, rv = random_resv_case(seed)
    eng = _engine(c, run, rv)
    try:
        got = eng.node_select(now, j)
        _same(f"resv {seed}", c, j, eng, got, pyoracle.select(c, j, now, running=run, reservations=rv, backend="ref"))
    finally:
        eng.close()


@pytest.mark.parametrize("seed", range(12))
def test_engine_vs_reference_code_preemption(built, seed):
    from tests.test_preempt import random_preempt_case
    c, j, now, run, pre = random_preempt_case(500 + seed, N=6 + seed % 7, J=50 + seed % 40, P=1 + seed % 2, running=10 + seed % 11)
    ref = pyoracle.select(c, j, now, running=run, preempt=pre, backend="ref")
    eng = _engine(c, run)
    try:
        pl, po = eng.node_select_preempt(now, j, pre)
        _same(f"preempt {seed}", c, j, eng, pl, ref)
        assert po.lists() == ref.preempt_out.lists() and po.cancelled_ids() == ref.preempt_out.cancelled_ids()
        assert po.preempting_ids() == ref.preempt_out.preempting_ids()
    finally:
        eng.close()


@pytest.mark.parametrize("seed,lay", [(11, "all+subsets"), (12, "chain"), (13, "random")])
def test_engine_vs_reference_code_shared_nodes(built, seed, lay):
    from tests.test_overlap import overlap_case
    c, j, now, run = overlap_case(seed, N=32, J=200, layout=lay)
    eng = _engine(c, run)
    try:
        got = eng.node_select(now, j)
        _same(f"overlap {seed} {lay}", c, j, eng, got, pyoracle.select(c, j, now, running=run, backend="ref"))
    finally:
        eng.close()
