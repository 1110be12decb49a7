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


# ---- run-limit admission and step scheduler: the engine against the reference's own AccountMetaContainer /
# ---- JobInCtld::SchedulePendingSteps (oracle/_ref, round 4; tests/test_ref_pin_limits_steps.py pins the oracles) ----
def _limits_vs_reference(tag, cluster, jobs, now, lay, t, lj):
    from cranesched_amd import limits as lm
    eng = _engine(cluster)
    try:
        got = eng.node_select(now, jobs)
        eng.set_run_limits(t)
        reason, adm = eng.apply_run_limits(lj)
        usage = eng.usage()
        # the reference admits over the ENGINE's placements: nothing of the restated oracle is in the loop
        r_ref, a_ref, u_ref = pyoracle.run_limits(lay, t, lj, got, backend="ref")
        s_gpu, s_ref = [lm.LIMIT_REASON_STR[int(x)] for x in reason], [lm.LIMIT_REASON_STR[int(x)] for x in r_ref]
        bad = [i for i in range(len(s_gpu)) if s_gpu[i] != s_ref[i]]
        assert not bad, f"{tag}: job {bad[0]}: engine {s_gpu[bad[0]]!r}, reference {s_ref[bad[0]]!r}"
        assert adm == a_ref
        for f in usage.__dataclass_fields__:
            assert np.array_equal(getattr(usage, f), getattr(u_ref, f)), f"{tag}: usage table {f} differs from the reference's"
        return reason
    finally:
        eng.close()


@pytest.mark.parametrize("mode", ["parallel", "ordered"])
@pytest.mark.parametrize("seed,tight", [(1, True), (2, True), (3, True), (4, False), (7, True), (8, True)])
def test_engine_run_limits_vs_reference_code_random(built, monkeypatch, seed, tight, mode):
    from tests.test_run_limits import random_limit_case
    if mode == "ordered":
        monkeypatch.setenv("CNS_LIMITS_MODE", "seq")
    else:
        monkeypatch.delenv("CNS_LIMITS_MODE", raising=False)
    cluster, jobs, now, lay, t, lj = random_limit_case(seed, J=900, N=128, tight=tight)
    r = _limits_vs_reference(f"limits {seed}", cluster, jobs, now, lay, t, lj)
    if tight:
        assert set(np.unique(r)) - {0, 255}


def test_engine_run_limits_vs_reference_code_hand_derived(built):
    from tests import test_run_limits as trl
    for name in sorted(trl.SCENARIOS):
        specs, keys, ua, t, exp, extra = trl.SCENARIOS[name]()
        cluster, lay = trl._cluster()
        jobs = kat.jobs(specs)
        r = _limits_vs_reference(name, cluster, jobs, trl.NOW, lay, t, trl._limjobs(keys, ua, jobs.time_limit_sec))
        assert list(r) == exp, name


def test_engine_run_limits_vs_reference_code_c4_accounts(built):
    """BASELINE config 4's account / QoS tables, caps tightened until they bind, scaled queue."""
    c, j, now = synth.make_config("C4", J=8000, N=1024, P=8)
    tables, lj = synth.make_limits("C4", c, j)
    tables.qos["max_jobs_per_user"][:] = 3
    tables.qos["max_tres_per_account"]["cpu_raw"][:] = 200 * 256
    r = _limits_vs_reference("C4 accounts", c, j, now, c.gres, tables, lj)
    assert 0 < int((r == 0).sum()) < int((r != 255).sum())


@pytest.mark.parametrize("seed,wide", [(0, False), (1, False), (2, False), (10, True), (11, True)])
def test_engine_steps_vs_reference_code(built, seed, wide):
    from tests.test_steps import random_step_case
    lay, jobs, steps = random_step_case(seed, J=300, wide=wide)
    eng = _engine(kat.cluster([4], layout=lay))
    try:
        got, _ = eng.schedule_steps(jobs, steps)
        ref = pyoracle.schedule_steps(lay, jobs, steps, backend="ref")
        assert got.diff(ref) is None, f"steps {seed}: engine differs from the reference's SchedulePendingSteps: {got.diff(ref)}"
        assert 0 < got.scheduled[:steps.num_steps].sum() < steps.num_steps
    finally:
        eng.close()
