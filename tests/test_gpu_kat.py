"""HIP engine through the C ABI vs the hand-derived known answers and the committed golden fixtures."""
import numpy as np
import pytest

from tests import kat
from tests.golden.make_golden import CASES
from tests.test_golden import compare

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scn", kat.scenarios(), ids=lambda s: s[0])
def test_kat_on_gpu(engine_cls, scn):
    name, c, j, cfg, expect = scn
    eng = engine_cls(device=0, **cfg)
    try:
        eng.set_nodes(c)
        pl = eng.node_select(kat.NOW, j)
        kat.check(name, c, j, pl, expect, costs=eng.costs(), timeline=eng.timeline)
    finally:
        eng.close()


@pytest.mark.parametrize("name", sorted(CASES))
def test_golden_on_gpu(engine_cls, name):
    c, j, now, run = CASES[name]()
    eng = engine_cls(device=0)
    try:
        eng.set_nodes(c)
        if run is not None:
            eng.set_running(run)
        pl = eng.node_select(now, j)
        compare(name, pl, eng.costs())
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["limits_2", "limits_6"])
def test_golden_run_limits_on_gpu(engine_cls, name):
    """Engine (NodeSelect + run-limit admission) vs the frozen fixture: reasons and every usage table, byte for byte."""
    from tests.golden.make_golden import LIMIT_CASES
    from tests.test_golden import load
    cluster, jobs, now, lay, t, lj = LIMIT_CASES[name]()
    g = load(name)
    eng = engine_cls(device=0)
    try:
        eng.set_nodes(cluster)
        eng.node_select(now, jobs)
        eng.set_run_limits(t)
        reason, adm = eng.apply_run_limits(lj)
        usage = eng.usage()
        assert np.array_equal(g["reason"], reason) and int(g["admitted"][0]) == adm
        for f in usage.__dataclass_fields__:
            assert np.array_equal(g[f], getattr(usage, f).view(np.uint8)), f"{name}: usage table {f}"
    finally:
        eng.close()



# The reference's own request-vs-slots vectors (/root/reference test/Utilities/dedicated_resource_test.cpp:173-251,
# req_map .. req_map5: (untyped u, {type: c}) against a node's slot sets), restated as data in tests/test_oracle_algebra.py
# for the oracle — here through the DEVICE GetFeasibleResourceInNode: a one-node cluster holding exactly those slots and a
# one-task job carrying the request (GresCount{total = u + sum(c), specified}); "fits" = the job starts now on those slots,
# "does not fit" = no node fits res_total either -> pending reason "Resource" (JobScheduler.cpp:6335-6343, :6768).
REQ_MAP = [
    ("req_map", 2, [1], (0, [0, 1, 3, 2]), True, 0b0011),     # (1, {A100: 1}) vs 4 x A100: typed slot 0, then the same type serves the untyped one
    ("req_map2", 5, [1], (0, [0, 1, 3, 2]), False, 0),        # (4, {A100: 1}) vs 4 x A100
    ("req_map3", 1, [], (0, [0]), True, 0b0001),              # (1, {}) vs 1 x A100
    ("req_map4", 5, [1], (1, [0, 1, 3, 2]), False, 0),        # A100 typed vs B100 slots
    ("req_map5", 5, [1], (2, [0, 1, 3, 2]), False, 0),        # GPU request vs XPU slots (other name)
]


@pytest.mark.parametrize("vec", REQ_MAP, ids=lambda v: v[0])
def test_reference_request_vectors_on_device(engine_cls, vec):
    from cranesched_amd import abi
    name, tot, spec, (cls, idx), fits, want_slots = vec
    lay = abi.GresLayout(class_name=[0, 0, 1], class_shift=[0, 8, 16], class_width=[8, 8, 8])
    g = 0
    for i in idx:
        g |= 1 << (lay.class_shift[cls] + i)
    c = kat.cluster([4], [8], gres=[g], layout=lay)
    j = kat.jobs([dict(L=100, gtot=[tot], gspec=spec)])
    eng = engine_cls(device=0)
    try:
        eng.set_nodes(c)
        pl = eng.node_select(kat.NOW, j)
        if fits:
            assert pl.reason[0] == 0 and pl.start_sec[0] == kat.NOW and pl.node_idx[0] == 0, name
            assert int(pl.gres[0]) == want_slots, f"{name}: slots {int(pl.gres[0]):#x}"
        else:
            assert pl.reason[0] == abi.REASON_RESOURCE and pl.start_sec[0] == 0, name
    finally:
        eng.close()
