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

