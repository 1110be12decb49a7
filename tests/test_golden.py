"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py) vs today's oracle."""
import os

import numpy as np
import pytest

from oracle import pyoracle
from tests.golden.make_golden import CASES, HERE, LIMIT_CASES, RESV_CASES, STEP_CASES, limit_outputs, step_outputs


def load(name):
    return np.load(os.path.join(HERE, name + ".npz"))


def compare(name, placements, costs):
    g = load(name)
    t = placements.trimmed()
    for k in t:
        if k not in g.files:   # fixtures made before ABI 3 (core ids 128..255): the cases have no such core, the planes are zero
            assert k in ("core_w2", "core_w3") and not t[k].any(), f"{name}: {k} is not in the golden fixture"
            continue
        assert np.array_equal(g[k], t[k]), f"{name}: {k} differs from the golden fixture"
    assert np.array_equal(g["costs"], costs.view(np.uint64)), f"{name}: fp64 costs differ"


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_golden(name):
    c, j, now, run = CASES[name]()
    r = pyoracle.select(c, j, now, running=run)
    compare(name, r.placements, r.costs())


@pytest.mark.parametrize("name", sorted(RESV_CASES))
def test_oracle_matches_golden_reservations(name):
    c, j, now, run, rv = RESV_CASES[name]()
    r = pyoracle.select(c, j, now, running=run, reservations=rv)
    compare(name, r.placements, r.costs())


@pytest.mark.parametrize("name", sorted(LIMIT_CASES))
def test_oracle_matches_golden_run_limits(name):
    out = limit_outputs(LIMIT_CASES[name]())
    g = load(name)
    for k, v in out.items():
        if k not in g.files:   # (as above)
            assert k.endswith(("core_w2", "core_w3")) and not v.any(), f"{name}: {k} is not in the golden fixture"
            continue
        assert np.array_equal(g[k], v), f"{name}: {k} differs from the golden fixture"


@pytest.mark.parametrize("name", sorted(STEP_CASES))
def test_oracle_matches_golden_steps(name):
    out = step_outputs(STEP_CASES[name])
    g = load(name)
    for k, v in out.items():
        if k not in g.files:   # (as above)
            assert k.endswith(("core_w2", "core_w3")) and not v.any(), f"{name}: {k} is not in the golden fixture"
            continue
        assert np.array_equal(g[k], v), f"{name}: {k} differs from the golden fixture"
