#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz: inputs are re-derived from seeds, outputs come from the CPU oracle
(mask algebra, cross-checked against the literal-container algebra before writing).

    python tests/golden/make_golden.py

The reference itself cannot be built or imported here, so these are ORACLE outputs
frozen as regression fixtures — they pin the engine and the oracle against silent drift, they are
not outputs of the reference binary.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cranesched_amd import synth  # noqa: E402
from oracle import pyoracle  # noqa: E402
from tests import helpers  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    "c1_full": lambda: synth.make_config("C1") + (None,),
    "c2_small": lambda: synth.make_config("C2", J=5000, N=128) + (None,),
    "c4_small": lambda: synth.make_config("C4", J=6000, N=512, P=8) + (None,),
    "c5_small": lambda: synth.make_config("C5", J=6000, N=256, P=8) + (None,),
    "hetero_3": lambda: helpers.random_case(3),
    "hetero_5": lambda: helpers.random_case(5),
}

# cases with reservations: (cluster, jobs, now, running, reservations)
def _resv_case(seed):
    from tests import test_reservations
    return test_reservations.random_resv_case(seed)


RESV_CASES = {"resv_1": lambda: _resv_case(1), "resv_4": lambda: _resv_case(4)}


# run-limit admission (tests/test_run_limits.py::random_limit_case): reasons + usage tables of the oracle's pass
def _limit_case(seed, tight):
    from tests import test_run_limits
    return test_run_limits.random_limit_case(seed, J=700, N=96, tight=tight)


LIMIT_CASES = {"limits_2": lambda: _limit_case(2, True), "limits_6": lambda: _limit_case(6, False)}


def limit_outputs(case):
    cluster, jobs, now, lay, t, lj = case
    sel = pyoracle.select(cluster, jobs, now)
    reason, adm, usage = pyoracle.run_limits(lay, t, lj, sel.placements)
    out = {"reason": reason, "admitted": np.array([adm], np.uint64)}
    for f in usage.__dataclass_fields__:
        out[f] = getattr(usage, f).view(np.uint8)
    return out


# step scheduler (tests/test_steps.py::random_step_case): every output array of the oracle's pass
STEP_CASES = {"steps_1": 1, "steps_2": 2}


def step_outputs(seed):
    from tests import test_steps
    lay, jobs, steps = test_steps.random_step_case(seed, J=200)
    res = pyoracle.schedule_steps(lay, jobs, steps)
    lit = pyoracle.schedule_steps(lay, jobs, steps, pyoracle.LITERAL)
    assert res.diff(lit) is None
    return {f: getattr(res, f) for f in res.FIELDS}


def main():
    for name, make in CASES.items():
        c, j, now, run = make()
        r = pyoracle.select(c, j, now, running=run)
        r2 = pyoracle.select(c, j, now, running=run, algebra=pyoracle.LITERAL)
        assert r.placements.diff(r2.placements) is None, name
        t = r.placements.trimmed()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), costs=r.costs().view(np.uint64), **t)
        print(name, j.num_jobs, "jobs", np.bincount(t["reason"], minlength=3))
    for name, make in RESV_CASES.items():
        c, j, now, run, rv = make()
        r = pyoracle.select(c, j, now, running=run, reservations=rv)
        r2 = pyoracle.select(c, j, now, running=run, reservations=rv, algebra=pyoracle.LITERAL)
        assert r.placements.diff(r2.placements) is None, name
        t = r.placements.trimmed()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), costs=r.costs().view(np.uint64), **t)
        print(name, j.num_jobs, "jobs", np.bincount(t["reason"], minlength=7))
    for name, make in LIMIT_CASES.items():
        out = limit_outputs(make())
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, len(out["reason"]), "jobs", int(out["admitted"][0]), "admitted")
    for name, seed in STEP_CASES.items():
        out = step_outputs(seed)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, int(out["scheduled"].sum()), "steps scheduled")


if __name__ == "__main__":
    main()
