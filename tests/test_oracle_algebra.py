"""Resource algebra of the oracle, pinned against the reference's own known-answer vectors
(/root/reference test/Utilities/dedicated_resource_test.cpp:27-251, restated here as data) and
cross-checked literal-containers == bit-masks on random inputs."""
import numpy as np
import pytest

from cranesched_amd import abi
from oracle import pyoracle as po

# name 0 = "GPU": class 0 "A100" (bits 0..7), class 1 "A200"/"B100" (bits 8..15); name 1 = "XPU": class 2 (bits 16..23)
L = abi.GresLayout(class_name=[0, 0, 1], class_shift=[0, 8, 16], class_width=[8, 8, 8])
ALG = [po.MASK, po.LITERAL]


def slots(cls, idx):
    m = 0
    for i in idx:
        m |= 1 << (L.class_shift[cls] + i)
    return m


def res(gres=0, cpu=0, mem=0, clo=0, chi=0):
    return po.make_res(cpu, mem, clo, chi, gres)


@pytest.mark.parametrize("alg", ALG)
def test_dedicated_le_vectors(alg):
    # le_gt :27-35 — {A100:{0,1,3,2}} <= {A100:{0,1,3}} is false
    assert not po.binop(L, alg, "le", res(slots(0, [0, 1, 3, 2])), res(slots(0, [0, 1, 3])))
    # le_lt1/2 :37-53 — {A100:{0}} <= {A100:{0,1,3}}
    assert po.binop(L, alg, "le", res(slots(0, [0])), res(slots(0, [0, 1, 3])))
    # le_lt3 :55-61 — extra type on the right does not matter
    assert po.binop(L, alg, "le", res(slots(0, [0])), res(slots(0, [0, 1, 3]) | slots(1, [0, 1, 3])))
    # le_nle :63-70 — {A100:{0,2}} <= {A100:{0,1,3}} is false (slot identity, not counts)
    assert not po.binop(L, alg, "le", res(slots(0, [0, 2])), res(slots(0, [0, 1, 3])))
    # le_equ / le_equ2 :72-88
    assert po.binop(L, alg, "le", res(slots(0, [0, 1])), res(slots(0, [0, 1])))
    both = slots(0, [0, 1]) | slots(2, [0, 1])
    assert po.binop(L, alg, "le", res(both), res(both))


@pytest.mark.parametrize("alg", ALG)
def test_dedicated_plus_minus_vectors(alg):
    a = slots(0, [0, 1, 3])
    x = slots(2, [1, 6, 5])
    # plus1 :114-119, plus2 :121-130, plus3 :132-140
    assert po.binop(L, alg, "add", res(0), res(a))[4] == a
    assert po.binop(L, alg, "add", res(x), res(a))[4] == (a | x)
    assert po.binop(L, alg, "add", res(a | x), res(a))[4] == (a | x)
    # minus1 :142-151 — removing GPU leaves XPU
    assert po.binop(L, alg, "sub", res(a | x), res(a))[4] == x
    # minus2 :153-162 — minus-to-empty erases the key: result equals the empty resource
    assert po.binop(L, alg, "sub", res(slots(0, [0])), res(slots(0, [0])))[4] == 0


@pytest.mark.parametrize("alg", ALG)
def test_request_vs_slots_vectors(alg):
    """req_map* :173-251: (untyped u, {type: c}) -> GresCount{total = u + sum(c), specified}."""
    big = dict(cpu=256, mem=1)
    four_a100 = res(slots(0, [0, 1, 3, 2]), **big)
    rq = lambda tot, spec: po.make_req(0, 0, gtot=[tot], gspec=spec)
    assert po.feasible(L, alg, rq(2, [1]), four_a100)[0]            # req_map : (1,{A100:1}) fits 4 x A100
    assert not po.feasible(L, alg, rq(5, [1]), four_a100)[0]        # req_map2: (4,{A100:1}) does not
    assert po.feasible(L, alg, rq(1, []), res(slots(0, [0]), **big))[0]  # req_map3: (1,{}) fits 1 x A100
    four_b100 = res(slots(1, [0, 1, 3, 2]), **big)
    assert not po.feasible(L, alg, rq(5, [1]), four_b100)[0]        # req_map4: A100 typed vs B100 slots
    four_xpu = res(slots(2, [0, 1, 3, 2]), **big)
    assert not po.feasible(L, alg, rq(5, [1]), four_xpu)[0]         # req_map5: wrong name


def _rand_res(rng):
    clo = int(rng.integers(0, 1 << 16)) if rng.random() > 0.2 else 0
    return po.make_res(int(rng.integers(0, 40)) * 128, int(rng.integers(0, 64)), clo,
                       int(rng.integers(0, 16)) if rng.random() < 0.2 else 0, int(rng.integers(0, 1 << 24)) & int(rng.integers(0, 1 << 24)))


def test_literal_equals_mask_random():
    rng = np.random.default_rng(7)
    for _ in range(3000):
        a, b = _rand_res(rng), _rand_res(rng)
        for op in ("ckmin", "add", "le"):
            assert po.binop(L, po.MASK, op, a, b) == po.binop(L, po.LITERAL, op, a, b), op
        # sub is only ever applied to a subset on this path
        sub = po.make_res(min(a.cpu, b.cpu), min(a.mem, b.mem), a.clo & b.clo, a.chi & b.chi, a.gres & b.gres)
        assert po.binop(L, po.MASK, "sub", a, sub) == po.binop(L, po.LITERAL, "sub", a, sub)
        q = po.make_req(int(rng.integers(0, 12)) * 128, int(rng.integers(0, 48)),
                        gtot=[int(rng.integers(0, 5)), int(rng.integers(0, 5))] if rng.random() < 0.5 else [],
                        gspec=[int(rng.integers(0, 3)), int(rng.integers(0, 3)), int(rng.integers(0, 3))] if rng.random() < 0.4 else [])
        assert po.feasible(L, po.MASK, q, a) == po.feasible(L, po.LITERAL, q, a)
