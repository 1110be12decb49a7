"""The compressed PreemptSegTree (what the device runs: cranesched_amd/csrc/preempt_dev.inc) against the reference's tree
node for node, on random operation sequences: `satisfied` must agree after EVERY operation.  CPU only."""
import random

import pytest

from tests.seg_compact import CompactTree, LiteralTree, R

TICKS = 4_000_000_000


def rand_res(rng, scale=4):
    if rng.random() < 0.05:
        return R()
    return R(cpu=rng.randrange(0, scale) * 256, mem=rng.randrange(0, scale) << 30, cores=rng.getrandbits(6) if rng.random() < 0.5 else 0,
             gres=rng.getrandbits(4) if rng.random() < 0.4 else 0)


def run_case(seed, L, n_ends, n_ops, unit):
    rng = random.Random(seed)
    seg_end = L * unit
    pool = sorted({rng.randrange(-3, L + 4) * unit for _ in range(n_ends)} | {0, seg_end})
    target = R(cpu=rng.randrange(1, 5) * 256, mem=rng.randrange(0, 4) << 30, gres=rng.getrandbits(3) if rng.random() < 0.5 else 0)
    a, b = LiteralTree(0, seg_end, target), CompactTree(0, seg_end, target)
    done = []
    for i in range(n_ops):
        if done and rng.random() < 0.3:          # take an earlier range away again (the reverse pass of TryPreempt_)
            st, ed, r = done.pop(rng.randrange(len(done)))
            plus = False
        else:
            st, ed = sorted(rng.sample(pool, 2))
            r = rand_res(rng)
            plus = rng.random() < 0.85
            if plus:
                done.append((st, ed, r))
        a.op(st, ed, r, plus)
        b.op(st, ed, r, plus)
        assert a.satisfied == b.satisfied, f"seed {seed}, operation {i}: [{st}, {ed}) {'+' if plus else '-'} {r.key()}"
    return a.visits, b.visits


@pytest.mark.parametrize("unit", [TICKS, 1, 7])
@pytest.mark.parametrize("seed", range(40))
def test_compact_tree_agrees_with_the_literal_tree(seed, unit):
    run_case(seed, L=(3600, 37, 1 << 12, 600 * 24, 5)[seed % 5], n_ends=(4, 12, 30, 8)[seed % 4], n_ops=120, unit=unit)


def test_compact_tree_visits_far_fewer_nodes():
    lit = cmp = 0
    for seed in range(10):
        v = run_case(1000 + seed, L=7200, n_ends=16, n_ops=60, unit=TICKS)
        lit += v[0]; cmp += v[1]
    assert cmp * 4 < lit, (lit, cmp)
