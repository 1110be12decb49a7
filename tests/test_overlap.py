"""Partitions that share nodes (JobScheduler.cpp:6563,6585-6617: ONE NodeState per craned, listed by every partition that
contains it; JobScheduler.h:498-516: one cost per partition selector).  The engine merges partitions connected through
shared nodes into one workgroup that runs their jobs in queue order: a node gets one slot (own cost) per partition, all
pointing at one time map.  Hand-derived scenario: tests/kat.py "shared_node_two_partitions"; here random clusters with
overlapping partitions (an "all nodes" partition plus subsets, chains of pairwise overlaps, disjoint rest) against the
oracle — CPU: both algebras agree; GPU: bit-exact incl. every (partition, node) cost and the shared time maps."""
import numpy as np
import pytest

from cranesched_amd import abi, synth
from tests import helpers


def overlap_case(seed: int, N: int = 64, J: int = 500, layout: str = "all+subsets"):
    c, j, now, run = helpers.random_case(seed, N=N, J=J, P=4, running=24)
    rng = np.random.default_rng(5000 + seed)
    if layout == "all+subsets":      # the routine site configuration: ALL + two subsets + one disjoint partition outside... of ALL's complement
        parts = [np.arange(N - 8), np.sort(rng.choice(N - 8, (N - 8) // 3, replace=False)),
                 np.sort(rng.choice(N - 8, (N - 8) // 4, replace=False)), np.arange(N - 8, N)]
    elif layout == "chain":          # p0 ∩ p1, p1 ∩ p2 non-empty, p0 ∩ p2 empty; p3 disjoint
        a, b = N // 4, N // 2
        parts = [np.arange(0, a + 4), np.arange(a - 4, b + 4), np.arange(b - 4, 3 * N // 4), np.arange(3 * N // 4, N)]
    else:                            # random membership: every node in 0..3 partitions
        parts = [np.nonzero(rng.random(N) < 0.45)[0] for _ in range(4)]
    off = np.cumsum([0] + [len(p) for p in parts]).astype(np.uint32)
    pn = np.concatenate(parts).astype(np.uint32)
    c = abi.Cluster(c.cpu_total_raw, c.mem_total, c.core_lo, c.core_hi, c.gres_slots, off, pn, gres=c.gres,
                    schedulable=c.schedulable)
    j.partition[:] = np.where(j.partition >= 4, j.partition, rng.integers(0, 4, j.num_jobs)).astype(np.uint32)
    return c, j, now, run


CASES = [(s, lay) for s in (1, 2, 3) for lay in ("all+subsets", "chain", "random")]


@pytest.mark.parametrize("seed,lay", CASES)
def test_oracle_algebras_agree_on_shared_nodes(built, seed, lay):
    from oracle import pyoracle
    c, j, now, run = overlap_case(seed, layout=lay)
    a = pyoracle.select(c, j, now, running=run)
    b = pyoracle.select(c, j, now, running=run, algebra=pyoracle.LITERAL)
    assert a.placements.diff(b.placements) is None
    assert np.array_equal(a.costs().view(np.uint64), b.costs().view(np.uint64))
    r = a.placements.reason[:j.num_jobs]
    assert (r == 0).sum() > 50 and (r == 1).sum() > 50, "cases must start jobs now and backfill"


@pytest.mark.gpu
@pytest.mark.parametrize("seed,lay", CASES)
def test_shared_nodes_on_gpu(engine_cls, seed, lay):
    from oracle import pyoracle
    c, j, now, run = overlap_case(seed, layout=lay)
    eng = engine_cls(device=0)
    try:
        eng.set_nodes(c)
        eng.set_running(run)
        got = eng.node_select(now, j)
        ref = pyoracle.select(c, j, now, running=run)
        helpers.assert_same(eng, got, ref, c, sample_nodes=c.num_nodes, tag=f"overlap {seed} {lay}")
    finally:
        eng.close()


@pytest.mark.gpu
def test_shared_nodes_with_reservations_on_gpu(engine_cls):
    """A reservation on a node that two partitions share: the dip / the active share shows up in both partitions' slots."""
    from oracle import pyoracle
    from tests import test_reservations
    c, j, now, run, rv = test_reservations.random_resv_case(3)
    N = c.num_nodes
    parts = [np.arange(N), np.arange(0, N, 2)]
    off = np.cumsum([0] + [len(p) for p in parts]).astype(np.uint32)
    c2 = abi.Cluster(c.cpu_total_raw, c.mem_total, c.core_lo, c.core_hi, c.gres_slots, off,
                     np.concatenate(parts).astype(np.uint32), gres=c.gres, schedulable=c.schedulable)
    j.partition[:] = np.where(j.partition >= c.num_partitions, j.partition, np.arange(j.num_jobs) % 2).astype(np.uint32)
    eng = engine_cls(device=0)
    try:
        eng.set_nodes(c2)
        eng.set_reservations(rv)
        eng.set_running(run)
        got = eng.node_select(now, j)
        ref = pyoracle.select(c2, j, now, running=run, reservations=rv)
        helpers.assert_same(eng, got, ref, c2, sample_nodes=N, tag="overlap + reservations")
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,lay,J", [(1, "all+subsets", 6000), (2, "random", 6000), (3, "all+subsets", 20000)])
def test_shared_group_wider_than_k_select_s_register_tile_on_gpu(gpu, seed, lay, J):
    """An "ALL" partition over a cluster of 12 000 nodes next to subsets of it: one group of ~19 000 - 22 000 (partition, node)
    slots, more than k_select's tile holds (16 576).  It runs on k_wide's home workgroup alone (KParams::serial_only: the
    sequential protocol with the tester waves as memory scanners) — round 3 refused it.  Bit-exact vs the oracle incl. every
    (partition, node) cost and the shared time maps; the disjoint partition beside the group keeps its fast kernel."""
    from cranesched_amd.engine import GpuNodeSelector
    from oracle import pyoracle
    c, j, now, run = overlap_case(seed, N=12000, J=J, layout=lay)
    assert int(c.part_offsets[-1]) > 16_576
    ref = pyoracle.select(c, j, now, running=run)
    eng = GpuNodeSelector(device=0)
    try:
        eng.set_nodes(c)
        eng.set_running(run)
        got = eng.node_select(now, j)
        assert "k_mem" in eng.last_kernel(), eng.last_kernel()
        helpers.assert_same(eng, got, ref, c, sample_nodes=600, tag=f"wide group {seed} {lay}")
    finally:
        eng.close()


@pytest.mark.gpu
def test_one_cycle_on_three_kernels_wide_group_small_group_plain_partition(gpu):
    """One snapshot, one cycle: a group too wide for k_select (ALL over 12 000 nodes + two subsets -> k_mem), a small group (two
    partitions sharing 1 000 nodes -> k_select) and a plain partition (-> k_wide), launched side by side; every result against
    the oracle."""
    from cranesched_amd.engine import GpuNodeSelector
    from oracle import pyoracle
    N = 24000
    c, j, now, run = helpers.random_case(11, N=N, J=9000, P=6, running=60)
    rng = np.random.default_rng(777)
    parts = [np.arange(12000), np.sort(rng.choice(12000, 4000, replace=False)), np.sort(rng.choice(12000, 3000, replace=False)),
             np.arange(12000, 18000), np.arange(18000, 24000), np.arange(23000, 24000)]
    off = np.cumsum([0] + [len(p) for p in parts]).astype(np.uint32)
    c = abi.Cluster(c.cpu_total_raw, c.mem_total, c.core_lo, c.core_hi, c.gres_slots, off, np.concatenate(parts).astype(np.uint32),
                    gres=c.gres, schedulable=c.schedulable)
    j.partition[:] = np.where(j.partition >= 6, j.partition, rng.integers(0, 6, j.num_jobs)).astype(np.uint32)
    ref = pyoracle.select(c, j, now, running=run)
    eng = GpuNodeSelector(device=0)
    try:
        eng.set_nodes(c)
        eng.set_running(run)
        got = eng.node_select(now, j)
        k = eng.last_kernel()
        assert k.startswith("k_wide") and "k_select" in k and "k_mem" in k, k
        helpers.assert_same(eng, got, ref, c, sample_nodes=600, tag="three kernels")
    finally:
        eng.close()
