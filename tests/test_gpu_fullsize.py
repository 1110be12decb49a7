"""BASELINE.json's full sizes (1 M jobs x 64 k nodes), where the CPU oracle is too slow to be the
checker: size-independent properties of the sequential algorithm instead.
  * prefix closure: job j's decision depends only on jobs < j, so the first K results of the full run
    equal a run on the first K jobs — and THAT run is checked bit-exact against the oracle;
  * conservation: replaying all placements, no node is ever over-subscribed in cpu, mem or GRES slots
    (UpdateResourceInNode asserts res <= available on every commit, JobScheduler.h:383,440);
  * every node of a placement belongs to the job's partition; start >= now; run-to-run determinism."""
import zlib

import numpy as np
import pytest

from cranesched_amd import abi, synth

pytestmark = pytest.mark.gpu

K = 60_000


def crc(pl):
    c = 0
    for a in pl.trimmed().values():
        c = zlib.crc32(np.ascontiguousarray(a).view(np.uint8), c)
    return c


def replay_ok(cluster, jobs, pl):
    J = jobs.num_jobs
    k = jobs.node_num.astype(np.int64)
    job_of = np.repeat(np.arange(J), k)
    node = pl.node_idx[:len(job_of)].astype(np.int64)
    live = node != abi.NODE_NONE
    job_of, node = job_of[live], node[live]
    start = pl.start_sec[job_of]
    end = start + jobs.time_limit_sec[job_of]
    assert (start >= synth.NOW).all()
    part = jobs.partition[job_of].astype(np.int64)
    assert ((node >= cluster.part_offsets[part]) & (node < cluster.part_offsets[part + 1])).all(), "node outside partition"
    gres = pl.gres[:len(live)][live]
    quantities = {"cpu": (pl.cpu_raw[:len(live)][live].astype(np.int64), cluster.cpu_total_raw),
                  "mem": (pl.mem[:len(live)][live].astype(np.int64), cluster.mem_total.astype(np.int64))}
    for g in range(len(cluster.gres.class_name)):
        m = np.uint64(cluster.gres.class_mask(g))
        cnt = np.array([bin(int(x)).count("1") for x in np.unique(gres & m)])
        lut = dict(zip(np.unique(gres & m).tolist(), cnt.tolist()))
        quantities[f"gres{g}"] = (np.vectorize(lut.get)((gres & m)).astype(np.int64),
                                  np.vectorize(lambda x: bin(int(x)).count("1"))(cluster.gres_slots & m).astype(np.int64))
    n_ev = len(node)
    ev_node = np.concatenate([node, node])
    ev_time = np.concatenate([start, end])
    ev_kind = np.concatenate([np.ones(n_ev, np.int8), np.zeros(n_ev, np.int8)])  # releases (0) sort first
    order = np.lexsort((ev_kind, ev_time, ev_node))
    seg_start = np.r_[True, ev_node[order][1:] != ev_node[order][:-1]]
    for name, (amount, total) in quantities.items():
        delta = np.concatenate([amount, -amount])[order]
        run = np.cumsum(delta)
        base = np.maximum.accumulate(np.where(seg_start, run - delta, -(1 << 62)))
        used = run - base
        assert (used <= total[ev_node[order]]).all(), f"{name} over-subscribed"
        assert (used >= 0).all()
    return True


def test_c4_full_size(engine_cls):
    from oracle import pyoracle
    cluster, jobs, now = synth.make_config("C4")
    eng = engine_cls(device=0)
    try:
        eng.set_nodes(cluster)
        full = eng.node_select(now, jobs)
        t_full = eng.timing()["select_ms"]
        c1 = crc(full)
        again = eng.node_select(now, jobs)
        assert crc(again) == c1, "run-to-run nondeterminism"
        assert replay_ok(cluster, jobs, full)
        # prefix closure + oracle on the prefix
        _, pre_jobs, _ = synth.make_config("C4", J=K)
        pre = eng.node_select(now, pre_jobs)
        ref = pyoracle.select(cluster, pre_jobs, now)
        assert pre.diff(ref.placements) is None, f"prefix differs from the oracle: {pre.diff(ref.placements)}"
        nrec = int(pre_jobs.node_num.astype(np.int64).sum())
        assert np.array_equal(full.start_sec[:K], pre.start_sec[:K]) and np.array_equal(full.reason[:K], pre.reason[:K])
        for f in ("node_idx", "ntasks", "cpu_raw", "mem", "core_lo", "core_hi", "gres"):
            assert np.array_equal(getattr(full, f)[:nrec], getattr(pre, f)[:nrec]), f
        print(f"C4 full: k_select {t_full:.1f} ms, {1e3 * jobs.num_jobs / t_full:.0f} decisions/s, "
              f"start-now {(full.reason[:jobs.num_jobs] == 0).sum()}, backfilled {(full.reason[:jobs.num_jobs] == 1).sum()}")
    finally:
        eng.close()


def test_c5_walltimes_full_partition_width(engine_cls):
    # C5 at full node count, reduced queue: 8192-node partitions (9 nodes per scanner lane), deep backfill
    from oracle import pyoracle
    cluster, jobs, now = synth.make_config("C5", J=400_000)
    eng = engine_cls(device=0)
    try:
        eng.set_nodes(cluster)
        got = eng.node_select(now, jobs)
        assert replay_ok(cluster, jobs, got)
        _, pre_jobs, _ = synth.make_config("C5", J=K)
        ref = pyoracle.select(cluster, pre_jobs, now)
        pre = eng.node_select(now, pre_jobs)
        assert pre.diff(ref.placements) is None
    finally:
        eng.close()


@pytest.mark.parametrize("N,J,Kpre", [(32_768, 300_000, 60_000), (65_536, 1_000_000, 100_000)])
def test_single_partition_wider_than_the_register_tiles_of_k_select(engine_default, N, J, Kpre):
    """SURVEY 8(d)'s "single 64 k partition variant" (3 145 832 algorithmic bytes per decision) and its half: ONE partition of
    32 768 / 65 536 nodes — more than k_select's 16 576 slots — on k_wide with 8 / 16 tile rows per scanner lane (the home
    workgroup's last-task table in HBM, multi-word row masks in the serial protocol).  Checked as the full-size C4 run: bit-exact
    against the oracle on a prefix (prefix closure), conservation by replay, determinism."""
    from oracle import pyoracle
    cluster, jobs, now = synth.make_config("C4", J=J, N=N, P=1)
    eng = engine_default(device=0)
    try:
        eng.set_nodes(cluster)
        full = eng.node_select(now, jobs)
        t_full = eng.timing()["select_ms"]
        assert eng.last_kernel().startswith("k_wide<16>" if N > 32_768 else "k_wide<8>"), eng.last_kernel()
        c1 = crc(full)
        assert crc(eng.node_select(now, jobs)) == c1, "run-to-run nondeterminism"
        assert replay_ok(cluster, jobs, full)
        _, pre_jobs, _ = synth.make_config("C4", J=Kpre, N=N, P=1)
        pre = eng.node_select(now, pre_jobs)
        ref = pyoracle.select(cluster, pre_jobs, now)
        assert pre.diff(ref.placements) is None, f"prefix differs from the oracle: {pre.diff(ref.placements)}"
        assert np.array_equal(eng.costs().view(np.uint64), ref.costs().view(np.uint64)), "fp64 costs of the prefix run differ"
        nrec = int(pre_jobs.node_num.astype(np.int64).sum())
        assert np.array_equal(full.start_sec[:Kpre], pre.start_sec[:Kpre]) and np.array_equal(full.reason[:Kpre], pre.reason[:Kpre])
        for f in ("node_idx", "ntasks", "cpu_raw", "mem", "core_lo", "core_hi", "gres"):
            assert np.array_equal(getattr(full, f)[:nrec], getattr(pre, f)[:nrec]), f
        r = full.reason[:jobs.num_jobs]
        bytes_per = N * 48 + 104
        print(f"one partition of {N} nodes, {J} jobs: {eng.last_kernel()} {t_full:.1f} ms = {1e3 * J / t_full:.0f} decisions/s "
              f"= {J * bytes_per / (t_full * 1e-3) / 8e12:.3f} of the HBM roofline at {bytes_per} B per decision; "
              f"start-now {(r == 0).sum()}, later {(r == 1).sum() + (r == 2).sum()}; wide_stats {eng.wide_stats()}")
    finally:
        eng.close()


@pytest.mark.parametrize("seed,N", [(21, 20_000), (22, 40_000)])
def test_wide_tile_heterogeneous_serial_paths(engine_default, seed, N):
    """Exclusive jobs, ntasks > node_num, node lists, wide multi-node jobs (the serial protocol inside the home workgroup, its
    memory scanners with 45 / 90 rows per lane: multi-word row masks) on partitions of 20 000 / 40 000 unequal nodes with running
    jobs — whole result, every cost and a node sample's time maps against the oracle."""
    from oracle import pyoracle
    from tests import helpers
    c, j, now, run = helpers.random_case(seed, N=N, J=2500, P=1, running=3000)
    eng = engine_default(device=0)
    try:
        eng.set_nodes(c)
        eng.set_running(run)
        got = eng.node_select(now, j)
        assert eng.last_kernel().startswith("k_wide<16>" if N > 32_768 else "k_wide<8>"), eng.last_kernel()
        ref = pyoracle.select(c, j, now, running=run)
        helpers.assert_same(eng, got, ref, c, sample_nodes=64, tag=f"wide tile {N}")
        assert eng.wide_stats()["serial_jobs"] > 50
    finally:
        eng.close()


def c4_with_all_partition(J):
    """synth.MIXED "C4all64k": C4's cluster and queue plus an "ALL" partition over all 65 536 nodes (one group of 131 072 slots)."""
    return synth.make_mixed("C4all64k", J=J)[:3]


def test_c4_with_an_all_partition_over_the_whole_cluster(built):
    """c4all64k: round 3 refused this snapshot (a group above k_select's 16 576 slots).  Now the group runs on k_wide's home workgroup
    alone (k_mem: the sequential protocol, tester waves scanning the committed HBM arrays): the first 60 000 jobs of C4's queue on
    the full 65 536-node cluster, whole result and every (partition, node) cost against the oracle, time maps of a node sample."""
    from cranesched_amd.engine import GpuNodeSelector
    from oracle import pyoracle
    from tests import helpers
    c, j, now = c4_with_all_partition(60_000)
    assert int(c.part_offsets[-1]) == 131_072
    ref = pyoracle.select(c, j, now)
    eng = GpuNodeSelector(device=0)
    try:
        eng.set_nodes(c)
        got = eng.node_select(now, j)
        t = eng.timing()["select_ms"]
        assert eng.last_kernel().startswith("k_mem"), eng.last_kernel()
        helpers.assert_same(eng, got, ref, c, sample_nodes=256, tag="c4all64k")
        print(f"c4all64k: {j.num_jobs} jobs x {c.num_nodes} nodes, one group of {int(c.part_offsets[-1])} slots: {eng.last_kernel()} "
              f"{t:.1f} ms = {1e3 * t / j.num_jobs:.1f} us per decision")
    finally:
        eng.close()
