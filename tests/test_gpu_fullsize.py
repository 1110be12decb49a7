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
