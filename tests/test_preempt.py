"""Preemption inside the cycle: the CPU oracle's restatement of TryPreempt_ / PreemptSegTree against
hand-derived scenarios (tests/kat_preempt.py), both resource algebras."""
import numpy as np
import pytest

from cranesched_amd import abi
from tests import kat, kat_preempt


def check(run, jobs, cluster, expect, name):
    pl = run.placements
    for j, e in expect.items():
        if not isinstance(j, int):
            continue
        reason, start, recs = e
        assert int(pl.reason[j]) == reason, (name, j, "reason", int(pl.reason[j]))
        assert int(pl.start_sec[j]) == start, (name, j, "start", int(pl.start_sec[j]))
        o = int(pl.place_offsets[j])
        got = [(int(pl.node_idx[o + i]), int(pl.ntasks[o + i]), int(pl.cpu_raw[o + i]), int(pl.core_lo[o + i]), int(pl.gres[o + i]))
               for i in range(int(pl.place_offsets[j + 1]) - o) if int(pl.node_idx[o + i]) != abi.NODE_NONE]
        assert got == recs, (name, j, got)
    po = run.preempt_out
    lists = po.lists()
    for j, want in expect["preempted"].items():
        assert lists[j] == want, (name, j, lists[j])
    assert po.cancelled_ids() == expect["cancelled"], (name, po.cancelled_ids())
    assert po.preempting_ids() == expect["preempting"], (name, po.preempting_ids())
    if "costs" in expect:
        assert np.array_equal(run.costs(), np.asarray(expect["costs"], np.float64)), (name, run.costs())
    for node, tl in expect.get("timeline", {}).items():
        m = run.timeline(node)
        got = [(int(t), int(c), int(lo)) for t, c, lo in zip(m["t"], m["cpu_raw"], m["core_lo"])]
        assert got == tl, (name, node, got)


@pytest.mark.parametrize("algebra", [0, 1])
@pytest.mark.parametrize("scn", kat_preempt.scenarios(), ids=lambda s: s[0])
def test_oracle_preempt_known_answers(built, scn, algebra):
    from oracle import pyoracle
    name, cluster, jobs, running, pre, expect = scn
    run = pyoracle.select(cluster, jobs, kat_preempt.NOW, running=running, algebra=algebra, preempt=pre)
    check(run, jobs, cluster, expect, name)


def test_disabled_preemption_is_the_plain_cycle(built):
    """PreemptType == NONE: the same inputs go through Backfill_ (JobScheduler.cpp:6140)."""
    from oracle import pyoracle
    name, cluster, jobs, running, pre, _ = kat_preempt.scenarios()[0]
    pre.enabled = False
    a = pyoracle.select(cluster, jobs, kat_preempt.NOW, running=running, preempt=pre)
    b = pyoracle.select(cluster, jobs, kat_preempt.NOW, running=running)
    assert a.placements.diff(b.placements) is None
    assert int(a.placements.start_sec[0]) == 1500 and a.preempt_out.lists() == [[]]


# ---------------------------------------------------------------------------------------------------------------------
# The C++ oracle against the independent Python restatement (tests/select_pyref.py: PreemptCycle, SegTree)
# ---------------------------------------------------------------------------------------------------------------------
def run_pyref_preempt(c, j, now, run, pre, max_job_num_per_node=0, max_time_window_sec=0, rv=None):
    from tests import select_pyref as pr
    from tests.test_select_pyref import _res
    lay = c.gres
    N = c.num_nodes
    chi = c.core_hi if c.core_hi is not None else np.zeros(N, np.uint64)
    gs = c.gres_slots if c.gres_slots is not None else np.zeros(N, np.uint64)
    totals = [_res(lay, c.cpu_total_raw[n], c.mem_total[n], c.core_lo[n], chi[n], gs[n]) for n in range(N)]
    parts = [list(map(int, c.part_nodes[c.part_offsets[p]:c.part_offsets[p + 1]])) for p in range(c.num_partitions)]
    types_of = lambda name: [g for g in range(len(lay.class_name)) if lay.class_name[g] == name]
    resvs, pend = [], set()
    jresv = j.reservation if getattr(j, "reservation", None) is not None else None
    if rv is not None:
        resvs = [dict(start=int(rv.start_sec[v]), end=int(rv.end_sec[v]),
                      allocs=[(int(rv.alloc_node[a]), _res(lay, rv.alloc_cpu_raw[a], rv.alloc_mem[a], rv.alloc_core_lo[a], rv.alloc_core_hi[a], rv.alloc_gres[a]))
                              for a in range(int(rv.alloc_offsets[v]), int(rv.alloc_offsets[v + 1]))]) for v in range(len(rv.start_sec))]
        pend = {int(x) for x in jresv if x != abi.RESV_NONE} if jresv is not None else set()
    cyc = pr.PreemptCycle(now, totals, parts, schedulable=None if c.schedulable is None else list(c.schedulable), types_of=types_of,
                          max_jobs_per_node=max_job_num_per_node or pr.MAX_JOBS_PER_NODE, max_window=max_time_window_sec or pr.MAX_WINDOW,
                          qos_preempt=pre.qos_preempt if pre.enabled else [], preempting=[int(x) for x in pre.preempting],
                          reservations=resvs, pending_resv=pend)
    if run is not None:
        ahi = run.alloc_core_hi if run.alloc_core_hi is not None else np.zeros(len(run.alloc_node), np.uint64)
        ag = run.alloc_gres if run.alloc_gres is not None else np.zeros(len(run.alloc_node), np.uint64)
        for r in range(len(run.end_sec)):
            al = [(int(run.alloc_node[a]), _res(lay, run.alloc_cpu_raw[a], run.alloc_mem[a], run.alloc_core_lo[a], ahi[a], ag[a]))
                  for a in range(int(run.alloc_offsets[r]), int(run.alloc_offsets[r + 1]))]
            rres = None
            if getattr(run, "reservation", None) is not None and run.reservation[r] != abi.RESV_NONE:
                rres = int(run.reservation[r])
            cyc.add_running_job(int(pre.rn_job_id[r]), int(pre.rn_qos[r]), int(pre.rn_qos_priority[r]), int(pre.rn_start_sec[r]),
                                int(run.end_sec[r]), al, resv=rres)
    cyc.start()
    out, lists = [], []
    for i in range(j.num_jobs):
        if j.skip is not None and j.skip[i]:
            out.append([abi.REASON_SKIPPED, 0, []]); lists.append([])
            continue
        jv = None if jresv is None or jresv[i] == abi.RESV_NONE else int(jresv[i])
        if jv is None and j.partition[i] >= c.num_partitions:
            out.append([abi.REASON_PARTITION_NOT_FOUND, 0, []]); lists.append([])
            continue
        gtot = {a: int(v) for a, v in enumerate(j.gres_total[i]) if v} if j.gres_total is not None else {}
        gspec = {(lay.class_name[g], g): int(v) for g, v in enumerate(j.gres_spec[i]) if v} if j.gres_spec is not None else {}
        node_view = pr.Req(int(j.node_cpu_raw[i]) if j.node_cpu_raw is not None else 0, int(j.node_mem[i]), gtot, gspec)
        job = dict(part=int(j.partition[i]), L=int(j.time_limit_sec[i]), k=int(j.node_num[i]), ntasks=int(j.ntasks[i]),
                   tmin=int(j.ntasks_per_node_min[i]), tmax=int(j.ntasks_per_node_max[i]), tcpu=int(j.task_cpu_raw[i]),
                   tmem=int(j.task_mem[i]), node_view=node_view, exclusive=bool(j.exclusive[i]) if j.exclusive is not None else False,
                   incl=set(map(int, j.incl_nodes[int(j.incl_offsets[i]):int(j.incl_offsets[i + 1])])) if j.incl_offsets is not None else set(),
                   excl=set(map(int, j.excl_nodes[int(j.excl_offsets[i]):int(j.excl_offsets[i + 1])])) if j.excl_offsets is not None else set())
        job["min_view"] = pr.compose(node_view, job["tcpu"], job["tmem"], job["tmin"])
        job["resv"] = jv
        jinfo = dict(qos=int(pre.pd_qos[i]), qprio=int(pre.pd_qos_priority[i]), prio=float(pre.pd_priority[i]))
        reason, start, picks, refs = cyc.run_job_p(i, job, jinfo)
        out.append([reason, start, picks])
        lists.append([(kind == "pd", x) for kind, x in refs])
    for i in range(j.num_jobs):          # a job preempted later in the cycle carries "Preempted"
        if i in cyc.pd and cyc.pd[i]["reason"] == 7:
            out[i][0] = 7
    return cyc, [tuple(o) for o in out], lists


def compare_preempt(tag, c, j, ref, cyc, out, lists):
    from tests.test_select_pyref import compare
    compare(tag, c, j, ref, cyc, out)
    po = ref.preempt_out
    assert po.lists() == lists, f"{tag}: preempted lists {po.lists()} (oracle) vs {lists} (python)"
    assert po.cancelled_ids() == cyc.cancelled, f"{tag}: cancelled {po.cancelled_ids()} vs {cyc.cancelled}"
    assert po.preempting_ids() == sorted(cyc.preempting), f"{tag}: preempting set"
    for n in range(c.num_nodes):         # final time maps, entry by entry
        if n not in cyc.nodes:
            continue
        m = ref.timeline(n)
        got = [(int(t), int(cpu), int(mem)) for t, cpu, mem in zip(m["t"], m["cpu_raw"], m["mem"])]
        want = [(t, r.cpu, r.mem) for t, r in cyc.nodes[n].tmap]
        assert got == want, f"{tag}: time map of node {n}: {got} (oracle) vs {want} (python)"


@pytest.mark.parametrize("scn", kat_preempt.scenarios(), ids=lambda s: s[0])
def test_python_restatement_on_hand_derived_preempt_scenarios(built, scn):
    from oracle import pyoracle
    name, c, j, r, pre, expect = scn
    cyc, out, lists = run_pyref_preempt(c, j, kat_preempt.NOW, r, pre)
    ref = pyoracle.select(c, j, kat_preempt.NOW, running=r, preempt=pre)
    compare_preempt(name, c, j, ref, cyc, out, lists)


def random_preempt_case(seed, N=10, J=60, P=1, running=14, nq=3):
    """A loaded little cluster with three QoS levels (2 may preempt 1 and 0, 1 may preempt 0), distinct (qos_priority,
    start) / (qos_priority, priority) keys so that the reference's comparator leaves nothing unordered."""
    from tests import helpers
    c, j, now, run = helpers.random_case(seed, N=N, J=J, P=P, running=running)
    rng = np.random.default_rng(seed * 7919 + 13)
    R = len(run.end_sec)
    rn_qos = rng.integers(0, nq, R)
    pd_qos = rng.integers(0, nq, j.num_jobs)
    qprio = np.array([10, 20, 30])
    rn_start = now - 1 - rng.permutation(R) * 7            # all different
    pd_prio = rng.permutation(j.num_jobs).astype(np.float64) + 0.5
    preempting = [int(1000 + r) for r in range(R) if rng.random() < 0.15] + [4242]
    pre = abi.Preempt([[], [0], [1, 0]][:nq], np.arange(j.num_jobs) + 1, pd_qos, qprio[pd_qos], pd_prio,
                      1000 + np.arange(R), rn_qos, qprio[rn_qos], rn_start, preempting=preempting)
    return c, j, now, run, pre


@pytest.mark.parametrize("seed", range(60))
def test_python_restatement_on_random_preempt_cases(built, seed):
    from oracle import pyoracle
    c, j, now, run, pre = random_preempt_case(500 + seed, N=6 + seed % 7, J=50 + seed % 40, P=1 + seed % 2, running=10 + seed % 11)
    cyc, out, lists = run_pyref_preempt(c, j, now, run, pre)
    ref = pyoracle.select(c, j, now, running=run, preempt=pre)
    compare_preempt(f"random preempt {seed}", c, j, ref, cyc, out, lists)
    if seed == 0:
        # the generator must actually exercise the path
        tot = sum(len(x) for s in range(500, 520) for x in pyoracle.select(*random_preempt_case(s)[:3], running=random_preempt_case(s)[3], preempt=random_preempt_case(s)[4]).preempt_out.lists())
        assert tot > 20, f"only {tot} preemptions in 20 random cases"


# ---------------------------------------------------------------------------------------------------------------------
# The HIP engine (cns_select_preempt: k_select's general path + csrc/preempt_dev.inc) against the oracle
# ---------------------------------------------------------------------------------------------------------------------
def run_engine_preempt(c, j, now, run, pre, resv=None, **cfg):
    from cranesched_amd.engine import GpuNodeSelector
    eng = GpuNodeSelector(device=0, **cfg)
    eng.set_nodes(c)
    if resv is not None:
        eng.set_reservations(resv)
    eng.set_running(run)
    pl, po = eng.node_select_preempt(now, j, pre)
    return eng, pl, po


def compare_engine(tag, c, j, ref, eng, pl, po):
    d = pl.diff(ref.placements)
    assert d is None, f"{tag}: engine placements differ from the oracle: {d}"
    rpo = ref.preempt_out
    assert po.lists() == rpo.lists(), f"{tag}: preempted lists {po.lists()} (engine) vs {rpo.lists()} (oracle)"
    assert po.cancelled_ids() == rpo.cancelled_ids(), f"{tag}: cancelled {po.cancelled_ids()} vs {rpo.cancelled_ids()}"
    assert po.preempting_ids() == rpo.preempting_ids(), f"{tag}: preempting set"
    assert np.array_equal(eng.costs().view(np.uint64), ref.costs().view(np.uint64)), f"{tag}: fp64 costs differ"
    in_part = np.zeros(c.num_nodes, bool)
    in_part[np.asarray(c.part_nodes, np.int64)] = True
    for n in range(c.num_nodes):
        if not in_part[n]:   # a node in no partition has no NodeState at all (JobScheduler.cpp:6584-6606)
            continue
        a, b = eng.timeline(n), ref.timeline(n)
        for f in ("t", "cpu_raw", "mem", "core_lo", "core_hi", "gres"):
            assert np.array_equal(a[f], b[f]), f"{tag}: time map of node {n}, field {f}: {a[f]} vs {b[f]}"


@pytest.mark.gpu
@pytest.mark.parametrize("scn", kat_preempt.scenarios(), ids=lambda s: s[0])
def test_engine_preempt_known_answers(gpu, scn):
    from oracle import pyoracle
    name, c, j, r, pre, expect = scn
    ref = pyoracle.select(c, j, kat_preempt.NOW, running=r, preempt=pre)
    eng, pl, po = run_engine_preempt(c, j, kat_preempt.NOW, r, pre)
    try:
        compare_engine(name, c, j, ref, eng, pl, po)
        has_list = any(len(pre.qos_preempt[int(q)]) for q in pre.pd_qos)
        assert eng.last_kernel().startswith("k_select") == has_list   # (no list anywhere: the plain cycle on the fast kernels)
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(40))
def test_engine_preempt_random_cases(gpu, seed):
    from oracle import pyoracle
    c, j, now, run, pre = random_preempt_case(500 + seed, N=6 + seed % 7, J=50 + seed % 40, P=1 + seed % 2, running=10 + seed % 11)
    ref = pyoracle.select(c, j, now, running=run, preempt=pre)
    eng, pl, po = run_engine_preempt(c, j, now, run, pre)
    try:
        compare_engine(f"random preempt {seed}", c, j, ref, eng, pl, po)
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(0, 40, 3))
def test_engine_preempt_on_the_node_for_node_trees(gpu, seed, monkeypatch):
    """TryPreempt_'s trees run in a compressed form on the device (tests/seg_compact.py); a call that needs more records than fit
    is redone node for node.  CNS_PREEMPT_TREE=literal sends every call down that path, =tiny leaves the compressed form room for
    a handful of records, so that calls run out of them half way and start again."""
    monkeypatch.setenv("CNS_PREEMPT_TREE", "literal" if seed % 3 == 0 and seed % 2 == 0 else "tiny")
    from oracle import pyoracle
    if seed % 2:
        c, j, now, run, pre = random_preempt_case(900 + seed, N=48 + 8 * (seed % 6), J=500, P=2, running=120)
    else:
        c, j, now, run, pre = random_preempt_case(500 + seed, N=6 + seed % 7, J=50 + seed % 40, P=1 + seed % 2, running=10 + seed % 11)
    ref = pyoracle.select(c, j, now, running=run, preempt=pre)
    eng, pl, po = run_engine_preempt(c, j, now, run, pre)
    try:
        compare_engine(f"literal trees {seed}", c, j, ref, eng, pl, po)
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_engine_preempt_larger_cases(gpu, seed):
    """More nodes, multi-partition, many running jobs: long candidate lists, deep trees, several releases per job."""
    from oracle import pyoracle
    c, j, now, run, pre = random_preempt_case(900 + seed, N=48 + 8 * seed, J=500, P=2, running=120)
    ref = pyoracle.select(c, j, now, running=run, preempt=pre)
    eng, pl, po = run_engine_preempt(c, j, now, run, pre)
    try:
        compare_engine(f"larger preempt {seed}", c, j, ref, eng, pl, po)
        assert sum(len(x) for x in po.lists()) > 0
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_engine_preempt_on_nodes_with_192_and_256_cores(gpu, seed):
    """Preemption (releases, the segment trees' Res records, the placement records read back at a release) with core ids 128..255."""
    from oracle import pyoracle
    from tests import helpers
    c, j, now, run, pre = random_preempt_case(1300 + seed, N=24 + 8 * seed, J=400, P=1 + seed % 2, running=60)
    c = helpers.widen_cores(c, seed)
    ref = pyoracle.select(c, j, now, running=run, preempt=pre)
    if pyoracle.ref_available():   # ... and the reference's own code agrees with the oracle on this case
        r2 = pyoracle.select(c, j, now, running=run, preempt=pre, backend="ref")
        assert r2.placements.diff(ref.placements) is None and r2.preempt_out.lists() == ref.preempt_out.lists()
    eng, pl, po = run_engine_preempt(c, j, now, run, pre)
    try:
        compare_engine(f"preempt wide cores {seed}", c, j, ref, eng, pl, po)
        assert sum(len(x) for x in po.lists()) > 0 and (pl.core_w2 != 0).any()
    finally:
        eng.close()


@pytest.mark.gpu
def test_engine_preempt_disabled_is_the_plain_cycle(gpu):
    from oracle import pyoracle
    c, j, now, run, pre = random_preempt_case(777, N=12, J=80, P=1, running=16)
    pre.enabled = False
    pre.preempting = np.zeros(0, np.uint32)
    eng, pl, po = run_engine_preempt(c, j, now, run, pre)
    try:
        ref = pyoracle.select(c, j, now, running=run)
        assert pl.diff(ref.placements) is None and po.lists() == [[] for _ in range(j.num_jobs)]
    finally:
        eng.close()


def resv_preempt_case(seed):
    """Reservations (active / future / expired, jobs inside them, running jobs inside them) AND preemption."""
    from tests.test_reservations import random_resv_case
    c, j, now, run, rv = random_resv_case(seed, N=24, J=260, V=6)
    rng = np.random.default_rng(seed * 104729 + 7)
    R = len(run.end_sec)
    rn_qos = rng.integers(0, 3, R)
    pd_qos = rng.integers(0, 3, j.num_jobs)
    qprio = np.array([10, 20, 30])
    pre = abi.Preempt([[], [0], [1, 0]], np.arange(j.num_jobs) + 1, pd_qos, qprio[pd_qos], rng.permutation(j.num_jobs).astype(np.float64) + 0.5,
                      1000 + np.arange(R), rn_qos, qprio[rn_qos], now - 1 - rng.permutation(R) * 7,
                      preempting=[int(1000 + r) for r in range(R) if rng.random() < 0.1])
    return c, j, now, run, rv, pre


@pytest.mark.parametrize("seed", range(4))
def test_oracle_preempt_with_reservations_lit_vs_mask(built, seed):
    from oracle import pyoracle
    c, j, now, run, rv, pre = resv_preempt_case(seed)
    a = pyoracle.select(c, j, now, running=run, reservations=rv, preempt=pre, algebra=0)
    b = pyoracle.select(c, j, now, running=run, reservations=rv, preempt=pre, algebra=1)
    assert a.placements.diff(b.placements) is None and a.preempt_out.lists() == b.preempt_out.lists()
    assert a.preempt_out.cancelled_ids() == b.preempt_out.cancelled_ids()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_engine_preempt_with_reservations(gpu, seed):
    from oracle import pyoracle
    c, j, now, run, rv, pre = resv_preempt_case(seed)
    ref = pyoracle.select(c, j, now, running=run, reservations=rv, preempt=pre)
    eng, pl, po = run_engine_preempt(c, j, now, run, pre, resv=rv)
    try:
        d = pl.diff(ref.placements)
        assert d is None, f"engine placements differ from the oracle: {d}"
        assert po.lists() == ref.preempt_out.lists() and po.cancelled_ids() == ref.preempt_out.cancelled_ids()
        assert po.preempting_ids() == ref.preempt_out.preempting_ids()
        assert np.array_equal(eng.costs().view(np.uint64), ref.costs().view(np.uint64))
        for n in range(c.num_nodes):
            a, b = eng.timeline(n), ref.timeline(n)
            for f in ("t", "cpu_raw", "mem", "core_lo", "core_hi", "gres"):
                assert np.array_equal(a[f], b[f]), f"time map of node {n}, field {f}"
    finally:
        eng.close()


@pytest.mark.gpu
def test_engine_preempt_enabled_without_lists_runs_the_fast_kernels(gpu):
    """PreemptType == QOS but no pending job's qos may preempt anything: TryPreempt_ returns at :6385 every time, the cycle
    is the plain one except for the preempting set (jobs in it end at now + 1) — and it runs on k_wide / k_pipe."""
    from oracle import pyoracle
    c, j, now, run, pre = random_preempt_case(555, N=12, J=120, P=2, running=18)
    pre = abi.Preempt([[], [], []], pre.pd_job_id, pre.pd_qos, pre.pd_qos_priority, pre.pd_priority, pre.rn_job_id, pre.rn_qos,
                      pre.rn_qos_priority, pre.rn_start_sec, preempting=[1001, 1003, 4242])
    ref = pyoracle.select(c, j, now, running=run, preempt=pre)
    eng, pl, po = run_engine_preempt(c, j, now, run, pre)
    try:
        compare_engine("no lists", c, j, ref, eng, pl, po)
        assert not eng.last_kernel().startswith("k_select"), eng.last_kernel()
        # ... and the patched end times do not leak into the next plain cycle
        pl2 = eng.node_select(now, j)
        ref2 = pyoracle.select(c, j, now, running=run)
        assert pl2.diff(ref2.placements) is None
    finally:
        eng.close()


# ---------------------------------------------------------------------------------------------------------------------
# Preemption with partitions that share nodes: the candidate lists are the NODE's (one NodeState per craned,
# JobScheduler.cpp:6585-6615), a release acts on the nodes of the preempting job's own partition only (h:577-587) and
# lowers that partition's cost of the node alone (h:498-516)
# ---------------------------------------------------------------------------------------------------------------------
def overlap_preempt_case(seed, N=24, J=120, layout="all+subsets", nq=3):
    from tests.test_overlap import overlap_case
    c, j, now, run = overlap_case(seed, N=N, J=J, layout=layout)
    rng = np.random.default_rng(seed * 7919 + 17)
    R = len(run.end_sec)
    rn_qos = rng.integers(0, nq, R)
    pd_qos = rng.integers(0, nq, j.num_jobs)
    qprio = np.array([10, 20, 30])
    rn_start = now - 1 - rng.permutation(R) * 7
    pd_prio = rng.permutation(j.num_jobs).astype(np.float64) + 0.5
    preempting = [int(1000 + r) for r in range(R) if rng.random() < 0.15] + [4242]
    pre = abi.Preempt([[], [0], [1, 0]][:nq], np.arange(j.num_jobs) + 1, pd_qos, qprio[pd_qos], pd_prio,
                      1000 + np.arange(R), rn_qos, qprio[rn_qos], rn_start, preempting=preempting)
    return c, j, now, run, pre


OVERLAP_CASES = [(s, lay) for s in range(6) for lay in ("all+subsets", "chain", "random")]


@pytest.mark.parametrize("seed,lay", OVERLAP_CASES)
def test_python_restatement_on_preemption_with_shared_nodes(built, seed, lay):
    from oracle import pyoracle
    c, j, now, run, pre = overlap_preempt_case(seed, layout=lay)
    cyc, out, lists = run_pyref_preempt(c, j, now, run, pre)
    ref = pyoracle.select(c, j, now, running=run, preempt=pre)
    compare_preempt(f"overlap preempt {seed} {lay}", c, j, ref, cyc, out, lists)
    lit = pyoracle.select(c, j, now, running=run, preempt=pre, algebra=pyoracle.LITERAL)
    assert ref.placements.diff(lit.placements) is None and ref.preempt_out.lists() == lit.preempt_out.lists()
    if (seed, lay) == (0, "all+subsets"):
        tot = sum(len(x) for s, l in OVERLAP_CASES
                  for x in pyoracle.select(*overlap_preempt_case(s, layout=l)[:3], running=overlap_preempt_case(s, layout=l)[3],
                                           preempt=overlap_preempt_case(s, layout=l)[4]).preempt_out.lists())
        assert tot > 60, f"only {tot} preemptions over the shared-node cases"


@pytest.mark.gpu
@pytest.mark.parametrize("seed,lay", OVERLAP_CASES + [(s, lay) for s in (20, 21) for lay in ("all+subsets", "chain", "random")])
def test_engine_preempt_with_shared_nodes(gpu, seed, lay):
    from oracle import pyoracle
    big = seed >= 20
    c, j, now, run, pre = overlap_preempt_case(seed, N=64 if big else 24, J=500 if big else 120, layout=lay)
    ref = pyoracle.select(c, j, now, running=run, preempt=pre)
    eng, pl, po = run_engine_preempt(c, j, now, run, pre)
    try:
        compare_engine(f"overlap preempt {seed} {lay}", c, j, ref, eng, pl, po)
    finally:
        eng.close()


@pytest.mark.parametrize("seed", range(8))
def test_python_restatement_on_preemption_with_reservations(built, seed):
    """Reservations (their own schedulers, job lists and dips) AND preemption in the second restatement."""
    from oracle import pyoracle
    c, j, now, run, rv, pre = resv_preempt_case(seed)
    ref = pyoracle.select(c, j, now, running=run, reservations=rv, preempt=pre)
    cyc, out, lists = run_pyref_preempt(c, j, now, run, pre, rv=rv)
    compare_preempt(f"resv preempt {seed}", c, j, ref, cyc, out, lists)


@pytest.mark.gpu
def test_engine_split_preempt_cycle_survives_a_k_wide_retry(gpu, monkeypatch):
    """A cycle with preemption is split: partitions whose pending jobs may preempt run on k_select, the rest on k_wide.  A
    k_wide protocol fault re-runs the WHOLE cycle — k_select's partitions too — so the mutable preemption state (per-slot job
    lists, hidden candidates, the preempted-pair counter) must start every pass empty, not every call (a second
    pass over the first pass's lists walked a self-linked list).  Partition 0's jobs get a QoS that preempts nothing, the hook
    makes its leader scanner drop one exchange; the result must be the oracle's, preempted lists included."""
    from oracle import pyoracle
    c, j, now, run, pre = random_preempt_case(777, N=24, J=900, P=3, running=30)
    pd_qos = np.where(j.partition == 0, 0, pre.pd_qos).astype(np.uint32)
    qprio = np.array([10, 20, 30])
    pre = abi.Preempt([[], [0], [1, 0]], pre.pd_job_id, pd_qos, qprio[pd_qos], pre.pd_priority, pre.rn_job_id, pre.rn_qos,
                      pre.rn_qos_priority, pre.rn_start_sec, preempting=pre.preempting)
    assert int((j.partition == 0).sum()) > 200
    ref = pyoracle.select(c, j, now, running=run, preempt=pre)
    assert sum(len(x) for x in ref.preempt_out.lists()) > 0, "the case must preempt something"
    monkeypatch.setenv("CNS_SELECT_KERNEL", "wide")
    monkeypatch.setenv("CNS_WIDE_NO_RETRY", "0")
    eng, pl, po = run_engine_preempt(c, j, now, run, pre)
    try:
        assert eng.last_kernel().startswith("k_wide") and " + k_select" in eng.last_kernel(), eng.last_kernel()
        compare_engine("split cycle", c, j, ref, eng, pl, po)
        monkeypatch.setenv("CNS_WIDE_INJECT_STALL", "137")
        pl, po = eng.node_select_preempt(now, j, pre)
        assert "retry after" in eng.last_kernel() and not eng.last_kernel().startswith("k_wide"), eng.last_kernel()
        compare_engine("split cycle, retried", c, j, ref, eng, pl, po)
        monkeypatch.delenv("CNS_WIDE_INJECT_STALL")
        pl, po = eng.node_select_preempt(now, j, pre)
        compare_engine("split cycle, after", c, j, ref, eng, pl, po)
    finally:
        eng.close()


def stale_dip_case():
    """A dip that a release makes void (k_select's scanners keep one dip per node: select_kernels.hip, post_dip).  Node 0 has 6 cores:
    R0 (qos 0) holds {0,1} until 5000, R1 (qos 1) {2,3} until 2000; node 1 is full until 9000 (node 0 stays the cheaper one: TryPreempt_ works on the nodes phase B picks in cost order).
      J0 (qos 0, 4 cpus, 100 s)  cannot start now: backfilled on node 0 at 2000 — the entry at 2000 then has 0 cpus free.
      J1 (qos 0, 2 cpus, 5000 s) passes node 0's front (2 free) and trips over the entry at 2000: rejected, the scanners learn the dip.
      J2 (qos 1, 3 cpus, 50 s)   cannot start now, preempts R0: node 0 gets R0's 2 cpus back over [now, 5000) — incl. at 2000.
      J3 (qos 0, 1 cpu, 5000 s)  now fits node 0 at once (1 free now, 2 at 2000); a scanner that kept the dip would not propose node 0.
    (A library built with -DCNS_SEL_DIP_KEEP_ON_RELEASE keeps it: the engine test below fails on that build, profiles/r06_k_select_dips_c4rp.txt.)"""
    c = kat.cluster([6, 4], [64, 64])
    j = kat.jobs([dict(cpu=4, L=100), dict(cpu=2, L=5000), dict(cpu=3, L=50), dict(cpu=1, L=5000), dict(cpu=1, L=5000)])
    r = kat_preempt.running([dict(end=5000, allocs=[(0, 0x03, 1)]), dict(end=2000, allocs=[(0, 0x0C, 1)]), dict(end=9000, allocs=[(1, 0xF, 1)])])
    pre = kat_preempt.preempt([[], [0]], [(1, 0, 1, 5.0), (2, 0, 1, 4.0), (3, 1, 10, 3.0), (4, 0, 1, 2.0), (5, 0, 1, 1.0)],
                              [(50, 0, 1, 900), (51, 1, 10, 800), (52, 1, 10, 700)])
    return c, j, r, pre


def test_oracle_on_the_stale_dip_case(built):
    from oracle import pyoracle
    c, j, r, pre = stale_dip_case()
    ref = pyoracle.select(c, j, kat_preempt.NOW, running=r, preempt=pre)
    pl = ref.placements
    assert int(pl.start_sec[0]) == 2000 and int(pl.start_sec[1]) > kat_preempt.NOW, (pl.start_sec, pl.reason)   # J0 backfilled, J1 tripped over it
    assert ref.preempt_out.lists()[2] == [(False, 0)] and int(pl.start_sec[2]) == kat_preempt.NOW                 # J2 preempted R0
    assert int(pl.start_sec[3]) == kat_preempt.NOW and int(pl.node_idx[int(pl.place_offsets[3])]) == 0, (pl.start_sec, pl.node_idx)   # J3 at once on node 0


@pytest.mark.gpu
def test_engine_forgets_a_dip_when_preemption_releases_on_the_node(gpu):
    from oracle import pyoracle
    c, j, r, pre = stale_dip_case()
    ref = pyoracle.select(c, j, kat_preempt.NOW, running=r, preempt=pre)
    eng, pl, po = run_engine_preempt(c, j, kat_preempt.NOW, r, pre)
    try:
        compare_engine("stale dip", c, j, ref, eng, pl, po)
        assert eng.last_kernel().startswith("k_select")
    finally:
        eng.close()
