"""Reservations inside NodeSelect (JobScheduler.cpp:6619-6679,6692-6707,6715-6719,6729-6732,6754-6760,
6797-6830; InitTimeAvailResMap JobScheduler.h:301-338; NodeRater JobScheduler.h:498-511).

Hand-derived scenarios (expectations written from the cited lines before running anything), checked against
the CPU oracle here and against the HIP engine under `-m gpu`; plus random cases engine-vs-oracle."""
import numpy as np
import pytest

from cranesched_amd import abi
from oracle import pyoracle
from tests import helpers, kat

NOW = kat.NOW
GIB = 1 << 30
R = abi  # reason codes


def _resv(specs):
    """specs: list of (start, end, [(node, cpus, mem_gib, core_lo)])"""
    off, node, cpu, mem, lo = [0], [], [], [], []
    for _, _, al in specs:
        for n, c, m, l in al:
            node.append(n); cpu.append(c * 256); mem.append(m * GIB); lo.append(l)
        off.append(len(node))
    z = [0] * len(node)
    return abi.Reservations([s[0] for s in specs], [s[1] for s in specs], off, node, cpu, mem, lo, z, z)


def _running(specs):
    """specs: list of (end, resv_or_None, [(node, cpus, mem_gib, core_lo)])"""
    off, node, cpu, mem, lo = [0], [], [], [], []
    for _, _, al in specs:
        for n, c, m, l in al:
            node.append(n); cpu.append(c * 256); mem.append(m * GIB); lo.append(l)
        off.append(len(node))
    z = [0] * len(node)
    return abi.Running([s[0] for s in specs], off, node, cpu, mem, lo, z, z,
                       reservation=[abi.RESV_NONE if s[1] is None else s[1] for s in specs])


def scenario_future_dip():
    # one node, 4 cores / 16 GiB; a FUTURE reservation takes the whole node during [NOW+100, NOW+200)
    #   time map: {NOW: 4c, NOW+100: 0c, NOW+200: 4c, INF: 0}  (reserved_res: - at start, + at end, h:305-308,326-334)
    #   initial cost = (200-100) * 4/4 = 100   (NodeRater ctor, reserved_res first, h:502-506)
    # job 0: 4c, L=50  -> window [NOW, NOW+50) is free -> starts now, cost 100 + 50 = 150
    # job 1: 4c, L=150 -> no 150 s gap before the dip; first fit after it: NOW+200; start != now and the node's
    #                     first reservation (NOW+100) < now + L (NOW+150) -> "Resource Reserved" (:6799-6806)
    # job 2: 2c, L=50  -> [NOW+50, NOW+100) has 4c free (job 0 ended, dip not yet) -> start NOW+50;
    #                     first_resv NOW+100 < NOW+50?  now + L = NOW+50: 1100 < 1050 is false -> not reserved;
    #                     alloc (2c) <= res_avail (4c at cycle start) -> "Priority"
    c = kat.cluster([4])
    j = kat.jobs([dict(cpu=4, L=50), dict(cpu=4, L=150), dict(cpu=2, L=50)])
    rv = _resv([(NOW + 100, NOW + 200, [(0, 4, 16, 0xF)])])
    exp = {0: (R.REASON_NONE, NOW, 0xF), 1: (R.REASON_RESOURCE_RESERVED, NOW + 200, 0xF),
           2: (R.REASON_PRIORITY, NOW + 50, 0x3)}
    return c, j, None, rv, exp


def scenario_active_resv():
    # node 0: 8 cores / 16 GiB.  Reservation 0 is ACTIVE [NOW-10, NOW+500): 4 cores {4..7}, 8 GiB of node 0.
    # Reservation 1 is in the future, reservation 2 expired.
    #   real node 0: avail now = 4 cores {0..3}, 8 GiB until NOW+500 (allocated_res gets {resv end, res}, :6644-6652)
    #   virtual node (resv 0, node 0): total 4 cores {4..7}, 8 GiB; a job running inside the reservation holds core 4
    #   until NOW+50; its time map ends at NOW+500 (InitTimeAvailResMap(now, end), h:337)
    # job 0 (no resv) 6c L=100: not now (4 free); earliest NOW+500; first_resv NOW-10 < now+100 -> "Resource Reserved"
    # job 1 (resv 0)  2c L=100: starts now on the virtual node, lowest free reserved cores {5,6}
    # job 2 (resv 0)  4c L=600: never fits before the reservation ends -> "Resource", no start
    # job 3 (resv 0)  2c L=100: now only 1 reserved core is free (7) until NOW+50 -> cores free at NOW+50: {4,7};
    #                          allocation against res_total takes the lowest two reserved cores {4,5} (:6353-6361);
    #                          {4,5} <= avail needs cpu/mem only... entry at NOW+50 has 2 cpus free but job 1 holds {5,6}
    #                          until NOW+100: operator<= compares cpu count, mem and GRES slots, not core ids
    #                          (PublicHeader.cpp:886-890) -> fits at NOW+50; reason: alloc cpu 2 <= res_avail cpu 3 -> "Priority"
    # job 4 (resv 1, future) -> "Reservation Not Found"; job 5 (resv 9, unknown) -> same; job 6 (resv 2, expired) -> same
    c = kat.cluster([8])
    specs = [dict(cpu=6, L=100), dict(cpu=2, L=100), dict(cpu=4, L=600), dict(cpu=2, L=100),
             dict(cpu=1, L=10), dict(cpu=1, L=10), dict(cpu=1, L=10)]
    j = kat.jobs(specs)
    j.reservation = np.array([abi.RESV_NONE, 0, 0, 0, 1, 9, 2], np.uint32)
    rv = _resv([(NOW - 10, NOW + 500, [(0, 4, 8, 0xF0)]), (NOW + 1000, NOW + 2000, [(0, 1, 1, 0x1)]),
                (NOW - 500, NOW - 100, [(0, 8, 16, 0xFF)])])
    rn = _running([(NOW + 50, 0, [(0, 1, 1, 0x10)])])
    exp = {0: (R.REASON_RESOURCE_RESERVED, NOW + 500, 0x3F), 1: (R.REASON_NONE, NOW, 0x60),
           2: (R.REASON_RESOURCE, 0, None), 3: (R.REASON_PRIORITY, NOW + 50, 0x30),
           4: (R.REASON_RESERVATION_NOT_FOUND, 0, None), 5: (R.REASON_RESERVATION_NOT_FOUND, 0, None),
           6: (R.REASON_RESERVATION_NOT_FOUND, 0, None)}
    return c, j, rn, rv, exp


SCENARIOS = {"future_dip": scenario_future_dip, "active_resv": scenario_active_resv}


def _check(got, exp):
    for ji, (reason, start, cores) in exp.items():
        assert int(got.reason[ji]) == reason, f"job {ji}: reason {got.reason[ji]} != {reason}"
        assert int(got.start_sec[ji]) == start, f"job {ji}: start {got.start_sec[ji]} != {start}"
        if cores is not None:
            o = int(got.place_offsets[ji])
            assert int(got.core_lo[o]) == cores, f"job {ji}: cores {int(got.core_lo[o]):#x} != {cores:#x}"


@pytest.mark.parametrize("name", sorted(SCENARIOS))
@pytest.mark.parametrize("algebra", [pyoracle.MASK, pyoracle.LITERAL])
def test_oracle_reservation_kat(name, algebra):
    c, j, rn, rv, exp = SCENARIOS[name]()
    ref = pyoracle.select(c, j, NOW, running=rn, reservations=rv, algebra=algebra)
    _check(ref.placements if hasattr(ref, "placements") else ref, exp)


def random_resv_case(seed, N=48, J=400, V=6):
    c, j, now, run = helpers.random_case(seed, N=N, J=J, P=2, running=20, general=True, lists=False, exclusive=True)
    rng = np.random.default_rng(4242 + seed)
    specs = []
    for v in range(V):
        kind = v % 3   # active / future / expired
        if kind == 0: s, e = now - int(rng.integers(1, 500)), now + int(rng.integers(2000, 9000))
        elif kind == 1: s = now + int(rng.integers(100, 4000)); e = s + int(rng.integers(500, 5000))
        else: s, e = now - 5000, now - int(rng.integers(1, 100))
        nodes = rng.choice(N, size=int(rng.integers(2, 8)), replace=False)
        al = []
        for n in nodes:
            cores = int(c.cpu_total_raw[n] // 256)
            take = int(rng.integers(1, max(2, cores // 4)))
            first = int(rng.integers(0, min(cores, 64) - take + 1))
            al.append((int(n), take, max(1, take), ((1 << take) - 1) << first))
        specs.append((s, e, al))
    rv = _resv(specs)
    resv = np.full(j.num_jobs, abi.RESV_NONE, np.uint32)
    pick = rng.random(j.num_jobs) < 0.25
    resv[pick] = rng.integers(0, V + 1, int(pick.sum()))   # V = unknown reservation
    j.reservation = resv
    # some running jobs live inside the active reservations (on their nodes, inside the reserved cores)
    end = list(run.end_sec); off = list(run.alloc_offsets); node = list(run.alloc_node); cpu = list(run.alloc_cpu_raw)
    mem = list(run.alloc_mem); lo = list(run.alloc_core_lo); hi = list(run.alloc_core_hi); g = list(run.alloc_gres)
    rres = [abi.RESV_NONE] * len(end)
    for v, (s_, e_, al) in enumerate(specs):
        if v % 3 != 0:
            continue
        for (n, take, m_gib, mask) in al[:2]:
            low = mask & -mask                      # lowest reserved core
            end.append(now + int(rng.integers(10, 3000))); rres.append(v)
            node.append(n); cpu.append(256); mem.append(GIB // 2); lo.append(low); hi.append(0); g.append(0)
            off.append(len(node))
    run = abi.Running(end, off, node, cpu, mem, lo, hi, g, reservation=rres)
    return c, j, now, run, rv


@pytest.mark.parametrize("seed", range(3))
def test_oracle_reservation_random_lit_vs_mask(seed):
    c, j, now, run, rv = random_resv_case(seed)
    a = pyoracle.select(c, j, now, running=run, reservations=rv, algebra=pyoracle.MASK)
    b = pyoracle.select(c, j, now, running=run, reservations=rv, algebra=pyoracle.LITERAL)
    pa = a.placements if hasattr(a, "placements") else a
    pb = b.placements if hasattr(b, "placements") else b
    assert (pa.reason[:j.num_jobs] == pb.reason[:j.num_jobs]).all()
    assert (pa.start_sec[:j.num_jobs] == pb.start_sec[:j.num_jobs]).all()
    r = pa.reason[:j.num_jobs]
    assert (r == R.REASON_RESOURCE_RESERVED).sum() > 0 and (r == R.REASON_RESERVATION_NOT_FOUND).sum() > 0


# ---- GPU parity ------------------------------------------------------------------------------------------

def _gpu_run(engine_cls, c, j, now, rn, rv, **cfg):
    eng = engine_cls(device=0, **cfg)
    try:
        eng.set_nodes(c)
        eng.set_reservations(rv)
        eng.set_running(rn)
        got = eng.node_select(now, j)
        ref = pyoracle.select(c, j, now, running=rn, reservations=rv, **cfg)
        helpers.assert_same(eng, got, ref, c, tag="resv")
        return got
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_gpu_reservation_kat(engine_cls, name):
    c, j, rn, rv, exp = SCENARIOS[name]()
    got = _gpu_run(engine_cls, c, j, NOW, rn, rv)
    _check(got, exp)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_gpu_reservation_random(engine_cls, seed):
    c, j, now, run, rv = random_resv_case(seed)
    got = _gpu_run(engine_cls, c, j, now, run, rv)
    r = got.reason[:j.num_jobs]
    assert (r == R.REASON_RESERVATION_NOT_FOUND).sum() > 0


@pytest.mark.gpu
def test_gpu_reservations_then_none(engine_cls):
    # the same handle: a cycle with reservations, then one without (cns_set_nodes resets the layout)
    c, j, now, run, rv = random_resv_case(11)
    eng = engine_cls(device=0)
    try:
        eng.set_nodes(c); eng.set_reservations(rv); eng.set_running(run)
        a = eng.node_select(now, j)
        j2 = abi.Jobs(**{f: getattr(j, f) for f in j.__dataclass_fields__ if f != "reservation"})
        eng.set_nodes(c); eng.set_running(run)
        b = eng.node_select(now, j2)
        ref = pyoracle.select(c, j2, now, running=run)
        helpers.assert_same(eng, b, ref, c, tag="no-resv after resv")
    finally:
        eng.close()


@pytest.mark.gpu
def test_gpu_reservation_argument_checks(engine_cls):
    from cranesched_amd.engine import EngineError
    c = kat.cluster([4, 4])
    eng = engine_cls(device=0)
    try:
        with pytest.raises(EngineError) as ei:                       # call order
            eng.set_reservations(_resv([(NOW, NOW + 10, [(0, 1, 1, 0x1)])]))
        assert ei.value.status == -5
        eng.set_nodes(c)
        with pytest.raises(EngineError) as ei:                       # node listed twice in one reservation
            eng.set_reservations(_resv([(NOW, NOW + 10, [(0, 1, 1, 0x1), (0, 1, 1, 0x2)])]))
        assert ei.value.status == -1
        with pytest.raises(EngineError) as ei:                       # node outside the cluster
            eng.set_reservations(_resv([(NOW, NOW + 10, [(7, 1, 1, 0x1)])]))
        assert ei.value.status == -1
        # a valid one, then none: back to the plain layout
        eng.set_reservations(_resv([(NOW - 5, NOW + 50, [(1, 2, 2, 0x3)])]))
        eng.set_reservations(None)
        j = kat.jobs([dict(cpu=4, L=10), dict(cpu=4, L=10)])
        got = eng.node_select(NOW, j)
        assert got.reason[:2].tolist() == [0, 0] and sorted(got.node_idx[:2].tolist()) == [0, 1]
    finally:
        eng.close()


def core_id_shortfall_case():
    """res_avail with FEWER core ids than cpus (cpu_total 4, core ids {0,1}: the node table does not force them to
    agree) and two future dips whose core sets are disjoint, so the window minimum of a long job has an EMPTY core set:
    GetFeasibleResourceInNode succeeds on the window minimum through its count-only branch (PublicHeader.cpp:540) while
    the test on res_avail (JobScheduler.cpp:6274) fails on the core-id count (:534).  The engine evaluates :6285 first
    and derives :6274 from it, except for exactly this test."""
    c = abi.Cluster(np.array([4 * 256], np.int64), np.array([16 * GIB], np.uint64), np.array([0b111], np.uint64),
                    np.array([0], np.uint64), np.array([0], np.uint64), np.array([0, 1], np.uint32), np.array([0], np.uint32))
    j = kat.jobs([dict(cpu=3, L=100), dict(cpu=2, L=100), dict(cpu=3, L=5), dict(cpu=0.5, L=100)])
    # a running job holds core id 2 but only half a cpu; the two reservations take one core id each and a quarter cpu
    rn = _running([(NOW + 5000, None, [(0, 0.5, 1, 0b100)])])
    rv = _resv([(NOW + 10, NOW + 20, [(0, 0.25, 1, 0b01)]), (NOW + 30, NOW + 40, [(0, 0.25, 1, 0b10)])])
    return c, j, rn, rv


def test_oracle_core_id_shortfall():
    c, j, rn, rv = core_id_shortfall_case()
    ref = pyoracle.select(c, j, NOW, running=rn, reservations=rv)
    lit = pyoracle.select(c, j, NOW, running=rn, reservations=rv, algebra=pyoracle.LITERAL)
    assert ref.placements.diff(lit.placements) is None
    # job 0 (3 cpus, window over both dips): window minimum = 3.25 cpus, no core ids -> count-only success; res_avail has
    # 2 core ids < 3 -> :6274 rejects the node for "start now" -> the job is placed by the backfill branch instead: its
    # allocation is the one against res_total (core ids {0,1,2}; `<=` ignores core ids, PublicHeader.cpp:886-890) and it
    # still starts at `now`.  Had :6285 alone decided, the allocation would carry no core ids at all.
    assert ref.placements.reason[0] == 0 and ref.placements.start_sec[0] == NOW and ref.placements.core_lo[0] == 0b111


@pytest.mark.gpu
def test_gpu_core_id_shortfall(engine_cls):
    c, j, rn, rv = core_id_shortfall_case()
    _gpu_run(engine_cls, c, j, NOW, rn, rv)


# ---------------------------------------------------------------------------------------------------------------------
# The oracle against the independent Python restatement (tests/select_pyref.py: ResvCycle, written from the reference alone)
# ---------------------------------------------------------------------------------------------------------------------
def run_pyref_resv(c, j, now, run, rv):
    from tests import select_pyref as pr
    from tests.test_select_pyref import _res
    lay = c.gres
    N = c.num_nodes
    chi = c.core_hi if c.core_hi is not None else np.zeros(N, np.uint64)
    gs = c.gres_slots if c.gres_slots is not None else np.zeros(N, np.uint64)
    totals = [_res(lay, c.cpu_total_raw[n], c.mem_total[n], c.core_lo[n], chi[n], gs[n]) for n in range(N)]
    parts = [list(map(int, c.part_nodes[c.part_offsets[p]:c.part_offsets[p + 1]])) for p in range(c.num_partitions)]
    types_of = lambda name: [g for g in range(len(lay.class_name)) if lay.class_name[g] == name]
    V = len(rv.start_sec)
    resvs = [dict(start=int(rv.start_sec[v]), end=int(rv.end_sec[v]),
                  allocs=[(int(rv.alloc_node[a]), _res(lay, rv.alloc_cpu_raw[a], rv.alloc_mem[a], rv.alloc_core_lo[a], rv.alloc_core_hi[a], rv.alloc_gres[a]))
                          for a in range(int(rv.alloc_offsets[v]), int(rv.alloc_offsets[v + 1]))]) for v in range(V)]
    jresv = j.reservation if getattr(j, "reservation", None) is not None else np.full(j.num_jobs, abi.RESV_NONE, np.uint32)
    pend = {int(x) for x in jresv if x != abi.RESV_NONE}             # resv_pd_job_ptr_map (:6524-6530): every pending job counts
    cyc = pr.ResvCycle(now, totals, parts, schedulable=None if c.schedulable is None else list(c.schedulable), types_of=types_of,
                       reservations=resvs, pending_resv=pend)
    if run is not None:
        ahi = run.alloc_core_hi if run.alloc_core_hi is not None else np.zeros(len(run.alloc_node), np.uint64)
        ag = run.alloc_gres if run.alloc_gres is not None else np.zeros(len(run.alloc_node), np.uint64)
        rres = run.reservation if getattr(run, "reservation", None) is not None else [abi.RESV_NONE] * len(run.end_sec)
        for r in range(len(run.end_sec)):
            al = [(int(run.alloc_node[a]), _res(lay, run.alloc_cpu_raw[a], run.alloc_mem[a], run.alloc_core_lo[a], ahi[a], ag[a]))
                  for a in range(int(run.alloc_offsets[r]), int(run.alloc_offsets[r + 1]))]
            cyc.add_running(int(run.end_sec[r]), al, resv=None if rres[r] == abi.RESV_NONE else int(rres[r]))
    cyc.start()
    out = []
    for i in range(j.num_jobs):
        if j.skip is not None and j.skip[i]:
            out.append((abi.REASON_SKIPPED, 0, []))
            continue
        v = None if jresv[i] == abi.RESV_NONE else int(jresv[i])
        if v is None and j.partition[i] >= c.num_partitions:
            out.append((abi.REASON_PARTITION_NOT_FOUND, 0, []))
            continue
        gtot = {a: int(x) for a, x in enumerate(j.gres_total[i]) if x} if j.gres_total is not None else {}
        gspec = {(lay.class_name[g], g): int(x) for g, x in enumerate(j.gres_spec[i]) if x} if j.gres_spec is not None else {}
        node_view = pr.Req(int(j.node_cpu_raw[i]) if j.node_cpu_raw is not None else 0, int(j.node_mem[i]), gtot, gspec)
        incl = set(map(int, j.incl_nodes[int(j.incl_offsets[i]):int(j.incl_offsets[i + 1])])) if j.incl_offsets is not None else set()
        excl = set(map(int, j.excl_nodes[int(j.excl_offsets[i]):int(j.excl_offsets[i + 1])])) if j.excl_offsets is not None else set()
        job = dict(part=int(j.partition[i]), L=int(j.time_limit_sec[i]), k=int(j.node_num[i]), ntasks=int(j.ntasks[i]),
                   tmin=int(j.ntasks_per_node_min[i]), tmax=int(j.ntasks_per_node_max[i]), tcpu=int(j.task_cpu_raw[i]),
                   tmem=int(j.task_mem[i]), node_view=node_view, exclusive=bool(j.exclusive[i]) if j.exclusive is not None else False,
                   incl=incl, excl=excl, resv=v)
        job["min_view"] = pr.compose(node_view, job["tcpu"], job["tmem"], job["tmin"])
        out.append(cyc.run_job(job))
    return cyc, out


def _compare_pyref_resv(tag, c, j, ref, cyc, out):
    from tests.test_select_pyref import _mask
    lay = c.gres
    pl = ref.placements if hasattr(ref, "placements") else ref
    for i, (reason, start, picks) in enumerate(out):
        assert int(pl.reason[i]) == reason, f"{tag}: job {i} reason {pl.reason[i]} (oracle) vs {reason} (python)"
        assert int(pl.start_sec[i]) == start, f"{tag}: job {i} start {pl.start_sec[i]} vs {start}"
        o = int(pl.place_offsets[i])
        got = [(int(pl.node_idx[o + x]), int(pl.ntasks[o + x]), int(pl.cpu_raw[o + x]), int(pl.mem[o + x]), int(pl.core_lo[o + x]),
                int(pl.core_hi[o + x]), int(pl.gres[o + x])) for x in range(int(j.node_num[i])) if pl.node_idx[o + x] != abi.NODE_NONE]
        want = [(n, t, a.cpu, a.mem) + _mask(lay, a) for n, t, a in picks]
        assert got == want, f"{tag}: job {i} placements {got} (oracle) vs {want} (python)"
    if hasattr(ref, "timeline"):          # the real nodes' final time maps
        for n, nd in cyc.nodes.items():
            m = ref.timeline(n)
            got = [(int(t), int(cpu), int(lo)) for t, cpu, lo in zip(m["t"], m["cpu_raw"], m["core_lo"])]
            want = [(t if t != pr_inf() else int(np.iinfo(np.int64).max), r.cpu, _mask(lay, r)[0]) for t, r in nd.tmap]
            assert got == want, f"{tag}: time map of node {n}: {got} (oracle) vs {want} (python)"


def pr_inf():
    from tests import select_pyref as pr
    return pr.INF


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_python_restatement_on_the_hand_derived_reservation_scenarios(name):
    c, j, rn, rv, exp = SCENARIOS[name]()
    ref = pyoracle.select(c, j, NOW, running=rn, reservations=rv)
    cyc, out = run_pyref_resv(c, j, NOW, rn, rv)
    _compare_pyref_resv(name, c, j, ref, cyc, out)


@pytest.mark.parametrize("seed", range(8))
def test_python_restatement_agrees_with_the_oracle_on_reservations(seed):
    c, j, now, run, rv = random_resv_case(seed, N=24 + 4 * seed, J=150)
    ref = pyoracle.select(c, j, now, running=run, reservations=rv)
    cyc, out = run_pyref_resv(c, j, now, run, rv)
    _compare_pyref_resv(f"resv {seed}", c, j, ref, cyc, out)
    reasons = [o[0] for o in out]
    if seed == 0:
        assert reasons.count(abi.REASON_RESV_NOT_FOUND if hasattr(abi, "REASON_RESV_NOT_FOUND") else 6) > 0 and reasons.count(3) + reasons.count(1) > 0
