"""Step scheduler = JobInCtld::SchedulePendingSteps (src/CraneCtld/CtldPublicDefs.cpp:2038-2159) for every job with
pending steps (include/crane_gpu/steps.h).

The reference has no test of this function: two hand-derived scenarios (expectations written from the cited lines
before running anything), literal-container algebra == mask algebra on random inputs, and under `-m gpu` the HIP
engine vs the oracle."""
import numpy as np
import pytest

from cranesched_amd import abi, steps as st
from oracle import pyoracle
from tests import helpers

GIB = 1 << 30


def _jobs(job_nodes, job_steps):
    """job_nodes: per job [(node idx, cpus, mem GiB, core_lo, gres)], job_steps: per job number of steps"""
    off, idx, cpu, mem, lo, g = [0], [], [], [], [], []
    for nodes in job_nodes:
        for n, c, m, l, gr in nodes:
            idx.append(n); cpu.append(int(c * 256)); mem.append(m * GIB); lo.append(l); g.append(gr)
        off.append(len(idx))
    W = (1 << 64) - 1   # `core_lo` may be a mask of up to 256 core ids: split into the four planes of the ABI
    wide = any(l >> 128 for l in lo)
    return st.StepJobs(off, idx, cpu, mem, [l & W for l in lo], [(l >> 64) & W for l in lo], g, np.cumsum([0] + list(job_steps)),
                       avail_core_w2=[(l >> 128) & W for l in lo] if wide else None,
                       avail_core_w3=[(l >> 192) & W for l in lo] if wide else None)


def _steps(specs):
    """specs: dicts k, ntasks, cpu (per task), mem_gib, tmin, tmax, ncpu (per node), incl, excl"""
    g = lambda key, d: [s.get(key, d) for s in specs]
    io, inn, eo, enn = [0], [], [0], []
    for s in specs:
        inn += s.get("incl", []); io.append(len(inn))
        enn += s.get("excl", []); eo.append(len(enn))
    return st.Steps(node_cpu_raw=[int(x * 256) for x in g("ncpu", 0)], node_mem=[x * GIB for x in g("nmem_gib", 0)],
                    task_cpu_raw=[int(x * 256) for x in g("cpu", 1)], task_mem=[x * GIB for x in g("mem_gib", 1)],
                    node_num=g("k", 1), ntasks=g("ntasks", 1), tmin=g("tmin", 1), tmax=g("tmax", 1),
                    incl_offsets=io, incl_nodes=inn or [0], excl_offsets=eo, excl_nodes=enn or [0])


def scenario_fifo():
    # job 0 owns n0, n1: 4 cores {0..3}, 8 GiB each.
    # A: 1 node, 2 tasks of 1 cpu / 1 GiB, <= 4 per node: n0 holds 4 tasks -> 1 node with >= 2 tasks: the walk stops
    #    (:2099-2101); rest = 2 - 1 = 1, n0 gets min(1, 4-1) + 1 = 2 tasks (:2116-2117): cores {0} and {1}
    #    (lowest free ids, PublicHeader.cpp:533-538); step_res_avail_[n0] = 2 cpus {2,3}, 6 GiB
    # B: 2 nodes, 4 tasks of 2 cpus, <= 2 per node: n0 now holds 1, n1 holds 2: 2 nodes, 3 tasks < 4 -> does not fit,
    #    the queue stops (:2104-2106)
    # C: would fit trivially, but is behind B: stays pending
    jobs = _jobs([[(0, 4, 8, 0xF, 0), (1, 4, 8, 0xF, 0)]], [3])
    steps = _steps([dict(k=1, ntasks=2, cpu=1, tmax=4), dict(k=2, ntasks=4, cpu=2, tmax=2), dict(k=1, ntasks=1, cpu=1)])
    exp = dict(scheduled=[1, 0, 0], node_idx=[0], node_ntasks=[2], node_core_lo=[0b0011], task_node=[0, 0],
               task_core_lo=[0b0001, 0b0010], avail_cpu=[2 * 256, 4 * 256], avail_core_lo=[0b1100, 0b1111])
    return jobs, steps, exp


def scenario_topk():
    # job 0 owns n0 (1 cpu), n1 (3 cpus), n2 (2 cpus); the step wants 2 nodes, 5 tasks of 1 cpu, <= 4 per node.
    # walk: n0 (1 task), n1 (3): 2 nodes, 4 tasks < 5 -> go on; n2 (2): three candidates -> the one with the fewest
    # tasks (n0) leaves the queue (:2093-2097), 2 nodes with 5 tasks -> stop.  Pop order = fewest tasks first
    # (:2059-2061): n2 then n1.  rest = 5 - 2 = 3: n2 gets min(3, 2-1) + 1 = 2 tasks (ids 0, 1), rest = 2;
    # n1 gets min(2, 3-1) + 1 = 3 tasks (ids 2, 3, 4).
    jobs = _jobs([[(0, 1, 8, 0b1, 0), (1, 3, 8, 0b111, 0), (2, 2, 8, 0b11, 0)]], [1])
    steps = _steps([dict(k=2, ntasks=5, cpu=1, tmax=4)])
    exp = dict(scheduled=[1], node_idx=[2, 1], node_ntasks=[2, 3], node_core_lo=[0b11, 0b111], task_node=[2, 2, 1, 1, 1],
               task_core_lo=[0b01, 0b10, 0b001, 0b010, 0b100], avail_cpu=[256, 0, 0], avail_core_lo=[0b1, 0, 0])
    return jobs, steps, exp


SCENARIOS = {"fifo": scenario_fifo, "topk": scenario_topk}


def _check(res, exp):
    S = len(exp["scheduled"])
    assert list(res.scheduled[:S]) == exp["scheduled"]
    n, t = len(exp["node_idx"]), len(exp["task_node"])
    assert list(res.node_idx[:n]) == exp["node_idx"] and list(res.node_ntasks[:n]) == exp["node_ntasks"]
    assert list(res.node_core_lo[:n]) == exp["node_core_lo"]
    assert list(res.task_node[:t]) == exp["task_node"] and list(res.task_core_lo[:t]) == exp["task_core_lo"]
    a = len(exp["avail_cpu"])
    assert list(res.avail_cpu_raw[:a]) == exp["avail_cpu"] and list(res.avail_core_lo[:a]) == exp["avail_core_lo"]


@pytest.mark.parametrize("name", sorted(SCENARIOS))
@pytest.mark.parametrize("algebra", [pyoracle.MASK, pyoracle.LITERAL])
def test_oracle_kat(name, algebra):
    jobs, steps, exp = SCENARIOS[name]()
    _check(pyoracle.schedule_steps(abi.GresLayout(), jobs, steps, algebra), exp)


def random_step_case(seed, J=300, wide=False):
    """wide: some nodes' free core ids lie above 127 (ABI 3: the avail_core_w2 / _w3 planes)"""
    rng = np.random.default_rng(seed + 4100)
    lay = helpers.multi_type_layout()
    job_nodes, nsteps = [], []
    for _ in range(J):
        k = int(rng.integers(1, 7))
        nodes = sorted(rng.choice(500, k, replace=False).tolist())
        rows = []
        for n in nodes:
            cores = int(rng.integers(0, 1 << 16)) | (int(rng.integers(0, 2)) << 16)     # random free core ids
            if wide and rng.random() < 0.5:   # the low ids are taken, what is free sits around the 128 and 192 boundaries
                cores = (int(rng.integers(0, 1 << 10)) << 122) | (int(rng.integers(0, 1 << 8)) << 188) | (cores & 0x3)
            frac = int(rng.integers(0, 3)) * 64                                           # some fractional cpu left over
            g = int(rng.choice([0, 0, 0x0F, 0xF3, 0xFF00, 0x3C5A]))
            rows.append((n, bin(cores).count("1") + frac / 256, int(rng.integers(1, 64)), cores, g))
        job_nodes.append(rows)
        nsteps.append(int(rng.integers(1, 5)))
    jobs = _jobs(job_nodes, nsteps)
    S = int(jobs.step_offsets[-1])
    specs = []
    for j in range(J):
        nodes = [r[0] for r in job_nodes[j]]
        for _ in range(nsteps[j]):
            k = int(rng.integers(1, min(len(nodes), 4) + 1))
            extra = int(rng.integers(0, 6))
            tmax = int(rng.integers(1, 5))
            d = dict(k=k, ntasks=k + extra, cpu=float(rng.choice([0.5, 1, 1, 2, 4])), mem_gib=int(rng.integers(0, 8)),
                     tmin=int(rng.integers(1, tmax + 1)), tmax=tmax, ncpu=float(rng.choice([0, 0, 0, 1])),
                     nmem_gib=int(rng.choice([0, 0, 1])))
            if rng.random() < 0.15:
                d["incl"] = rng.choice(nodes, int(rng.integers(1, len(nodes) + 1)), replace=False).tolist()
            if rng.random() < 0.15:
                d["excl"] = rng.choice(nodes, 1).tolist()
            specs.append(d)
    steps = _steps(specs)
    gt, gs = np.zeros((S, 4), np.uint8), np.zeros((S, 8), np.uint8)
    sel = rng.integers(0, 8, S)
    for s in range(S):
        if sel[s] == 0: gt[s, 0] = rng.integers(1, 3)                       # untyped gpu per node
        elif sel[s] == 1: c = rng.integers(1, 3); gt[s, 0] = c; gs[s, 0] = c   # typed a100
        elif sel[s] == 2: gt[s, 1] = rng.integers(1, 5)                     # npu
    steps.node_gres_total, steps.node_gres_spec = gt, gs
    return lay, jobs, steps


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_random_lit_vs_mask(seed):
    lay, jobs, steps = random_step_case(seed)
    a = pyoracle.schedule_steps(lay, jobs, steps, pyoracle.MASK)
    b = pyoracle.schedule_steps(lay, jobs, steps, pyoracle.LITERAL)
    assert a.diff(b) is None
    S = steps.num_steps
    sch = a.scheduled[:S]
    assert 0 < sch.sum() < S
    # FIFO: inside a job no scheduled step follows an unscheduled one
    for j in range(jobs.num_jobs):
        x = sch[jobs.step_offsets[j]:jobs.step_offsets[j + 1]]
        assert not np.any(np.diff(x.astype(np.int8)) > 0)
    # conservation: what the scheduled steps took is what left the availability
    took = np.zeros(jobs.num_nodes, np.int64)
    pos = {}
    for j in range(jobs.num_jobs):
        for p in range(jobs.node_offsets[j], jobs.node_offsets[j + 1]):
            pos[(j, int(jobs.node_idx[p]))] = p
    for j in range(jobs.num_jobs):
        for s in range(jobs.step_offsets[j], jobs.step_offsets[j + 1]):
            if sch[s]:
                for r in range(int(a.place_offsets[s]), int(a.place_offsets[s + 1])):
                    took[pos[(j, int(a.node_idx[r]))]] += a.node_cpu_raw[r]
    assert np.array_equal(jobs.avail_cpu_raw - took, a.avail_cpu_raw[:jobs.num_nodes])


def test_steps_abi_symbols(built):
    import os, re
    from cranesched_amd import engine
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "crane_gpu", "steps.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    assert set(re.findall(r"\b(cns_[a-z_]+)\s*\(", hdr)) == set(engine.STEPS_ABI_SYMBOLS)
    for s in engine.STEPS_ABI_SYMBOLS:
        getattr(engine.lib(), s)


# ---------------------------------------------------------------- GPU ----------------------------------------------------
def _gpu(engine_default, lay, jobs, steps):
    from tests import kat
    eng = engine_default(device=0)
    try:
        eng.set_nodes(kat.cluster([4], layout=lay))      # the handle's GRES layout comes with a node table
        got, ms = eng.schedule_steps(jobs, steps)
        ref = pyoracle.schedule_steps(lay, jobs, steps)
        assert got.diff(ref) is None, got.diff(ref)
        return got
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_gpu_kat(engine_default, name):
    jobs, steps, exp = SCENARIOS[name]()
    _check(_gpu(engine_default, abi.GresLayout(), jobs, steps), exp)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_gpu_random(engine_default, seed):
    lay, jobs, steps = random_step_case(seed, J=2000 if seed == 3 else 300)
    _gpu(engine_default, lay, jobs, steps)


@pytest.mark.parametrize("seed", [10, 11])
def test_oracle_random_core_ids_above_127(seed):
    lay, jobs, steps = random_step_case(seed, wide=True)
    a = pyoracle.schedule_steps(lay, jobs, steps, pyoracle.MASK)
    b = pyoracle.schedule_steps(lay, jobs, steps, pyoracle.LITERAL)
    assert a.diff(b) is None
    assert a.task_core_w2.any() and a.task_core_w3.any() and a.node_core_w2.any(), "case must hand out core ids above 127"


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [10, 11])
def test_gpu_random_core_ids_above_127(engine_default, seed):
    lay, jobs, steps = random_step_case(seed, J=600, wide=True)
    res = _gpu(engine_default, lay, jobs, steps)
    assert res.task_core_w2.any() and res.task_core_w3.any()


@pytest.mark.gpu
def test_gpu_argument_checks(engine_default):
    from cranesched_amd.engine import EngineError
    from tests import kat
    jobs, steps, _ = scenario_fifo()
    eng = engine_default(device=0)
    try:
        with pytest.raises(EngineError):     # the GRES layout arrives with the nodes
            eng.schedule_steps(jobs, steps)
        eng.set_nodes(kat.cluster([4]))
        bad = _steps([dict(k=2, ntasks=1), dict(k=1, ntasks=1), dict(k=1, ntasks=1)])   # ntasks < node_num
        with pytest.raises(EngineError):
            eng.schedule_steps(jobs, bad)
        got, _ = eng.schedule_steps(jobs, steps)
        assert list(got.scheduled[:3]) == [1, 0, 0]
    finally:
        eng.close()


# ---------------------------------------------------------------------------------------------------------------------
# The oracle against the independent Python restatement (tests/steps_pyref.py, written from the reference alone)
# ---------------------------------------------------------------------------------------------------------------------
def _pyref_steps(lay, jobs, steps):
    from tests import select_pyref as pr, steps_pyref
    from tests.test_select_pyref import _res, _mask
    types_of = lambda name: [g for g in range(len(lay.class_name)) if lay.class_name[g] == name]

    def view(cpu, mem, gt, gs, s):
        gtot = {} if gt is None else {a: int(v) for a, v in enumerate(gt[s]) if v}
        gspec = {} if gs is None else {(lay.class_name[g], g): int(v) for g, v in enumerate(gs[s]) if v}
        return pr.Req(int(cpu[s]), int(mem[s]), gtot, gspec)

    rows = []   # per step: None or (places, tasks) with masks
    avail_out = [None] * jobs.num_nodes
    for j in range(jobs.num_jobs):
        lo, hi = int(jobs.node_offsets[j]), int(jobs.node_offsets[j + 1])
        nodes = [int(x) for x in jobs.node_idx[lo:hi]]
        avail = [_res(lay, jobs.avail_cpu_raw[p], jobs.avail_mem[p], jobs.avail_core_lo[p], jobs.avail_core_hi[p], jobs.avail_gres[p])
                 for p in range(lo, hi)]
        sts = []
        for s in range(int(jobs.step_offsets[j]), int(jobs.step_offsets[j + 1])):
            incl = set() if steps.incl_offsets is None else set(int(x) for x in steps.incl_nodes[int(steps.incl_offsets[s]):int(steps.incl_offsets[s + 1])])
            excl = set() if steps.excl_offsets is None else set(int(x) for x in steps.excl_nodes[int(steps.excl_offsets[s]):int(steps.excl_offsets[s + 1])])
            sts.append(dict(node_view=view(steps.node_cpu_raw, steps.node_mem, steps.node_gres_total, steps.node_gres_spec, s),
                            task_view=view(steps.task_cpu_raw, steps.task_mem, steps.task_gres_total, steps.task_gres_spec, s),
                            k=int(steps.node_num[s]), ntasks=int(steps.ntasks[s]), tmin=int(steps.tmin[s]), tmax=int(steps.tmax[s]),
                            incl=incl, excl=excl))
        rows += steps_pyref.schedule_pending_steps(nodes, avail, sts, types_of)
        for p in range(lo, hi):
            avail_out[p] = avail[p - lo]
    return rows, avail_out, _mask


def _compare_steps(lay, jobs, steps, ref):
    rows, avail, _mask = _pyref_steps(lay, jobs, steps)
    for s, row in enumerate(rows):
        assert bool(ref.scheduled[s]) == (row is not None), f"step {s}: scheduled {ref.scheduled[s]} (oracle)"
        if row is None:
            continue
        places, tasks = row
        o, t = int(ref.place_offsets[s]), int(ref.task_offsets[s])
        got = [(int(ref.node_idx[o + i]), int(ref.node_ntasks[o + i]), int(ref.node_cpu_raw[o + i]), int(ref.node_mem[o + i]),
                int(ref.node_core_lo[o + i]), int(ref.node_core_hi[o + i]), int(ref.node_gres[o + i])) for i in range(len(places))]
        want = [(n, k, a.cpu, a.mem) + _mask(lay, a) for n, k, a in places]
        assert got == want, f"step {s}: nodes {got} (oracle) vs {want} (python)"
        got = [(int(ref.task_node[t + i]), int(ref.task_cpu_raw[t + i]), int(ref.task_mem[t + i]), int(ref.task_core_lo[t + i]),
                int(ref.task_core_hi[t + i]), int(ref.task_gres[t + i])) for i in range(len(tasks))]
        want = [(n, a.cpu, a.mem) + _mask(lay, a) for n, a in tasks]
        assert got == want, f"step {s}: tasks {got} (oracle) vs {want} (python)"
    for p, a in enumerate(avail):
        got = (int(ref.avail_cpu_raw[p]), int(ref.avail_mem[p]), int(ref.avail_core_lo[p]), int(ref.avail_core_hi[p]), int(ref.avail_gres[p]))
        assert got == (a.cpu, a.mem) + _mask(lay, a), f"node row {p}: step_res_avail_ {got} (oracle)"
    return sum(r is not None for r in rows), len(rows), sum(len(r[1]) for r in rows if r is not None)


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_python_restatement_on_the_hand_derived_scenarios(name):
    jobs, steps, exp = SCENARIOS[name]()
    lay = abi.GresLayout()
    _compare_steps(lay, jobs, steps, pyoracle.schedule_steps(lay, jobs, steps))


@pytest.mark.parametrize("seed", range(6))
def test_python_restatement_agrees_with_the_oracle(seed):
    lay, jobs, steps = random_step_case(10 + seed, J=120)
    done, total, tasks = _compare_steps(lay, jobs, steps, pyoracle.schedule_steps(lay, jobs, steps))
    assert 20 < done < total and tasks > 2 * done, (done, total, tasks)   # the cases schedule, refuse, and place several tasks per step
