"""The C++ oracle against a second restatement of the selection path that shares no code with it (tests/select_pyref.py:
plain Python, literal containers).  Hand-derived scenarios + random small clusters with every feature of the slice
(unequal nodes, multi-type GRES, ntasks > node_num with the libstdc++ heap tie behaviour, exclusive jobs, include /
exclude lists, fractional CPUs, running jobs, partitions that share nodes)."""
import numpy as np
import pytest

from cranesched_amd import abi
from tests import helpers, kat, select_pyref as pr
from tests.test_overlap import overlap_case


def _bits(x):
    x = int(x)
    return {i for i in range(x.bit_length()) if (x >> i) & 1}


def _res(layout, cpu, mem, lo, hi, g):
    gres = {}
    for c in range(len(layout.class_name)):
        m = (int(g) >> layout.class_shift[c]) & ((1 << layout.class_width[c]) - 1)
        if m:
            gres[(layout.class_name[c], c)] = {layout.class_shift[c] + i for i in _bits(m)}
    return pr.Res(int(cpu), int(mem), _bits(lo) | {64 + i for i in _bits(hi)}, gres)


def _mask(layout, r: pr.Res):
    lo = sum(1 << c for c in r.cores if c < 64)
    hi = sum(1 << (c - 64) for c in r.cores if c >= 64)
    g = sum(1 << s for v in r.gres.values() for s in v)
    return lo, hi, g


def run_pyref(c: abi.Cluster, j: abi.Jobs, now, run=None, max_job_num_per_node=0, max_time_window_sec=0):
    lay = c.gres
    N = c.num_nodes
    chi = c.core_hi if c.core_hi is not None else np.zeros(N, np.uint64)
    gs = c.gres_slots if c.gres_slots is not None else np.zeros(N, np.uint64)
    # (core ids 128..255, ABI 3: the restatement keeps a set of ids, so they ride in `hi` as bits 64.. of one Python int)
    w2 = c.core_w2 if c.core_w2 is not None else np.zeros(N, np.uint64)
    w3 = c.core_w3 if c.core_w3 is not None else np.zeros(N, np.uint64)
    totals = [_res(lay, c.cpu_total_raw[n], c.mem_total[n], c.core_lo[n], int(chi[n]) | int(w2[n]) << 64 | int(w3[n]) << 128, gs[n]) for n in range(N)]
    parts = [list(map(int, c.part_nodes[c.part_offsets[p]:c.part_offsets[p + 1]])) for p in range(c.num_partitions)]
    types_of = lambda name: [g for g in range(len(lay.class_name)) if lay.class_name[g] == name]
    cyc = pr.Cycle(now, totals, parts, schedulable=None if c.schedulable is None else list(c.schedulable), types_of=types_of,
                   max_jobs_per_node=max_job_num_per_node or pr.MAX_JOBS_PER_NODE, max_window=max_time_window_sec or pr.MAX_WINDOW)
    if run is not None:
        ahi = run.alloc_core_hi if run.alloc_core_hi is not None else np.zeros(len(run.alloc_node), np.uint64)
        ag = run.alloc_gres if run.alloc_gres is not None else np.zeros(len(run.alloc_node), np.uint64)
        for r in range(len(run.end_sec)):
            al = [(int(run.alloc_node[a]), _res(lay, run.alloc_cpu_raw[a], run.alloc_mem[a], run.alloc_core_lo[a], ahi[a], ag[a]))
                  for a in range(int(run.alloc_offsets[r]), int(run.alloc_offsets[r + 1]))]
            cyc.add_running(int(run.end_sec[r]), al)
    cyc.start()
    out = []
    J = j.num_jobs
    for i in range(J):
        if j.skip is not None and j.skip[i]:
            out.append((abi.REASON_SKIPPED, 0, []))
            continue
        if j.partition[i] >= c.num_partitions:
            out.append((abi.REASON_PARTITION_NOT_FOUND, 0, []))
            continue
        gtot = {a: int(v) for a, v in enumerate(j.gres_total[i]) if v} if j.gres_total is not None else {}
        gspec = {(lay.class_name[g], g): int(v) for g, v in enumerate(j.gres_spec[i]) if v} if j.gres_spec is not None else {}
        node_view = pr.Req(int(j.node_cpu_raw[i]) if j.node_cpu_raw is not None else 0, int(j.node_mem[i]), gtot, gspec)
        incl = set(map(int, j.incl_nodes[int(j.incl_offsets[i]):int(j.incl_offsets[i + 1])])) if j.incl_offsets is not None else set()
        excl = set(map(int, j.excl_nodes[int(j.excl_offsets[i]):int(j.excl_offsets[i + 1])])) if j.excl_offsets is not None else set()
        job = dict(part=int(j.partition[i]), L=int(j.time_limit_sec[i]), k=int(j.node_num[i]), ntasks=int(j.ntasks[i]),
                   tmin=int(j.ntasks_per_node_min[i]), tmax=int(j.ntasks_per_node_max[i]), tcpu=int(j.task_cpu_raw[i]),
                   tmem=int(j.task_mem[i]), node_view=node_view, exclusive=bool(j.exclusive[i]) if j.exclusive is not None else False,
                   incl=incl, excl=excl)
        job["min_view"] = pr.compose(node_view, job["tcpu"], job["tmem"], job["tmin"])
        out.append(cyc.run_job(job))
    return cyc, out


def compare(tag, c, j, ref, cyc, out):
    lay = c.gres
    pl = ref.placements
    for i, (reason, start, picks) in enumerate(out):
        assert int(pl.reason[i]) == reason, f"{tag}: job {i} reason {pl.reason[i]} (oracle) vs {reason} (python)"
        assert int(pl.start_sec[i]) == start, f"{tag}: job {i} start {pl.start_sec[i]} vs {start}"
        o = int(pl.place_offsets[i])
        got = [(int(pl.node_idx[o + x]), int(pl.ntasks[o + x]), int(pl.cpu_raw[o + x]), int(pl.mem[o + x]), int(pl.core_lo[o + x]),
                int(pl.core_hi[o + x]) | int(pl.core_w2[o + x]) << 64 | int(pl.core_w3[o + x]) << 128, int(pl.gres[o + x])) for x in range(int(j.node_num[i])) if pl.node_idx[o + x] != abi.NODE_NONE]
        want = [(n, t, a.cpu, a.mem) + _mask(lay, a) for n, t, a in picks]
        assert got == want, f"{tag}: job {i} placements {got} (oracle) vs {want} (python)"
    costs = ref.costs().view(np.uint64)
    pos = 0
    for p in range(c.num_partitions):
        for n in c.part_nodes[c.part_offsets[p]:c.part_offsets[p + 1]]:
            if int(n) in cyc.cost[p]:
                assert np.float64(cyc.cost[p][int(n)]).view(np.uint64) == costs[pos], f"{tag}: cost of partition {p} node {n}"
            pos += 1


@pytest.mark.parametrize("scn", kat.scenarios(), ids=lambda s: s[0])
def test_python_restatement_on_hand_derived_scenarios(built, scn):
    from oracle import pyoracle
    name, c, j, cfg, expect = scn
    cyc, out = run_pyref(c, j, kat.NOW, **cfg)
    ref = pyoracle.select(c, j, kat.NOW, **cfg)
    compare(name, c, j, ref, cyc, out)


@pytest.mark.parametrize("seed", range(200))
def test_python_restatement_on_random_clusters(built, seed):
    from oracle import pyoracle
    c, j, now, run = helpers.random_case(300 + seed, N=20 + seed % 13, J=160, P=1 + seed % 3, running=8 + seed % 9)
    cyc, out = run_pyref(c, j, now, run)
    ref = pyoracle.select(c, j, now, running=run)
    compare(f"random {seed}", c, j, ref, cyc, out)


@pytest.mark.parametrize("seed,lay", [(11, "all+subsets"), (12, "chain"), (13, "random")])
def test_python_restatement_on_shared_nodes(built, seed, lay):
    from oracle import pyoracle
    c, j, now, run = overlap_case(seed, N=32, J=200, layout=lay)
    cyc, out = run_pyref(c, j, now, run)
    ref = pyoracle.select(c, j, now, running=run)
    compare(f"overlap {seed} {lay}", c, j, ref, cyc, out)


@pytest.mark.parametrize("seed", [100, 101])
def test_python_restatement_tight_limits(built, seed):
    from oracle import pyoracle
    c, j, now, run = helpers.random_case(seed, N=12, J=400, P=1, running=6)
    cfg = dict(max_job_num_per_node=12, max_time_window_sec=6 * 3600)
    cyc, out = run_pyref(c, j, now, run, **cfg)
    ref = pyoracle.select(c, j, now, running=run, **cfg)
    compare(f"tight {seed}", c, j, ref, cyc, out)
