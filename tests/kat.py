"""Hand-derived known-answer scenarios for the node-selection cycle.

The reference ships no test of NodeSelect, so each expectation below was derived by
hand from the cited reference lines and then frozen; the same vectors are checked against the CPU
oracle (tests/test_oracle_kat.py, both algebras) and against the HIP engine (tests/test_gpu_kat.py).
"""
from __future__ import annotations

import numpy as np

from cranesched_amd import abi

GIB = 1 << 30
NOW = 1000


def cluster(cores, mem_gib=None, gres=None, layout=None, parts=None):
    n = len(cores)
    cores = np.asarray(cores, np.int64)
    mem = np.asarray(mem_gib if mem_gib is not None else [16] * n, np.uint64) * np.uint64(GIB)
    W = 0xFFFFFFFFFFFFFFFF
    word = lambda c, w: np.array([(((1 << int(x)) - 1) >> (64 * w)) & W for x in c], np.uint64)   # core ids 0..c-1, word w
    lo, hi, w2, w3 = (word(cores, w) for w in range(4))
    g = np.asarray(gres if gres is not None else [0] * n, np.uint64)
    parts = parts or [list(range(n))]
    off = np.cumsum([0] + [len(p) for p in parts]).astype(np.uint32)
    pn = np.concatenate([np.asarray(p, np.uint32) for p in parts])
    wide = bool(w2.any() or w3.any())
    return abi.Cluster(cores * 256, mem, lo, hi, g, off, pn, gres=layout or abi.GresLayout(),
                       core_w2=w2 if wide else None, core_w3=w3 if wide else None)


def jobs(specs):
    """specs: dicts with cpu (cores, may be fractional), mem_gib (per task), L, and optionally
    k, ntasks, tmin, tmax, part, excl, gtot (list4), gspec (list8), node_mem_gib."""
    J = len(specs)
    g = lambda key, d: np.array([s.get(key, d) for s in specs])
    k = g("k", 1).astype(np.uint32)
    nt = np.array([s.get("ntasks", s.get("k", 1)) for s in specs], np.uint32)
    gt = np.zeros((J, 4), np.uint8); gs = np.zeros((J, 8), np.uint8)
    for i, s in enumerate(specs):
        for a, v in enumerate(s.get("gtot", [])): gt[i, a] = v
        for a, v in enumerate(s.get("gspec", [])): gs[i, a] = v
    return abi.Jobs(partition=g("part", 0).astype(np.uint32), time_limit_sec=g("L", 100).astype(np.int64),
                    node_mem=(g("node_mem_gib", 0).astype(np.uint64) * np.uint64(GIB)),
                    task_cpu_raw=np.round(g("cpu", 1).astype(np.float64) * 256).astype(np.int64),
                    task_mem=(g("mem_gib", 1).astype(np.uint64) * np.uint64(GIB)), node_num=k, ntasks=nt,
                    ntasks_per_node_min=g("tmin", 1).astype(np.uint32), ntasks_per_node_max=g("tmax", 1).astype(np.uint32),
                    exclusive=g("excl", 0).astype(np.uint8), gres_total=gt, gres_spec=gs)


def two_type_layout():
    # name 0 "gpu": class 0 a100 bits 0..3, class 1 h100 bits 4..7
    return abi.GresLayout(class_name=[0, 0], class_shift=[0, 4], class_width=[4, 4])


# Each scenario: (name, cluster, jobs, cfg, expect) where expect maps a job index to
#   (reason, start, [(node, ntasks, cpu_raw, core_lo, gres)])  -- mem is implied by the request
# plus optional "costs" (per part-slot fp64) and "timeline" {node: [(t, cpu_raw, core_lo)]}.
def scenarios():
    out = []

    # A. min-load-first with ties on the dense node index (JobScheduler.h:41-55,594; cpp:6188-6297).
    #    cost += secs * cpu_alloc/cpu_total: 100*0.25 = 25 per first job; 4th job returns to node 0 and
    #    takes the lowest free core id (PublicHeader.cpp:535-537).
    c = cluster([4, 4, 4])
    j = jobs([dict(L=100), dict(L=100), dict(L=100), dict(L=50)])
    INF = np.iinfo(np.int64).max
    out.append(("min_load_tie_order", c, j, {}, {
        0: (0, NOW, [(0, 1, 256, 0x1, 0)]), 1: (0, NOW, [(1, 1, 256, 0x1, 0)]),
        2: (0, NOW, [(2, 1, 256, 0x1, 0)]), 3: (0, NOW, [(0, 1, 256, 0x2, 0)]),
        "costs": [37.5, 25.0, 25.0],
        # UpdateResourceInNode case #3/#4 (JobScheduler.h:400-458): boundary at 1050 copies the
        # pre-subtraction value of the covering entry
        "timeline": {0: [(NOW, 512, 0xC), (1050, 768, 0xE), (1100, 1024, 0xF), (INF, 0, 0)]}}))

    # B. conservative backfill (JobScheduler.h:792-865; cpp:6335-6376, reasons :6797-6833) and the
    #    tolerant core erase (PublicHeader.cpp:758-762): job 3 is allocated core 0 against res_total,
    #    but at its start only core 1 is free -> cpu count drops to 0 while core 1 stays in the set.
    c = cluster([2], [8])
    j = jobs([dict(cpu=2, L=100), dict(cpu=1, L=50), dict(cpu=2, L=10), dict(cpu=1, L=10)])
    out.append(("backfill_and_tolerant_erase", c, j, {}, {
        0: (0, NOW, [(0, 1, 512, 0x3, 0)]), 1: (1, 1100, [(0, 1, 256, 0x1, 0)]),
        2: (1, 1150, [(0, 1, 512, 0x3, 0)]), 3: (1, 1100, [(0, 1, 256, 0x1, 0)]),
        "timeline": {0: [(NOW, 0, 0x0), (1100, 0, 0x2), (1110, 256, 0x2), (1150, 0, 0x0), (1160, 512, 0x3),
                         (INF, 0, 0)]}}))

    # C. kAlgoMaxTimeWindow (JobScheduler.h:270,815): earliest start 8 days out -> "Resource", no commit
    c = cluster([1], [8])
    j = jobs([dict(L=8 * 86400), dict(L=10)])
    out.append(("seven_day_horizon", c, j, {}, {0: (0, NOW, [(0, 1, 256, 0x1, 0)]), 1: (2, 0, [])}))

    # D. fractional request: no core ids are taken (PublicHeader.cpp:528-541); the next whole-cpu job
    #    does not fit the remaining 0.5 cpu and backfills behind it
    c = cluster([2], [8])
    j = jobs([dict(cpu=1.5, L=100), dict(cpu=1, L=10)])
    out.append(("fractional_cpu", c, j, {}, {0: (0, NOW, [(0, 1, 384, 0x0, 0)]), 1: (1, 1100, [(0, 1, 256, 0x1, 0)])}))

    # E. std::priority_queue tie behaviour (cpp:6157-6169,6233-6242,6288-6297): capacities
    #    1,1,1,3,3 in cost order, node_num 2, ntasks 4 -> the two oldest equal entries are evicted,
    #    nodes {2,3} stay; tasks are handed out smallest capacity first (:6304-6325).
    c = cluster([1, 1, 1, 3, 3])
    j = jobs([dict(cpu=1, L=100, k=2, ntasks=4, tmin=1, tmax=3)])
    out.append(("priority_queue_eviction", c, j, {}, {0: (0, NOW, [(2, 1, 256, 0x1, 0), (3, 3, 768, 0x7, 0)])}))

    # F. GRES slot choice (PublicHeader.cpp:549-594): typed count first, then the same type serves the
    #    untyped remainder (:577-578); a purely untyped request walks the types in ascending order.
    c = cluster([8, 8], gres=[0xFF, 0xFF], layout=two_type_layout())
    j = jobs([dict(L=100, gtot=[3], gspec=[0, 1]), dict(L=100, gtot=[2])])
    out.append(("gres_typed_then_untyped", c, j, {}, {0: (0, NOW, [(0, 1, 256, 0x1, 0x70)]),
                                                     1: (0, NOW, [(1, 1, 256, 0x1, 0x03)])}))

    # G. kAlgoMaxJobNumPerNode (cpp:6194): the map of node 0 reaches 4 entries after two jobs with
    #    different end times; with the limit set to 4 the third job finds no node -> "Resource"
    c = cluster([4], [8])
    j = jobs([dict(L=100), dict(L=200), dict(L=300)])
    out.append(("max_job_num_per_node", c, j, dict(max_job_num_per_node=4),
                {0: (0, NOW, [(0, 1, 256, 0x1, 0)]), 1: (0, NOW, [(0, 1, 256, 0x2, 0)]), 2: (2, 0, [])}))

    # H. exclusive job (cpp:6249-6271): needs a node that is completely free over its whole window;
    #    node 0 carries a 1-cpu job, node 1 is idle -> node 1, whole res_total allocated
    c = cluster([2, 2], [8, 8])
    j = jobs([dict(L=100), dict(L=50, excl=1)])
    out.append(("exclusive_whole_node", c, j, {}, {0: (0, NOW, [(0, 1, 256, 0x1, 0)]),
                                                  1: (0, NOW, [(1, 1, 512, 0x3, 0)])}))

    # I. UpdateResourceInNode when the boundaries ARE existing keys (JobScheduler.h:400-458, cases #3/#4 without an
    #    insertion): job 1 ends where job 0 ended (no new entry), job 2 (4 cpus) cannot start before 1100 — an existing
    #    key — and is allocated against res_total (cores {0..3}; 4 cpus <= the cycle-start res_avail -> "Priority"),
    #    job 3 still starts now: its window [1000, 1100) does not contain the entry at 1100 (:6279 `< now + L`) and it
    #    ends on that key.  cost = 100*1/4 + 100*1/4 + 50*4/4 + 100*2/4.
    c = cluster([4], [8])
    j = jobs([dict(L=100), dict(L=100), dict(cpu=4, L=50), dict(cpu=2, L=100)])
    out.append(("time_map_existing_keys", c, j, {}, {
        0: (0, NOW, [(0, 1, 256, 0x1, 0)]), 1: (0, NOW, [(0, 1, 256, 0x2, 0)]), 2: (1, 1100, [(0, 1, 1024, 0xF, 0)]),
        3: (0, NOW, [(0, 1, 512, 0xC, 0)]),
        "costs": [150.0],
        "timeline": {0: [(NOW, 0, 0x0), (1100, 0, 0x0), (1150, 1024, 0xF), (INF, 0, 0)]}}))

    # J. partitions that SHARE a node (JobScheduler.cpp:6563,6585-6617: one NodeState per craned; JobScheduler.h:498-516:
    #    one cost per partition selector).  Partition 0 = {0, 1}, partition 1 = {1, 2}.  Job 1 (partition 1) fills half of
    #    node 1; job 2 (partition 0, 4 cpus) still walks node 1 FIRST — partition 0's cost of node 1 is untouched by
    #    partition 1's job — finds only 2 cpus in its window there and on node 0, and is backfilled on node 1 behind job 1
    #    ("Priority": 4 cpus <= the cycle-start res_avail).  Job 4 (partition 1) sees that reservation in the shared time
    #    map and fits in front of it; job 5 ties 50.0 / 50.0 in partition 0 and takes the lower node index.
    c = cluster([4, 4, 4], parts=[[0, 1], [1, 2]])
    j = jobs([dict(part=0, cpu=2, L=100), dict(part=1, cpu=2, L=100), dict(part=0, cpu=4, L=50),
              dict(part=1, cpu=2, L=200), dict(part=1, cpu=2, L=40), dict(part=0, cpu=2, L=200)])
    out.append(("shared_node_two_partitions", c, j, {}, {
        0: (0, NOW, [(0, 1, 512, 0x3, 0)]), 1: (0, NOW, [(1, 1, 512, 0x3, 0)]), 2: (1, 1100, [(1, 1, 1024, 0xF, 0)]),
        3: (0, NOW, [(2, 1, 512, 0x3, 0)]), 4: (0, NOW, [(1, 1, 512, 0xC, 0)]), 5: (0, NOW, [(0, 1, 512, 0xC, 0)]),
        "costs": [150.0, 50.0, 70.0, 100.0],   # per (partition, node): [p0/n0, p0/n1, p1/n1, p1/n2]
        "timeline": {0: [(NOW, 0, 0x0), (1100, 512, 0x3), (1200, 1024, 0xF), (INF, 0, 0)],
                     1: [(NOW, 0, 0x0), (1040, 512, 0xC), (1100, 0, 0x0), (1150, 1024, 0xF), (INF, 0, 0)]}}))
    # K. a node with 192 cores (core ids above 127: the core_w2 / core_w3 planes; CpuSet::core_ids is an unbounded ordered set,
    #    PublicHeader.h:555-573, GetFeasibleResourceInNode takes the n LOWEST free ids, PublicHeader.cpp:536-541).
    #    Job 0 (100 cpus): both costs 0, tie -> node 0, ids 0..99.  Job 1 (60 cpus): node 1 is cheaper (0 < 100*100/192), ids 0..59
    #    there.  Job 2 (70 cpus, 50 s): node 0 (52.08 < 93.75) has 92 free -> ids 100..169, across the 128 boundary.  Job 3 (30 cpus):
    #    node 0 has 22 free, node 1 has 4 -> backfilled; node 0 is free enough at 1050 (job 2 ends), node 1 only at 1100 -> node 0
    #    at 1050, allocated against res_total (ids 0..29, :6353-6361), "Priority".  Job 4 (22 cpus, 40 s): fits node 0 now
    #    (ids 170..191) and ends before 1050.
    ids = lambda a, b: ((1 << b) - 1) ^ ((1 << a) - 1)    # core ids a..b-1 as one 256-bit mask
    c = cluster([192, 64], [1024, 1024])
    j = jobs([dict(cpu=100, L=100), dict(cpu=60, L=100), dict(cpu=70, L=50), dict(cpu=30, L=100), dict(cpu=22, L=40)])
    out.append(("node_with_192_cores", c, j, {}, {
        0: (0, NOW, [(0, 1, 100 * 256, ids(0, 100), 0)]), 1: (0, NOW, [(1, 1, 60 * 256, ids(0, 60), 0)]),
        2: (0, NOW, [(0, 1, 70 * 256, ids(100, 170), 0)]), 3: (1, 1050, [(0, 1, 30 * 256, ids(0, 30), 0)]),
        4: (0, NOW, [(0, 1, 22 * 256, ids(170, 192), 0)]),
        "costs": [100 * (100 / 192) + 50 * (70 / 192) + 100 * (30 / 192) + 40 * (22 / 192), 100 * (60 / 64)],
        # node 0: [1000, 1040) nothing free; at 1040 job 4 returns 170..191; at 1050 job 2 returns 100..169 and job 3 takes 30 cpus
        # (its ids 0..29 are still job 0's: the erase is tolerant, PublicHeader.cpp:758-762); at 1100 job 0 returns 0..99, of which
        # job 3 holds 0..29 until 1150
        "timeline": {0: [(NOW, 0, 0), (1040, 22 * 256, ids(170, 192)), (1050, 62 * 256, ids(100, 192)),
                         (1100, 162 * 256, ids(30, 192)), (1150, 192 * 256, ids(0, 192)), (INF, 0, 0)]}}))
    return out


def check(name, cluster_, jobs_, pl: abi.Placements, expect, costs=None, timeline=None):
    J = jobs_.num_jobs
    for ji in range(J):
        if ji not in expect:
            continue
        reason, start, recs = expect[ji]
        assert pl.reason[ji] == reason, f"{name}: job {ji} reason {pl.reason[ji]} != {reason}"
        assert pl.start_sec[ji] == start, f"{name}: job {ji} start {pl.start_sec[ji]} != {start}"
        o = int(pl.place_offsets[ji])
        k = int(jobs_.node_num[ji])
        core = lambda i: int(pl.core_lo[i]) | int(pl.core_hi[i]) << 64 | int(pl.core_w2[i]) << 128 | int(pl.core_w3[i]) << 192
        got = [(int(pl.node_idx[o + i]), int(pl.ntasks[o + i]), int(pl.cpu_raw[o + i]), core(o + i),
                int(pl.gres[o + i])) for i in range(k) if pl.node_idx[o + i] != abi.NODE_NONE]
        assert got == recs, f"{name}: job {ji} placements {got} != {recs}"
        if recs:
            t = int(pl.ntasks[o])
            want_mem = int(jobs_.node_mem[ji]) + t * int(jobs_.task_mem[ji])
            if not jobs_.exclusive[ji]:
                assert int(pl.mem[o]) == want_mem, f"{name}: job {ji} mem"
    if "costs" in expect and costs is not None:
        assert np.array_equal(np.asarray(expect["costs"], np.float64).view(np.uint64), costs.view(np.uint64)), \
            f"{name}: costs {costs} != {expect['costs']}"
    if "timeline" in expect and timeline is not None:
        for node, rows in expect["timeline"].items():
            tl = timeline(node)
            cores = [int(a) | int(b) << 64 | int(c2) << 128 | int(c3) << 192
                     for a, b, c2, c3 in zip(tl["core_lo"], tl["core_hi"], tl["core_w2"], tl["core_w3"])]
            got = list(zip(tl["t"].tolist(), tl["cpu_raw"].tolist(), cores))
            assert got == rows, f"{name}: timeline of node {node}: {got} != {rows}"
