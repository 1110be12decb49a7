"""Shared helpers of the parity tests: random scenario generator and oracle-vs-engine comparison."""
from __future__ import annotations

import numpy as np

from cranesched_amd import abi, synth

GIB = 1 << 30


def multi_type_layout() -> abi.GresLayout:
    # name 0 "gpu": class 0 (a100, 4 slots, bits 0..3), class 1 (h100, 4 slots, bits 4..7)
    # name 1 "npu": class 2 (a910, 8 slots, bits 8..15)
    return abi.GresLayout(class_name=[0, 0, 1], class_shift=[0, 4, 8], class_width=[4, 4, 8])


def random_case(seed: int, N: int = 96, J: int = 600, P: int = 2, running: int = 40,
                general: bool = True, lists: bool = True, exclusive: bool = True, frac: bool = True):
    """A small heterogeneous scenario touching every feature of the slice: unequal node sizes,
    multi-type GRES, ntasks > node_num, exclusive jobs, include/exclude lists, fractional CPUs,
    running jobs (avail0 != total, initial cost != 0)."""
    rng = np.random.default_rng(seed)
    layout = multi_type_layout()
    kind = rng.integers(0, 5, N)
    cores = np.array([16, 32, 64, 96, 128])[kind]
    cpu_total_raw = (cores * 256).astype(np.int64)
    mem_total = (cores.astype(np.uint64) * np.uint64(4 * GIB))
    core_lo = np.where(cores >= 64, np.uint64(0xFFFFFFFFFFFFFFFF), (np.uint64(1) << cores.astype(np.uint64)) - np.uint64(1)).astype(np.uint64)
    core_hi = np.where(cores > 64, (np.uint64(1) << np.minimum(cores - 64, 63).astype(np.uint64)) - np.uint64(1), np.uint64(0)).astype(np.uint64)
    core_hi = np.where(cores == 128, np.uint64(0xFFFFFFFFFFFFFFFF), core_hi).astype(np.uint64)
    gk = rng.integers(0, 6, N)
    gres_slots = np.array([0, 0, 0x0F, 0xFF, 0xFF00, 0xFFF3], np.uint64)[gk]
    sched = (rng.random(N) > 0.05).astype(np.uint8)
    # disjoint partitions, round-robin
    part_lists = [np.nonzero(np.arange(N) % P == p)[0] for p in range(P)]
    part_offsets = np.cumsum([0] + [len(x) for x in part_lists]).astype(np.uint32)
    part_nodes = np.concatenate(part_lists).astype(np.uint32)
    cluster = abi.Cluster(cpu_total_raw, mem_total, core_lo, core_hi, gres_slots, part_offsets, part_nodes,
                          gres=layout, schedulable=sched)
    now = synth.NOW
    # running jobs: one node each, a slice of the node
    run = None
    if running:
        nodes = rng.integers(0, N, running)
        ncpu = rng.integers(1, 5, running)
        end = now + rng.integers(-50, 5000, running)
        # non-overlapping core ranges per node: k-th allocation on a node takes cores [4k, 4k+ncpu)
        seen = {}
        lo = np.zeros(running, np.uint64); hi = np.zeros(running, np.uint64)
        cpu = np.zeros(running, np.int64)
        keep = []
        for i, n in enumerate(nodes):
            k = seen.get(int(n), 0)
            if 4 * k + 4 > cores[n]:
                continue
            seen[int(n)] = k + 1
            m = ((1 << int(ncpu[i])) - 1) << (4 * k)
            lo[i] = np.uint64(m & 0xFFFFFFFFFFFFFFFF); hi[i] = np.uint64(m >> 64)
            cpu[i] = int(ncpu[i]) * 256
            keep.append(i)
        keep = np.array(keep, np.int64)
        r = len(keep)
        run = abi.Running(end_sec=end[keep], alloc_offsets=np.arange(r + 1), alloc_node=nodes[keep],
                          alloc_cpu_raw=cpu[keep], alloc_mem=(ncpu[keep].astype(np.uint64) * np.uint64(GIB)),
                          alloc_core_lo=lo[keep], alloc_core_hi=hi[keep], alloc_gres=np.zeros(r, np.uint64))
    # jobs
    k = np.where(rng.random(J) < 0.7, 1, rng.integers(1, 5, J)).astype(np.uint32)
    if general:
        extra = np.where(rng.random(J) < 0.3, rng.integers(0, 6, J), 0).astype(np.uint32)
    else:
        extra = np.zeros(J, np.uint32)
    ntasks = k + extra
    dist_max = ntasks - k + 1
    tmax = dist_max.copy()
    capmax = np.where(rng.random(J) < 0.3, rng.integers(1, 4, J), 0).astype(np.uint32)   # user ntasks-per-node cap
    tmax = np.where(capmax > 0, np.minimum(tmax, capmax), tmax).astype(np.uint32)
    tmin = np.maximum(1, ntasks.astype(np.int64) - (k.astype(np.int64) - 1) * tmax.astype(np.int64)).astype(np.uint32)
    tmax = np.minimum(tmax.astype(np.int64), ntasks.astype(np.int64) - (k.astype(np.int64) - 1) * tmin.astype(np.int64))
    bad = tmin.astype(np.int64) > tmax
    ntasks = np.where(bad, k, ntasks).astype(np.uint32)
    tmin = np.where(bad, 1, tmin).astype(np.uint32)
    tmax = np.where(bad, 1, tmax).astype(np.uint32)
    cpus = rng.choice([1, 2, 4, 8, 16], J)
    task_cpu_raw = (cpus * 256).astype(np.int64)
    if frac:
        fr = rng.random(J) < 0.15
        task_cpu_raw = np.where(fr, task_cpu_raw // 2 + 128, task_cpu_raw).astype(np.int64)   # x.5 cpus
    task_mem = (cpus.astype(np.uint64) * np.uint64(2 * GIB))
    node_mem = np.where(rng.random(J) < 0.2, np.uint64(GIB), np.uint64(0)).astype(np.uint64)
    L = (60 * rng.integers(1, 200, J)).astype(np.int64)
    partition = rng.integers(0, P, J).astype(np.uint32)
    partition = np.where(rng.random(J) < 0.01, P + 3, partition).astype(np.uint32)   # "Partition Not Found"
    gres_total = np.zeros((J, abi.MAX_GRES_NAMES), np.uint8)
    gres_spec = np.zeros((J, abi.MAX_GRES_CLASSES), np.uint8)
    gsel = rng.integers(0, 12, J)
    for j in range(J):
        s = gsel[j]
        if s == 0: gres_total[j, 0] = rng.integers(1, 5)                       # untyped gpu
        elif s == 1: c = rng.integers(1, 4); gres_total[j, 0] = c; gres_spec[j, 0] = c      # typed a100
        elif s == 2: c = rng.integers(1, 4); gres_total[j, 0] = c + 1; gres_spec[j, 1] = c  # h100 + 1 untyped
        elif s == 3: gres_total[j, 1] = rng.integers(1, 9)                     # npu
        elif s == 4: gres_spec[j, 0] = 1; gres_spec[j, 1] = 1; gres_total[j, 0] = 3         # both types + 1
    excl = ((rng.random(J) < 0.05) & exclusive).astype(np.uint8)
    skip = (rng.random(J) < 0.01).astype(np.uint8)
    incl_off = [0]; incl = []; excl_off = [0]; exn = []
    for j in range(J):
        if lists and rng.random() < 0.05:
            incl += list(rng.choice(N, size=rng.integers(1, min(12, N)), replace=False))
        if lists and rng.random() < 0.05:
            exn += list(rng.choice(N, size=rng.integers(1, min(30, N)), replace=False))
        incl_off.append(len(incl)); excl_off.append(len(exn))
    jobs = abi.Jobs(partition=partition, time_limit_sec=L, node_mem=node_mem, task_cpu_raw=task_cpu_raw,
                    task_mem=task_mem, node_num=k, ntasks=ntasks, ntasks_per_node_min=tmin,
                    ntasks_per_node_max=tmax.astype(np.uint32), exclusive=excl, gres_total=gres_total,
                    gres_spec=gres_spec, incl_offsets=np.array(incl_off, np.uint64),
                    incl_nodes=np.array(incl if incl else [0], np.uint32),
                    excl_offsets=np.array(excl_off, np.uint64), excl_nodes=np.array(exn if exn else [0], np.uint32),
                    skip=skip)
    return cluster, jobs, now, run


def widen_cores(cluster: abi.Cluster, seed: int = 0) -> abi.Cluster:
    """The same cluster with its 128-core nodes (and every fourth smaller one) grown to 192 or 256 cores — core ids 128..255
    in the core_w2 / core_w3 planes of ABI 3 (CpuSet::core_ids has no bound, PublicHeader.h:555-573).  Running allocations of a
    case stay valid: they hold low core ids."""
    rng = np.random.default_rng(9000 + seed)
    n = cluster.num_nodes
    cores = (cluster.cpu_total_raw // 256).astype(np.int64)
    grow = (cores == 128) | (rng.random(n) < 0.25)
    target = np.where(rng.random(n) < 0.5, 192, 256)
    new = np.where(grow, target, cores)
    full = np.uint64(0xFFFFFFFFFFFFFFFF)
    lo = np.where(grow, full, cluster.core_lo).astype(np.uint64)
    hi = np.where(grow, full, cluster.core_hi).astype(np.uint64)
    w2 = np.where(grow, full, np.uint64(0)).astype(np.uint64)
    w3 = np.where(grow & (new == 256), full, np.uint64(0)).astype(np.uint64)
    return abi.Cluster((new * 256).astype(np.int64), (new.astype(np.uint64) * np.uint64(4 * GIB)), lo, hi, cluster.gres_slots,
                       cluster.part_offsets, cluster.part_nodes, gres=cluster.gres, schedulable=cluster.schedulable,
                       core_w2=w2, core_w3=w3)


def assert_same(eng, got: abi.Placements, ref, cluster: abi.Cluster, sample_nodes: int = 24, tag: str = ""):
    """Placements, fp64 cost bit patterns and a sample of final time maps must be identical."""
    d = got.diff(ref.placements)
    assert d is None, f"{tag}: placements differ from the oracle at {d}"
    gc, rc = eng.costs().view(np.uint64), ref.costs().view(np.uint64)
    ne = np.nonzero(gc != rc)[0]
    assert len(ne) == 0, f"{tag}: {len(ne)} fp64 costs differ, first at part-slot {ne[0]}: " \
                         f"{eng.costs()[ne[0]]!r} vs {ref.costs()[ne[0]]!r}"
    nodes = np.unique(np.linspace(0, cluster.num_nodes - 1, sample_nodes).astype(np.int64))
    # plus the busiest nodes
    used = got.node_idx[:got.capacity]
    used = used[used != abi.NODE_NONE]
    if len(used):
        busiest = np.argsort(np.bincount(used, minlength=cluster.num_nodes))[-8:]
        nodes = np.unique(np.concatenate([nodes, busiest]))
    in_part = np.zeros(cluster.num_nodes, bool)
    in_part[np.asarray(cluster.part_nodes, np.int64)] = True
    for n in nodes:
        if not in_part[n]:   # a node in no partition has no NodeState at all (JobScheduler.cpp:6584-6606)
            assert len(eng.timeline(int(n))["t"]) == 0
            continue
        a, b = eng.timeline(int(n)), ref.timeline(int(n))
        for f in ("t", "cpu_raw", "mem", "core_lo", "core_hi", "gres", "core_w2", "core_w3"):
            assert np.array_equal(a[f], b[f]), f"{tag}: time map of node {n} differs in {f}:\n{a[f]}\n{b[f]}"
