"""Job-sharding of the pending queue across the GPUs of one node.

Each partition has its own LocalScheduler, node set and cost order in the reference
(src/CraneCtld/JobScheduler.cpp:6723-6732, :6746-6761), so jobs of different partitions never
interact when the partitions' node sets are disjoint.  Rank r therefore owns the partitions
{p : p % world == r}: their node lists and their slice of the queue (order preserved), runs the
engine on that shard alone, and one RCCL all-gather of the packed placement buffers leaves every
rank with the merged claim list.  Partitions that SHARE nodes also share those nodes' time maps
(one NodeState per craned, :6563,6609-6617): they form a group that must stay on one rank (the
engine runs a group as one workgroup, in queue order), so the unit of sharding is the group of
partitions connected through shared nodes; groups never interact, and the merge has no claim to
resolve.
"""
from __future__ import annotations

import numpy as np

from . import abi, synth


def partition_groups(cluster: abi.Cluster) -> list[list[int]]:
    """Partitions connected through shared nodes, each group ascending, groups ordered by their first partition
    (the same union-find as cns_set_nodes in csrc/engine.hip)."""
    P = cluster.num_partitions
    uf = list(range(P))

    def find(x):
        while uf[x] != x:
            uf[x] = uf[uf[x]]
            x = uf[x]
        return x

    first = {}
    for p in range(P):
        for n in cluster.part_nodes[cluster.part_offsets[p]:cluster.part_offsets[p + 1]]:
            n = int(n)
            if n in first:
                a, b = find(first[n]), find(p)
                if a != b:
                    uf[max(a, b)] = min(a, b)
            else:
                first[n] = p
    groups: dict[int, list[int]] = {}
    for p in range(P):
        groups.setdefault(find(p), []).append(p)
    return [groups[r] for r in sorted(groups)]


def partition_plan(num_parts: int, world: int, groups: list[list[int]] | None = None) -> list[list[int]]:
    """Partitions of every rank: group g goes to rank g % world (without shared nodes a group is one partition)."""
    groups = groups if groups is not None else [[p] for p in range(num_parts)]
    plan = [[] for _ in range(world)]
    for g, members in enumerate(groups):
        plan[g % world] += members
    return [sorted(x) for x in plan]


def shard(cluster: abi.Cluster, jobs: abi.Jobs, rank: int, world: int):
    """(jobs of this rank's partitions, their indices in the global queue)."""
    parts = partition_plan(cluster.num_partitions, world, partition_groups(cluster))[rank]
    return synth.select_partitions(cluster, jobs, parts)


def shard_cluster(cluster: abi.Cluster, jobs: abi.Jobs, rank: int, world: int):
    """(snapshot of this rank, its jobs, their indices in the global queue): the node table keeps every node (node indices stay
    global, so placements need no translation) but lists only this rank's partitions, renumbered 0.. in ascending order, and the
    jobs' partition ids are renumbered with them.  The rank's engine then holds time maps, costs and a launch only for its own
    partitions (the reference builds NodeStates only for partitions with pending jobs, JobScheduler.cpp:6571-6573)."""
    parts = partition_plan(cluster.num_partitions, world, partition_groups(cluster))[rank]
    mine, idx = synth.select_partitions(cluster, jobs, parts)
    local = np.full(cluster.num_partitions, 0xFFFFFFFF, np.uint32)
    local[np.asarray(parts, np.int64)] = np.arange(len(parts), dtype=np.uint32)
    po, pn = [0], []
    for p in parts:
        pn.append(np.asarray(cluster.part_nodes[cluster.part_offsets[p]:cluster.part_offsets[p + 1]], np.uint32))
        po.append(po[-1] + len(pn[-1]))
    import dataclasses
    sub = dataclasses.replace(cluster, part_offsets=np.asarray(po, np.uint32),
                              part_nodes=np.concatenate(pn) if pn else np.zeros(0, np.uint32))
    mine.partition = local[mine.partition.astype(np.int64)]
    return sub, mine, idx


def _align16(x: int) -> int:
    return (x + 15) & ~15


def results_layout(num_jobs: int, places: int, wide_cores: bool = False) -> dict:
    """Byte offsets of the packed result buffer (mirror of cns_upload_jobs in csrc/engine.hip).  `wide_cores`: the snapshot has
    a node with a core id above 127 (abi.Cluster.wide_cores) — only then the buffer ends with the core_w2 / core_w3 planes."""
    off, lay = 0, {}
    for name, elem, n in (("start_sec", 8, num_jobs), ("cpu_raw", 8, places), ("mem", 8, places),
                          ("core_lo", 8, places), ("core_hi", 8, places), ("gres", 8, places),
                          ("node_idx", 4, places), ("ntasks", 4, places), ("reason", 1, num_jobs)) + \
                         ((("core_w2", 8, places), ("core_w3", 8, places)) if wide_cores else ()):
        lay[name] = (off, elem, n)
        off = _align16(off + elem * max(n, 1))
    lay["total"] = off
    return lay


_DT = {"start_sec": np.int64, "cpu_raw": np.int64, "mem": np.uint64, "core_lo": np.uint64, "core_hi": np.uint64,
       "gres": np.uint64, "node_idx": np.uint32, "ntasks": np.uint32, "reason": np.uint8, "core_w2": np.uint64, "core_w3": np.uint64}


def unpack_results(buf: np.ndarray, jobs: abi.Jobs, wide_cores: bool = False) -> abi.Placements:
    """Packed result bytes of one shard -> Placements (host side of the merge)."""
    J, places = jobs.num_jobs, jobs.total_places()
    lay = results_layout(J, places, wide_cores)
    out = abi.Placements(J, places)
    for name, dt in _DT.items():
        if name not in lay:
            continue
        off, elem, n = lay[name]
        getattr(out, name)[:n] = np.frombuffer(buf, dtype=dt, count=n, offset=off)
    out.place_offsets[:] = np.concatenate([[0], np.cumsum(jobs.node_num.astype(np.uint64))])
    return out


def merge(jobs: abi.Jobs, shards: list[tuple[abi.Placements, np.ndarray]]) -> abi.Placements:
    """Scatter per-shard placements (with their global job indices) back into global queue order.
    Jobs owned by no shard (unknown partition, beyond the batch limit) keep what their shard
    reported — every job belongs to exactly one shard by construction of `shard`."""
    J, places = jobs.num_jobs, jobs.total_places()
    out = abi.Placements(J, places)
    goff = np.concatenate([[0], np.cumsum(jobs.node_num.astype(np.uint64))]).astype(np.uint64)
    out.place_offsets[:] = goff
    for pl, idx in shards:
        out.start_sec[idx] = pl.start_sec[:len(idx)]
        out.reason[idx] = pl.reason[:len(idx)]
        k = jobs.node_num[idx].astype(np.int64)
        if len(idx) == 0:
            continue
        # destination record index of every shard record
        starts = goff[idx].astype(np.int64)
        rep = np.repeat(starts - np.concatenate([[0], np.cumsum(k)[:-1]]), k) + np.arange(int(k.sum()))
        for f in ("node_idx", "ntasks", "cpu_raw", "mem", "core_lo", "core_hi", "gres", "core_w2", "core_w3"):
            getattr(out, f)[rep] = getattr(pl, f)[:len(rep)]
    return out


def device_bytes_tensor(ptr: int, nbytes: int, device):
    """uint8 torch tensor aliasing `nbytes` of HBM at `ptr` (for torch.distributed / RCCL)."""
    import torch

    class _Raw:
        __cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}

    return torch.as_tensor(_Raw(), device=device)
