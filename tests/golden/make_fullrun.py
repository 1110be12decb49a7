#!/usr/bin/env python3
"""Full-size oracle runs of BASELINE.json's configurations -> tests/golden/fullrun_<cfg>.npz (digests only).

    python tests/golden/make_fullrun.py [tag ...]          # default: every case; c3 is one 1 M-job chain (tens of minutes)

Partitions with disjoint node sets never interact (src/CraneCtld/JobScheduler.cpp:6723-6732,6746-6761), so the
oracle runs ONE PARTITION PER PROCESS (synth.select_partitions keeps node and job indices) and the per-partition
results are scattered back into the full result before it is digested (tests/fullrun.py).  The merge is itself
checked: for C4 at reduced size the merged result equals one oracle run over all partitions (tests/test_fullrun.py).

These are ORACLE outputs (the reference cannot be built here): they pin the engine to the oracle
over the WHOLE queues, i.e. in the loaded-cluster / backfill regime that prefixes of the queue never reach.
Also records the oracle's single-core seconds per partition: the same-queue CPU baseline of bench.py.
"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cranesched_amd import abi, synth  # noqa: E402
from tests import fullrun  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def load_case5(name, J=None, N=None, P=None):
    """(cluster, jobs, now, running or None, Preempt or None) of a CASES entry: a frozen config, its loaded-cluster variant
    (synth.LOADED) or a mixed cycle (synth.MIXED)."""
    if name in synth.MIXED:
        return synth.make_mixed(name, J=J, N=N)
    if name in synth.LOADED:
        return (*synth.make_loaded(name, J=J, N=N, P=P), None)
    return (*synth.make_config(name, J=J, N=N, P=P), None, None)


def load_case(name, J=None, N=None, P=None):
    return load_case5(name, J, N, P)[:4]


def load_resv(name, cluster):
    """The reservations of a CASES entry (None for most)."""
    return synth.mixed_reservations(name, cluster) if name in synth.MIXED else None


def case_groups(name, J=None, N=None, P=None):
    """The units that never interact: groups of partitions connected through shared nodes (mostly single partitions)."""
    from cranesched_amd import sharding
    return sharding.partition_groups(load_case5(name, J, N, P)[0])


def _one_partition(args):
    """One oracle run over one GROUP of partitions (a single partition unless partitions share nodes)."""
    name, J, N, P, group = args
    from oracle import pyoracle
    cluster, jobs, now, running, pre = load_case5(name, J, N, P)
    sub, idx = synth.select_partitions(cluster, jobs, group)
    rsub, rkeep = (None, np.zeros(0, np.int64)) if running is None else synth.running_of_partitions(cluster, running, group, with_index=True)
    psub = None if pre is None else synth.preempt_subset(pre, idx, rkeep)
    # (reservations: all of them in every group's run — one over another group's nodes only leaves dips on nodes nobody here looks at)
    r = pyoracle.select(cluster, sub, now, running=rsub, preempt=psub, reservations=load_resv(name, cluster))
    in_group = np.zeros(cluster.num_nodes, bool)
    slots = []
    for p in group:
        lo, hi = int(cluster.part_offsets[p]), int(cluster.part_offsets[p + 1])
        in_group[np.asarray(cluster.part_nodes[lo:hi], np.int64)] = True
        slots.append((lo, hi))
    nodes = [int(n) for n in fullrun.timeline_nodes(cluster.num_nodes) if in_group[n]]
    tl = {n: r.timeline(n) for n in nodes}
    costs = r.costs().view(np.uint64)
    pre_pairs = None
    if pre is not None:   # (pending job, reference) in GLOBAL indices: reference = running index, or pending index | 2^31
        pairs = []
        for sj, lst in enumerate(r.preempt_out.lists()):
            for is_pd, ref in lst:
                pairs.append((int(idx[sj]), (int(idx[ref]) | (1 << 31)) if is_pd else int(rkeep[ref])))
        pre_pairs = (np.asarray(pairs, np.int64).reshape(-1, 2), np.asarray(r.preempt_out.cancelled_ids(), np.int64))
    out = (group, idx, r.placements.trimmed(), [(lo, hi, costs[lo:hi].copy()) for lo, hi in slots], tl, r.seconds, pre_pairs)
    r.close()
    return out


# tag -> (config, J, N, P); None = the configuration's own size.  The `tile*` cases are C3's job mix (GRES, multi-node
# jobs) on ONE partition sized for each register-tile width of k_select (nodes per scanner lane 1,3,10,19,28,37), with
# J ~ 16 N so that the cluster fills and >= 20 % of the jobs are backfilled into time maps with several entries.
CASES = {
    "c2": ("C2", None, None, None), "c3": ("C3", None, None, None), "c4": ("C4", None, None, None),
    "c5": ("C5", None, None, None),
    # the frozen C5 queue (3.75 M cores of demand on 4.19 M cores) never fills its cluster: every job starts now.  The
    # same 1 M-job queue on a quarter of the nodes is the walltime-packing regime C5 is named after (72 % backfilled).
    "c5deep": ("C5", None, 16_384, 8),
    "tile1": ("C3", 6_000, 380, 1), "tile3": ("C3", 18_000, 1_100, 1), "tile10": ("C3", 65_000, 4_100, 1),
    "tile19": ("C3", 130_000, 8_192, 1), "tile28": ("C3", 160_000, 10_000, 1), "tile37": ("C3", 200_000, 13_000, 1),
    # the cycle CraneCtld normally runs: the same 1 M-job queue on a cluster that already RUNS 300 k jobs (480 k allocations:
    # cycle-start res_avail != res_total, initial time maps of up to ~30 entries, non-zero initial costs; synth.make_running)
    "c4r": ("C4r", None, None, None),
    # mixed cycles (synth.MIXED): ONE scheduler needs k_select — an ALL partition over partition 0's nodes / QoS preemption among
    # partition 0's jobs — the other seven run on k_wide in the same cycle
    "c4all": ("C4all", None, None, None), "c4rp": ("C4rp", None, None, None),
    # C4 with 8 active and 8 future reservations (24 schedulers: k_wide's nine-workgroup build)
    "c4v": ("C4v", None, None, None),
    # C4's cluster and mix cut into 64 partitions of 1 024 nodes (the configuration on which more GPUs add chains): on one
    # GPU the three-workgroup build of k_wide (8 scanner waves per partition), k_pipe under CNS_SELECT_KERNEL=pipe
    "c4p64": ("C4p64", None, None, None),
    # C4 + an ALL partition over all 65 536 nodes (synth.MIXED C4all64k): ONE group of 131 072 slots, the whole queue one chain — on
    # k_wide's home workgroup alone (k_mem).  The first 300 000 jobs of the queue (the oracle's single process: minutes; the engine's
    # chain: ~20 s); the cluster is at full size, which is what the case is about.
    "c4all64k": ("C4all64k", 300_000, None, None),
}


def merge_parts(name, J, N, P, parts):
    """Scatters per-partition oracle results (tuples of _one_partition) back into one full result."""
    cluster, jobs, now, _ = load_case(name, J, N, P)
    ngroups = len(parts)
    full = abi.Placements(jobs.num_jobs, jobs.total_places())
    off = np.concatenate([[0], np.cumsum(jobs.node_num.astype(np.int64))])
    full.place_offsets[:] = off.astype(np.uint64)
    costs = np.zeros(len(cluster.part_nodes), np.uint64)
    timelines, secs = {}, np.zeros(ngroups)
    pre_pairs, cancelled = [], []
    for gi, (group, idx, t, cs, tl, s, pp) in enumerate(parts):
        full.start_sec[idx] = t["start_sec"]
        full.reason[idx] = t["reason"]
        so = t["place_offsets"].astype(np.int64)
        k = (so[1:] - so[:-1])
        assert np.array_equal(k, off[idx + 1] - off[idx])
        dst = np.repeat(off[idx], k) + (np.arange(so[-1]) - np.repeat(so[:-1], k))
        for f in fullrun.REC_FIELDS:
            getattr(full, f)[dst] = t[f][:so[-1]]
        for lo, hi, c in cs:
            costs[lo:hi] = c
        timelines.update(tl)
        secs[gi] = s
        if pp is not None:
            pre_pairs.append(pp[0]); cancelled.append(pp[1])
    if pre_pairs:   # preempted_jobs lists of the whole cycle, by pending job (stable: push_back order within a job), and the cancel list
        allp = np.concatenate(pre_pairs)
        merge_parts.preempt = (allp[np.argsort(allp[:, 0], kind="stable")], np.sort(np.concatenate(cancelled)))
    else:
        merge_parts.preempt = None
    return cluster, jobs, full, costs, timelines, secs


def merged_run(name, J=None, N=None, P=None, procs=None):
    """(cluster, jobs, Placements, costs_u64, timelines, seconds per partition), one oracle process per partition."""
    groups = case_groups(name, J, N, P)
    with mp.get_context("fork").Pool(procs or min(len(groups), os.cpu_count() or 1)) as pool:
        parts = pool.map(_one_partition, [(name, J, N, P, g) for g in groups], chunksize=1)
    return merge_parts(name, J, N, P, parts)


def main(tags):
    t0 = time.time()
    work = []
    for tag in tags:
        name, J, N, P = CASES[tag]
        work += [(tag, (name, J, N, P, g)) for g in case_groups(name, J, N, P)]
    with mp.get_context("fork").Pool(min(len(work), os.cpu_count() or 1)) as pool:
        results = pool.map(_one_partition, [w[1] for w in work], chunksize=1)
    for tag in tags:
        name, J, N, P = CASES[tag]
        parts = [r for (t, _), r in zip(work, results) if t == tag]
        cluster, jobs, full, costs, timelines, secs = merge_parts(name, J, N, P, parts)
        d = fullrun.digest(full, costs, lambda n: timelines[n], cluster.num_nodes)
        if merge_parts.preempt is not None:
            d["preempt_crc"] = fullrun.preempt_crc(*merge_parts.preempt)
            d["preemptions"] = np.array([len(merge_parts.preempt[0]), len(merge_parts.preempt[1])])
        np.savez_compressed(os.path.join(HERE, f"fullrun_{tag}.npz"), oracle_seconds=secs,
                            jobs=np.array([jobs.num_jobs]), nodes=np.array([cluster.num_nodes]), **d)
        print(f"{tag}: {name} {jobs.num_jobs} jobs x {cluster.num_nodes} nodes, reasons {d['counts'].tolist()}, "
              f"oracle {secs.sum():.1f} core-s (max partition {secs.max():.1f} s), wall {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main([a.lower() for a in sys.argv[1:]] or list(CASES))
