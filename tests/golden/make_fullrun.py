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


def load_case(name, J=None, N=None, P=None):
    """(cluster, jobs, now, running or None) of a CASES entry: a frozen config, or its loaded-cluster variant (synth.LOADED)."""
    if name in synth.LOADED:
        return synth.make_loaded(name, J=J, N=N, P=P)
    return (*synth.make_config(name, J=J, N=N, P=P), None)


def _one_partition(args):
    name, J, N, P, p = args
    from oracle import pyoracle
    cluster, jobs, now, running = load_case(name, J, N, P)
    sub, idx = synth.select_partitions(cluster, jobs, [p])
    r = pyoracle.select(cluster, sub, now, running=None if running is None else synth.running_of_partitions(cluster, running, [p]))
    lo, hi = int(cluster.part_offsets[p]), int(cluster.part_offsets[p + 1])
    nodes = [int(n) for n in fullrun.timeline_nodes(cluster.num_nodes) if lo <= n < hi]  # contiguous partitions (synth)
    assert np.array_equal(cluster.part_nodes[lo:hi], np.arange(lo, hi))
    tl = {n: r.timeline(n) for n in nodes}
    out = (p, idx, r.placements.trimmed(), r.costs()[lo:hi].view(np.uint64).copy(), tl, r.seconds)
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
}


def merge_parts(name, J, N, P, parts):
    """Scatters per-partition oracle results (tuples of _one_partition) back into one full result."""
    cluster, jobs, now, _ = load_case(name, J, N, P)
    Pn = cluster.num_partitions
    full = abi.Placements(jobs.num_jobs, jobs.total_places())
    off = np.concatenate([[0], np.cumsum(jobs.node_num.astype(np.int64))])
    full.place_offsets[:] = off.astype(np.uint64)
    costs = np.zeros(len(cluster.part_nodes), np.uint64)
    timelines, secs = {}, np.zeros(Pn)
    for p, idx, t, c, tl, s in parts:
        full.start_sec[idx] = t["start_sec"]
        full.reason[idx] = t["reason"]
        so = t["place_offsets"].astype(np.int64)
        k = (so[1:] - so[:-1])
        assert np.array_equal(k, off[idx + 1] - off[idx])
        dst = np.repeat(off[idx], k) + (np.arange(so[-1]) - np.repeat(so[:-1], k))
        for f in fullrun.REC_FIELDS:
            getattr(full, f)[dst] = t[f][:so[-1]]
        costs[int(cluster.part_offsets[p]):int(cluster.part_offsets[p + 1])] = c
        timelines.update(tl)
        secs[p] = s
    return cluster, jobs, full, costs, timelines, secs


def merged_run(name, J=None, N=None, P=None, procs=None):
    """(cluster, jobs, Placements, costs_u64, timelines, seconds per partition), one oracle process per partition."""
    Pn = P or synth.CONFIGS[synth.LOADED.get(name, (name,))[0]]["P"]
    with mp.get_context("fork").Pool(procs or min(Pn, os.cpu_count() or 1)) as pool:
        parts = pool.map(_one_partition, [(name, J, N, P, p) for p in range(Pn)], chunksize=1)
    return merge_parts(name, J, N, P, parts)


def main(tags):
    t0 = time.time()
    work = []
    for tag in tags:
        name, J, N, P = CASES[tag]
        work += [(tag, (name, J, N, P, p)) for p in range(P or synth.CONFIGS[synth.LOADED.get(name, (name,))[0]]["P"])]
    with mp.get_context("fork").Pool(min(len(work), os.cpu_count() or 1)) as pool:
        results = pool.map(_one_partition, [w[1] for w in work], chunksize=1)
    for tag in tags:
        name, J, N, P = CASES[tag]
        parts = [r for (t, _), r in zip(work, results) if t == tag]
        cluster, jobs, full, costs, timelines, secs = merge_parts(name, J, N, P, parts)
        d = fullrun.digest(full, costs, lambda n: timelines[n], cluster.num_nodes)
        np.savez_compressed(os.path.join(HERE, f"fullrun_{tag}.npz"), oracle_seconds=secs,
                            jobs=np.array([jobs.num_jobs]), nodes=np.array([cluster.num_nodes]), **d)
        print(f"{tag}: {name} {jobs.num_jobs} jobs x {cluster.num_nodes} nodes, reasons {d['counts'].tolist()}, "
              f"oracle {secs.sum():.1f} core-s (max partition {secs.max():.1f} s), wall {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main([a.lower() for a in sys.argv[1:]] or list(CASES))
