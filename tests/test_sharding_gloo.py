"""The N > 1 path on CPU: two gloo ranks, partition-sharded queue, all-gather of the packed
placement buffers, merge — must reproduce the single-process result bit for bit.

No GPU here, so each rank's shard is computed by the CPU oracle standing in for the engine; what is
under test is everything bench.py adds for N > 1: the partition plan, the shard selection, the
packed-buffer layout (mirror of csrc/engine.hip), the collective and the merge."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pack(pl, jobs):
    from cranesched_amd import sharding
    lay = sharding.results_layout(jobs.num_jobs, jobs.total_places())
    buf = np.zeros(lay["total"], np.uint8)
    for name, (off, elem, n) in lay.items() if False else [(k, v) for k, v in lay.items() if k != "total"]:
        arr = getattr(pl, name)[:n]
        buf[off:off + elem * n] = np.ascontiguousarray(arr).view(np.uint8)
    return buf


def _worker(rank, world, port, q, cfg=("C4", 6000, 512, 8)):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cranesched_amd import sharding, synth
    from oracle import pyoracle
    name, J, N, P = cfg
    running = all_running = None
    if name in synth.LOADED:   # the same queue on a cluster that already runs jobs: every rank holds the whole running set
        cluster, jobs, now, running = synth.make_loaded(name, J=J, N=N, P=P)
        all_running = running
    else:
        cluster, jobs, now = synth.make_config(name, J=J, N=N, P=P)
    # what bench.py gives a rank: a snapshot that lists only its partitions (renumbered), its jobs, its running jobs
    sub, mine, idx = sharding.shard_cluster(cluster, jobs, rank, world)
    if running is not None:
        running = synth.running_of_partitions(cluster, running, sharding.partition_plan(cluster.num_partitions, world, sharding.partition_groups(cluster))[rank])
    r = pyoracle.select(sub, mine, now, running=running)
    buf = pack(r.placements, mine)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([len(buf)], dtype=torch.int64))
    pad = int(max(s.item() for s in sizes))
    send = torch.zeros(pad, dtype=torch.uint8)
    send[:len(buf)] = torch.from_numpy(buf)
    out = torch.empty(pad * world, dtype=torch.uint8)
    dist.all_gather_into_tensor(out, send)
    # every rank now holds all shards: unpack + merge into global queue order
    shards = []
    for rk in range(world):
        sj, sidx = sharding.shard(cluster, jobs, rk, world)
        raw = out[rk * pad:(rk + 1) * pad].numpy()
        shards.append((sharding.unpack_results(raw, sj), sidx))
    merged = sharding.merge(jobs, shards)
    ref = pyoracle.select(cluster, jobs, now, running=all_running)
    q.put((rank, merged.diff(ref.placements)))
    dist.destroy_process_group()


# C4: 8 partitions (rank r owns partitions p % world == r); C4p64: the same cluster cut into 64 partitions of 1 024 nodes — the
# configuration on which more GPUs add chains (bench.py --config C4p64); C4r: the loaded cluster
# C4p256: 256 partitions (one GPU: k_pipe; 4 / 8 GPUs: 64 / 32 busy partitions per rank -> k_wide x8 / x16)
@pytest.mark.parametrize("world,cfg", [(2, ("C4", 6000, 512, 8)), (2, ("C4p64", 8000, 1024, 64)), (4, ("C4p64", 8000, 1024, 64)),
                                       (2, ("C4r", 6000, 512, 8)), (2, ("C4p256", 12000, 2048, 256)), (4, ("C4p256", 12000, 2048, 256)),
                                       (8, ("C4p256", 12000, 2048, 256)),
                                       # BASELINE.json configuration 5 ("backfill window packing ... 8 x MI355X"): C5 scaled, and its
                                       # deep variant (a quarter of the nodes per job, as fullrun's c5deep: most of the queue is backfilled)
                                       (2, ("C5", 6000, 512, 8)), (8, ("C5", 8000, 1024, 8)), (4, ("C5", 9000, 256, 8))])
def test_shard_allgather_merge(world, cfg):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() % 2000) + 7 * world + len(cfg[0])
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, cfg)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, d in res:
        assert d is None, f"rank {rank}: merged result differs from the single-process run: {d}"


def test_partition_plan_covers_everything():
    from cranesched_amd import sharding
    for P in (1, 3, 8):
        for w in (1, 2, 4, 8):
            plan = sharding.partition_plan(P, w)
            assert sorted(p for r in plan for p in r) == list(range(P))


def test_groups_of_partitions_that_share_nodes_stay_on_one_rank():
    import numpy as np
    from cranesched_amd import abi, sharding
    # p0 = {0,1,2}, p1 = {2,3} (shares node 2 with p0), p2 = {4,5}, p3 = {5,6,0} (shares 5 with p2 and 0 with p0)
    parts = [[0, 1, 2], [2, 3], [4, 5], [5, 6, 0], [7]]
    off = np.cumsum([0] + [len(p) for p in parts]).astype(np.uint32)
    n = 8
    c = abi.Cluster(np.full(n, 1024, np.int64), np.full(n, 1 << 30, np.uint64), np.full(n, 0xF, np.uint64), np.zeros(n, np.uint64),
                    np.zeros(n, np.uint64), off, np.concatenate(parts).astype(np.uint32))
    assert sharding.partition_groups(c) == [[0, 1, 2, 3], [4]]
    plan = sharding.partition_plan(5, 2, sharding.partition_groups(c))
    assert plan == [[0, 1, 2, 3], [4]]
