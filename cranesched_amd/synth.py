"""Frozen synthetic queues C1..C5 (SURVEY.md §8d / BASELINE.md §3).

PRNG = splitmix64, seed 0x43524E45 ^ config index; draw f of job j is
mix(seed + GAMMA*(j*NF + f + 1)).  No running jobs, now = 1_700_000_000, FIFO order.
The reference has no workload generator; these distributions are the ones
BASELINE.md records for this project and are the same for the oracle and the engine.
"""
from __future__ import annotations

import numpy as np

from .abi import Cluster, GresLayout, Jobs, MAX_GRES_CLASSES, MAX_GRES_NAMES, Preempt, Running

NOW = 1_700_000_000
SEED0 = 0x43524E45
GAMMA = np.uint64(0x9E3779B97F4A7C15)
GIB = 1 << 30
NF = 8  # random draws per job

CONFIGS = {
    # name: (index, J, N, P, gres, slot quantum, max L multiples)
    "C1": dict(idx=1, J=1_000, N=128, P=1, gres=False, Q=600, LM=24),
    "C2": dict(idx=2, J=100_000, N=4_096, P=1, gres=False, Q=600, LM=24),
    "C3": dict(idx=3, J=1_000_000, N=16_384, P=1, gres=True, Q=600, LM=24),
    "C4": dict(idx=4, J=1_000_000, N=65_536, P=8, gres=True, Q=600, LM=24),
    "C5": dict(idx=5, J=1_000_000, N=65_536, P=8, gres=False, Q=675, LM=32),
    # C4's cluster and job mix cut into 64 partitions of 1 024 nodes: the configuration on which more GPUs DO add chains
    # (one GPU: 64 chains on k_pipe, one CU each; 8 GPUs: 8 chains per GPU on k_wide x64) — bench.py --config C4p64, DESIGN.md 7
    "C4p64": dict(idx=6, J=1_000_000, N=65_536, P=64, gres=True, Q=600, LM=24),
    # ... and into 256 partitions of 256 nodes: beyond k_wide's 80 partitions on ONE GPU (k_pipe, one workgroup per partition);
    # 4 GPUs: 64 busy partitions each -> k_wide x8, 8 GPUs: 32 each -> k_wide x16 (the launch is sized by the partitions that
    # have pending jobs on that rank) — bench.py --config C4p256, DESIGN.md 7
    "C4p256": dict(idx=7, J=1_000_000, N=65_536, P=256, gres=True, Q=600, LM=24),
}


def splitmix64(seed: int, n: int) -> np.ndarray:
    """n outputs of splitmix64 started at `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + GAMMA * np.arange(1, n + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def gres_layout_gpu_npu() -> GresLayout:
    # class 0 = gpu:mi300 (name 0) bits 0..7 ; class 1 = npu:a910 (name 1) bits 8..15
    return GresLayout(class_name=[0, 1], class_shift=[0, 8], class_width=[8, 8])


def make_cluster(N: int, P: int, gres: bool) -> Cluster:
    """Nodes: without GRES all 64c/256GiB; with GRES node i%4 in {0,1}: 64c/256GiB,
    2: 96c/1TiB + 8 x gpu:mi300, 3: 96c/1TiB + 8 x npu:a910.  P disjoint contiguous partitions."""
    idx = np.arange(N)
    kind = (idx % 4) if gres else np.zeros(N, np.int64)
    big = kind >= 2
    cores = np.where(big, 96, 64)
    cpu_total_raw = (cores * 256).astype(np.int64)
    mem_total = np.where(big, 1024 * GIB, 256 * GIB).astype(np.uint64)
    core_lo = np.full(N, np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64)
    core_hi = np.where(big, np.uint64(0xFFFFFFFF), np.uint64(0)).astype(np.uint64)
    gres_slots = np.where(kind == 2, np.uint64(0xFF), np.where(kind == 3, np.uint64(0xFF00), np.uint64(0))).astype(np.uint64)
    assert N % P == 0
    per = N // P
    part_offsets = (np.arange(P + 1) * per).astype(np.uint32)
    part_nodes = idx.astype(np.uint32)
    return Cluster(cpu_total_raw, mem_total, core_lo, core_hi, gres_slots, part_offsets, part_nodes,
                   gres=gres_layout_gpu_npu() if gres else GresLayout())


def make_jobs(J: int, P: int, gres: bool, seed: int, Q: int, LM: int) -> Jobs:
    r = splitmix64(seed, J * NF).reshape(J, NF) >> np.uint64(11)
    cpus = np.array([1, 2, 4, 8], np.int64)[(r[:, 0] % np.uint64(4)).astype(np.int64)]
    L = (Q * (1 + (r[:, 1] % np.uint64(LM)).astype(np.int64))).astype(np.int64)
    partition = (r[:, 2] % np.uint64(P)).astype(np.uint32)
    if gres:
        kd = (r[:, 3] % np.uint64(30)).astype(np.int64)          # 27/30 -> k=1, 1/30 each -> 2,4,8
        k = np.where(kd < 27, 1, np.where(kd == 27, 2, np.where(kd == 28, 4, 8))).astype(np.uint32)
        cls = (r[:, 4] % np.uint64(10)).astype(np.int64)         # 0..6 cpu-only, 7,8 gpu, 9 npu
        cnt = np.array([1, 2, 4, 8], np.uint8)[(r[:, 5] % np.uint64(4)).astype(np.int64)]
        typed = (r[:, 6] % np.uint64(2)).astype(bool)
        gres_total = np.zeros((J, MAX_GRES_NAMES), np.uint8)
        gres_spec = np.zeros((J, MAX_GRES_CLASSES), np.uint8)
        is_gpu, is_npu = (cls == 7) | (cls == 8), cls == 9
        gres_total[is_gpu, 0] = cnt[is_gpu]
        gres_total[is_npu, 1] = cnt[is_npu]
        gres_spec[is_gpu & typed, 0] = cnt[is_gpu & typed]
        gres_spec[is_npu & typed, 1] = cnt[is_npu & typed]
    else:
        k = np.ones(J, np.uint32)
        gres_total = gres_spec = None
    ones = np.ones(J, np.uint32)
    return Jobs(partition=partition, time_limit_sec=L, node_mem=np.zeros(J, np.uint64),
                task_cpu_raw=cpus * 256, task_mem=(cpus * 2 * GIB).astype(np.uint64),
                node_num=k, ntasks=k.copy(), ntasks_per_node_min=ones, ntasks_per_node_max=ones.copy(),
                gres_total=gres_total, gres_spec=gres_spec)


def make_config(name: str, J: int | None = None, N: int | None = None, P: int | None = None):
    """Returns (cluster, jobs, now) of config `name`, optionally scaled to J jobs / N nodes / P partitions
    (same distributions and seed; used for parity cases the CPU oracle finishes in seconds)."""
    c = CONFIGS[name]
    J = c["J"] if J is None else J
    N = c["N"] if N is None else N
    P = c["P"] if P is None else P
    cluster = make_cluster(N, P, c["gres"])
    jobs = make_jobs(J, P, c["gres"], SEED0 ^ c["idx"], c["Q"], c["LM"])
    return cluster, jobs, NOW


# Loaded-cluster ("steady state") variants: the same queue on a cluster that already RUNS jobs — the cycle CraneCtld
# normally executes (running jobs folded in at JobScheduler.cpp:6513-6514,6681-6709; cycle-start res_avail, deep initial
# time maps and non-zero initial costs, JobScheduler.h:301-338,498-511).  name -> (base config, running jobs)
LOADED = {"C4r": ("C4", 300_000), "C2r": ("C2", 20_000), "C5r": ("C5", 300_000)}


def make_running(cluster: Cluster, R: int, seed: int, now: int = NOW) -> Running:
    """R running jobs, frozen like the queue (splitmix64 draws): each inside ONE partition, on k in {1 (80 %), 2, 4, 8}
    distinct nodes, cpus in {1, 2, 4, 8} per node with 2 GiB per cpu, the lowest free core ids of the node at that point
    (what GetFeasibleResourceInNode hands out), 1 or 2 of the node's free GRES slots with probability 1/4 where the node
    has any; end times from 120 s in the past (clamped to now + 1 by the cycle, :6513-6514) to 10 h ahead in 60 s steps.
    A draw that no longer fits its nodes is skipped (the node state is kept consistent: allocations never overlap)."""
    ND = 8
    r = (splitmix64(seed ^ 0x52554E, R * ND).reshape(R, ND) >> np.uint64(11)).astype(np.int64)
    N, P = cluster.num_nodes, cluster.num_partitions
    free_lo = cluster.core_lo.astype(object).copy()
    free_hi = (cluster.core_hi if cluster.core_hi is not None else np.zeros(N, np.uint64)).astype(object).copy()
    free_g = (cluster.gres_slots if cluster.gres_slots is not None else np.zeros(N, np.uint64)).astype(object).copy()
    free_mem = cluster.mem_total.astype(object).copy()
    po = cluster.part_offsets.astype(np.int64)
    pn = cluster.part_nodes.astype(np.int64)
    end, off, node, cpu, mem, lo, hi, g = [], [0], [], [], [], [], [], []
    M64 = (1 << 64) - 1

    def lowest(mask, n):
        out = 0
        for _ in range(n):
            b = mask & -mask
            out |= b
            mask ^= b
        return out

    for i in range(R):
        p = int(r[i, 0] % P)
        cnt = int(po[p + 1] - po[p])
        kd = int(r[i, 1] % 20)
        k = 1 if kd < 16 else (2, 2, 4, 8)[kd - 16]
        k = min(k, cnt)
        cpus = (1, 2, 4, 8)[int(r[i, 2] % 4)]
        first = int(r[i, 3] % cnt)
        stride = 1 + int(r[i, 4] % 7)
        nodes = sorted({int(pn[po[p] + (first + t * stride) % cnt]) for t in range(k)})
        want_g = int(r[i, 5] % 8)           # 0, 1: one / two slots where the node has free ones
        ok = all((bin(int(free_lo[n])).count("1") + bin(int(free_hi[n])).count("1")) >= cpus and int(free_mem[n]) >= cpus * 2 * GIB
                 for n in nodes)
        if not ok or len(nodes) == 0:
            continue
        for n in nodes:
            full = int(free_lo[n]) | (int(free_hi[n]) << 64)
            take = lowest(full, cpus)
            tg = 0
            if want_g < 2 and int(free_g[n]):
                tg = lowest(int(free_g[n]), min(want_g + 1, bin(int(free_g[n])).count("1")))
            free_lo[n] = int(free_lo[n]) & ~(take & M64)
            free_hi[n] = int(free_hi[n]) & ~(take >> 64)
            free_g[n] = int(free_g[n]) & ~tg
            free_mem[n] = int(free_mem[n]) - cpus * 2 * GIB
            node.append(n); cpu.append(cpus * 256); mem.append(cpus * 2 * GIB)
            lo.append(take & M64); hi.append(take >> 64); g.append(tg)
        off.append(len(node))
        end.append(now - 120 + 60 * int(r[i, 6] % 602))
    return Running(end_sec=np.array(end, np.int64), alloc_offsets=np.array(off, np.uint32), alloc_node=np.array(node, np.uint32),
                   alloc_cpu_raw=np.array(cpu, np.int64), alloc_mem=np.array(mem, np.uint64), alloc_core_lo=np.array(lo, np.uint64),
                   alloc_core_hi=np.array(hi, np.uint64), alloc_gres=np.array(g, np.uint64))


def make_loaded(name: str, J: int | None = None, N: int | None = None, P: int | None = None, R: int | None = None):
    """(cluster, jobs, now, running) of a loaded-cluster variant (LOADED), optionally scaled like make_config."""
    base, r_full = LOADED[name]
    cluster, jobs, now = make_config(base, J=J, N=N, P=P)
    if R is None:
        R = max(1, int(r_full * cluster.num_nodes / CONFIGS[base]["N"]))
    return cluster, jobs, now, make_running(cluster, R, SEED0 ^ CONFIGS[base]["idx"], now)


def running_of_partitions(cluster: Cluster, running: Running, parts: list[int], with_index: bool = False):
    """The running jobs whose nodes lie in `parts` (every job of make_running lives inside one partition), order kept
    (with_index: also their indices in `running`)."""
    inpart = np.zeros(cluster.num_nodes, bool)
    for p in parts:
        inpart[np.asarray(cluster.part_nodes[cluster.part_offsets[p]:cluster.part_offsets[p + 1]], np.int64)] = True
    o = running.alloc_offsets.astype(np.int64)
    keep = np.nonzero(inpart[running.alloc_node[o[:-1]].astype(np.int64)])[0] if len(o) > 1 else np.zeros(0, np.int64)
    cnt = (o[1:] - o[:-1])[keep]
    idx = np.repeat(o[:-1][keep], cnt) + (np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt))
    sub = Running(end_sec=running.end_sec[keep], alloc_offsets=np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint32),
                  alloc_node=running.alloc_node[idx], alloc_cpu_raw=running.alloc_cpu_raw[idx], alloc_mem=running.alloc_mem[idx],
                  alloc_core_lo=running.alloc_core_lo[idx], alloc_core_hi=running.alloc_core_hi[idx], alloc_gres=running.alloc_gres[idx])
    return (sub, keep) if with_index else sub


# Mixed cycles: C4-sized cycles in which ONE scheduler needs the slow path and the other seven do not.
#   C4all: C4 plus an "ALL" partition (index 8) over the nodes of partition 0; every second job of partition 0 is submitted to it.
#          Partitions 0 and 8 share their 8 192 nodes (one time map per node, one cost per partition): one group on k_select.
#   C4rp:  the loaded cluster C4r with QoS preemption enabled: 8 % of the pending jobs of partition 0 (1 % of the queue) carry a
#          QoS that may preempt the QoS of everything else; only partition 0's scheduler can ever reach TryPreempt_.
#   C4v:   C4 with 16 reservations (JobScheduler.cpp:6619-6679): on every partition an ACTIVE one over its last 256 nodes (whole-node
#          shares; 3 % of the partition's jobs are submitted into it) and a FUTURE one (starts in an hour, for two hours) over the 256
#          nodes before those: jobs whose window crosses it wait behind it ("Resource Reserved"); a few jobs name the future
#          reservation (its scheduler does not exist yet) or a reservation nobody created ("Reservation Not Found").
#   C4all64k: C4 plus an "ALL" partition (index 8) over ALL 65 536 nodes — the ordinary CraneSched layout on a large site; every ninth
#          job is submitted to it.  ALL connects every partition: one group of 131 072 (partition, node) slots, one chain for the
#          whole queue — wider than k_select's tile: k_wide's home workgroup alone (k_mem).
MIXED = ("C4all", "C4rp", "C4v", "C4all64k")


def mixed_reservations(name: str, cluster: Cluster, now: int = NOW):
    """The reservations of a mixed cycle (None unless it has any)."""
    if name != "C4v":
        return None
    from .abi import Reservations
    P = cluster.num_partitions
    po = cluster.part_offsets.astype(np.int64)
    start, end, off, node = [], [], [0], []
    for v in range(2 * P):
        p = v % P
        per = int(po[p + 1] - po[p])
        w = min(256, per // 8)
        hi = po[p + 1] - (0 if v < P else w)
        node.append(cluster.part_nodes[hi - w:hi].astype(np.int64))
        off.append(off[-1] + w)
        if v < P: start.append(now - 1000 - 7 * p); end.append(now + 8 * 3600 + 60 * p)
        else: start.append(now + 3600 + 120 * p); end.append(now + 3 * 3600 + 120 * p)
    node = np.concatenate(node)
    return Reservations(start, end, off, node, cluster.cpu_total_raw[node], cluster.mem_total[node], cluster.core_lo[node],
                        cluster.core_hi[node], cluster.gres_slots[node])


def make_mixed(name: str, J: int | None = None, N: int | None = None):
    """(cluster, jobs, now, running or None, Preempt or None) of a mixed cycle, optionally scaled like make_config."""
    if name == "C4all":
        c, j, now = make_config("C4", J=J, N=N)
        P = c.num_partitions
        po = c.part_offsets.astype(np.int64)
        p0 = c.part_nodes[po[0]:po[1]]
        cluster = Cluster(c.cpu_total_raw, c.mem_total, c.core_lo, c.core_hi, c.gres_slots,
                          np.concatenate([c.part_offsets, [po[-1] + len(p0)]]).astype(np.uint32),
                          np.concatenate([c.part_nodes, p0]).astype(np.uint32), gres=c.gres)
        part = j.partition.copy()
        part[(part == 0) & (np.arange(j.num_jobs) % 2 == 1)] = P
        j.partition = part.astype(np.uint32)
        return cluster, j, now, None, None
    if name == "C4all64k":
        c, j, now = make_config("C4", J=J, N=N)
        Nn = c.num_nodes
        cluster = Cluster(c.cpu_total_raw, c.mem_total, c.core_lo, c.core_hi, c.gres_slots,
                          np.concatenate([c.part_offsets, [int(c.part_offsets[-1]) + Nn]]).astype(np.uint32),
                          np.concatenate([c.part_nodes, np.arange(Nn, dtype=np.uint32)]).astype(np.uint32), gres=c.gres)
        part = j.partition.copy()
        part[np.arange(j.num_jobs) % 9 == 8] = c.num_partitions
        j.partition = part.astype(np.uint32)
        return cluster, j, now, None, None
    if name == "C4rp":
        c, j, now, run = make_loaded("C4r", J=J, N=N)
        R = len(run.end_sec)
        r = (splitmix64(SEED0 ^ 0x50524545, j.num_jobs) >> np.uint64(11)).astype(np.int64)
        pd_qos = np.where((j.partition == 0) & (r % 100 < 8), 1, 0).astype(np.uint32)
        qprio = np.array([10, 20], np.uint32)
        pre = Preempt([[], [0]], np.arange(j.num_jobs, dtype=np.uint32) + 1, pd_qos, qprio[pd_qos],
                      (j.num_jobs - np.arange(j.num_jobs)).astype(np.float64) + 0.5,        # all different: nothing left unordered
                      1_000_000 + np.arange(R, dtype=np.uint32), np.zeros(R, np.uint32), np.full(R, 10, np.uint32),
                      now - 1 - np.arange(R, dtype=np.int64))                              # all different
        return c, j, now, run, pre
    if name == "C4v":
        c, j, now = make_config("C4", J=J, N=N)
        P = c.num_partitions
        r = (splitmix64(SEED0 ^ 0x52455356, j.num_jobs) >> np.uint64(11)).astype(np.int64) % 1000
        part = j.partition.astype(np.int64)
        resv = np.full(j.num_jobs, 0xFFFFFFFF, np.int64)                  # RESV_NONE
        resv = np.where(r < 30, part, resv)                              # the partition's active reservation
        resv = np.where((r >= 30) & (r < 32), P + part, resv)            # its future one
        resv = np.where(r == 32, 2 * P, resv)                            # one that does not exist
        j.reservation = resv.astype(np.uint32)
        return c, j, now, None, None
    raise KeyError(name)


def preempt_subset(pre: Preempt, job_idx: np.ndarray, run_idx: np.ndarray) -> Preempt:
    """The preemption inputs of a shard: pending jobs `job_idx` and running jobs `run_idx` of the whole cycle."""
    return Preempt(pre.qos_preempt, pre.pd_job_id[job_idx], pre.pd_qos[job_idx], pre.pd_qos_priority[job_idx], pre.pd_priority[job_idx],
                   pre.rn_job_id[run_idx], pre.rn_qos[run_idx], pre.rn_qos_priority[run_idx], pre.rn_start_sec[run_idx],
                   preempting=pre.preempting, enabled=pre.enabled)


def make_limits(name: str, cluster: Cluster, jobs: Jobs):
    """The accounts / QoS side of a config (C4: "64 accounts x 4 QoS with max_jobs_per_user, max_tres_per_account",
    SURVEY.md §8d), frozen like the queue: 8 root accounts with 7 children each, 1024 users (16 per account, one
    account each), job -> (user, qos) from draw 7 of the job.  QoS 0/1 cap running jobs per user (96 / 192) and CPUs
    per account (3000 / 6000 cores at every level of the tree), QoS 2 caps the QoS's total GPUs (4096), QoS 3 caps
    nothing; account x partition limits (max 1500 jobs) on every 4th account.  Every memory limit is set to 2^60:
    the reference's "unlimited" memory is kMaxJobMemoryBytes = 10 000 GB (DbClient.cpp:420-428), which as a cap on a
    SUM of allocations would stop a 64 k-node cluster after ~1 400 jobs per QoS.
    Returns (LimitTables, LimitJobs) in pending-vector order (= FIFO order = NodeSelect order)."""
    from . import limits as lm
    c = CONFIGS[name]
    J = jobs.num_jobs
    U, A, Q, Pn = 1024, 64, 4, max(cluster.num_partitions, 1)
    r7 = splitmix64(SEED0 ^ c["idx"], J * NF).reshape(J, NF)[:, 7] >> np.uint64(11)
    user = (r7 % np.uint64(U)).astype(np.uint32)
    qos = ((r7 // np.uint64(U)) % np.uint64(Q)).astype(np.uint32)
    account = (user % np.uint32(A)).astype(np.uint32)
    parent = np.where(np.arange(A) < 8, lm.LIM_NONE, np.arange(A) % 8).astype(np.uint32)
    big = 1 << 60
    free = lm.tres(mem=big)
    q = np.array([lm.qos_limits(max_jobs_per_user=96, max_tres_per_account=lm.tres(cpu=3000, mem=big), max_tres=free, max_tres_per_user=free),
                  lm.qos_limits(max_jobs_per_user=192, max_tres_per_account=lm.tres(cpu=6000, mem=big), max_tres=free, max_tres_per_user=free),
                  lm.qos_limits(max_tres=lm.tres(mem=big, names={0: 4096} if c["gres"] else None), max_tres_per_user=free, max_tres_per_account=free),
                  lm.qos_limits(max_tres=free, max_tres_per_user=free, max_tres_per_account=free)], lm.QOS_DT)
    pl = np.array([lm.part_limit(max_jobs=1500, max_tres=free)], lm.PART_LIMIT_DT)
    apl = np.where((np.arange(A * Pn) // Pn) % 4 == 3, 0, lm.LIM_NONE).astype(np.uint32)
    tables = lm.LimitTables(num_users=U, num_user_accts=U, num_partitions=Pn, qos=q, acct_parent=parent,
                            part_limits=pl, acct_part_limit=apl)
    part = np.minimum(jobs.partition, Pn - 1).astype(np.uint32)
    lj = lm.LimitJobs(user=user, user_acct=user, account=account, qos=qos, partition=part,
                      time_limit_sec=jobs.time_limit_sec)
    return tables, lj


def select_partitions(cluster: Cluster, jobs: Jobs, parts: list[int]):
    """Shard of a queue: the jobs (order preserved) and node lists of `parts` only, with node and
    partition indices unchanged.  Used to job-shard disjoint partitions across GPUs."""
    keep = np.isin(jobs.partition, np.asarray(parts, np.uint32))
    idx = np.nonzero(keep)[0]

    def take(a):
        return None if a is None else a[idx]

    sub = Jobs(partition=jobs.partition[idx], time_limit_sec=jobs.time_limit_sec[idx],
               node_mem=jobs.node_mem[idx], task_cpu_raw=jobs.task_cpu_raw[idx], task_mem=jobs.task_mem[idx],
               node_num=jobs.node_num[idx], ntasks=jobs.ntasks[idx],
               ntasks_per_node_min=jobs.ntasks_per_node_min[idx],
               ntasks_per_node_max=jobs.ntasks_per_node_max[idx], node_cpu_raw=take(jobs.node_cpu_raw),
               exclusive=take(jobs.exclusive), gres_total=take(jobs.gres_total), gres_spec=take(jobs.gres_spec),
               skip=take(jobs.skip), reservation=take(jobs.reservation))
    return sub, idx
