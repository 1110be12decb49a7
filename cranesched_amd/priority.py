"""MultiFactorPriority inputs for the engine (include/crane_gpu/priority.h) as numpy SoA + ctypes views.

Mirror of what `MultiFactorPriority::GetOrderedJobPtrVec` reads (src/CraneCtld/JobScheduler.cpp:7606-7819):
`PrioPending` = the PdJobInScheduler fields of :7664-7690, `PrioRunning` = the RnJobInScheduler fields of
:7692-7746, `PriorityConfig` = g_config.PriorityConfig (CtldPublicDefs.h:162-174).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np


class CnsPriorityConfig(C.Structure):
    _fields_ = [("max_age_sec", C.c_uint64), ("weight_age", C.c_uint32), ("weight_fair_share", C.c_uint32),
                ("weight_job_size", C.c_uint32), ("weight_partition", C.c_uint32), ("weight_qos", C.c_uint32),
                ("favor_small", C.c_uint32)]


class CnsPrioPendingSoa(C.Structure):
    _fields_ = [("num_jobs", C.c_uint32), ("submit_sec", C.c_void_p), ("qos_priority", C.c_void_p),
                ("partition_priority", C.c_void_p), ("node_num", C.c_void_p), ("total_cpu_raw", C.c_void_p),
                ("total_mem", C.c_void_p), ("account", C.c_void_p), ("cached_priority", C.c_void_p)]


class CnsPrioRunningSoa(C.Structure):
    _fields_ = [("num_jobs", C.c_uint32), ("start_sec", C.c_void_p), ("qos_priority", C.c_void_p),
                ("partition_priority", C.c_void_p), ("node_num", C.c_void_p), ("alloc_cpu_raw", C.c_void_p),
                ("alloc_mem", C.c_void_p), ("account", C.c_void_p)]


@dataclass
class PriorityConfig:
    max_age_sec: int = 14 * 86400
    weight_age: int = 500
    weight_fair_share: int = 10000
    weight_job_size: int = 0
    weight_partition: int = 1000
    weight_qos: int = 1000000
    favor_small: bool = True

    def to_c(self) -> CnsPriorityConfig:
        return CnsPriorityConfig(self.max_age_sec, self.weight_age, self.weight_fair_share, self.weight_job_size,
                                 self.weight_partition, self.weight_qos, 1 if self.favor_small else 0)


def _arr(a, dt):
    return np.ascontiguousarray(np.asarray(a, dtype=dt))


class PrioPending:
    def __init__(self, submit_sec, qos_priority, partition_priority, node_num, total_cpu_raw, total_mem, account,
                 cached_priority=None):
        self.submit_sec = _arr(submit_sec, np.int64)
        self.qos_priority = _arr(qos_priority, np.uint32)
        self.partition_priority = _arr(partition_priority, np.uint32)
        self.node_num = _arr(node_num, np.uint32)
        self.total_cpu_raw = _arr(total_cpu_raw, np.int64)
        self.total_mem = _arr(total_mem, np.uint64)
        self.account = _arr(account, np.uint32)
        self.cached_priority = None if cached_priority is None else _arr(cached_priority, np.float64)
        self.num_jobs = len(self.submit_sec)
        for f in ("qos_priority", "partition_priority", "node_num", "total_cpu_raw", "total_mem", "account"):
            assert len(getattr(self, f)) == self.num_jobs, f

    def to_c(self) -> CnsPrioPendingSoa:
        p = lambda a: None if a is None else a.ctypes.data
        return CnsPrioPendingSoa(self.num_jobs, p(self.submit_sec), p(self.qos_priority), p(self.partition_priority),
                                 p(self.node_num), p(self.total_cpu_raw), p(self.total_mem), p(self.account),
                                 p(self.cached_priority))


class PrioRunning:
    def __init__(self, start_sec, qos_priority, partition_priority, node_num, alloc_cpu_raw, alloc_mem, account):
        self.start_sec = _arr(start_sec, np.int64)
        self.qos_priority = _arr(qos_priority, np.uint32)
        self.partition_priority = _arr(partition_priority, np.uint32)
        self.node_num = _arr(node_num, np.uint32)
        self.alloc_cpu_raw = _arr(alloc_cpu_raw, np.int64)
        self.alloc_mem = _arr(alloc_mem, np.uint64)
        self.account = _arr(account, np.uint32)
        self.num_jobs = len(self.start_sec)

    def to_c(self) -> CnsPrioRunningSoa:
        p = lambda a: a.ctypes.data
        return CnsPrioRunningSoa(self.num_jobs, p(self.start_sec), p(self.qos_priority), p(self.partition_priority),
                                 p(self.node_num), p(self.alloc_cpu_raw), p(self.alloc_mem), p(self.account))


def synth_priority_case(J: int, R: int, A: int, seed: int, now: int = 1_700_000_000, cached_frac: float = 0.0):
    """Seeded synthetic inputs (numpy PCG64 — test data, not one of the frozen benchmark queues)."""
    rng = np.random.default_rng(seed)
    cpus = rng.choice([1, 2, 4, 8, 16, 64], J)
    k = rng.choice([1, 1, 1, 2, 4, 8], J)
    pd = PrioPending(
        submit_sec=now - rng.integers(0, 30 * 86400, J), qos_priority=rng.choice([0, 10, 100, 1000], J),
        partition_priority=rng.choice([1, 5, 50], J), node_num=k, total_cpu_raw=cpus * k * 256,
        total_mem=(cpus * k).astype(np.uint64) * np.uint64(2 << 30), account=rng.integers(0, max(A, 1), J),
        cached_priority=(np.where(rng.random(J) < cached_frac, rng.random(J) * 1e6, 0.0) if cached_frac > 0 else None))
    rcpus = rng.choice([1, 4, 16, 128], R)
    rk = rng.choice([1, 2, 16], R)
    rn = PrioRunning(
        start_sec=now - rng.integers(1, 5 * 86400, R), qos_priority=rng.choice([0, 10, 100, 1000], R),
        partition_priority=rng.choice([1, 5, 50], R), node_num=rk, alloc_cpu_raw=rcpus * rk * 256,
        alloc_mem=(rcpus * rk).astype(np.uint64) * np.uint64(4 << 30), account=rng.integers(0, max(A, 1), R))
    return pd, rn, now
