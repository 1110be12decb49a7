"""Run-limit admission inputs (include/crane_gpu/run_limits.h) as numpy tables + ctypes views.

Mirror of what `AccountMetaContainer::CheckAndMallocMetaResource` reads and writes
(src/CraneCtld/Accounting/AccountMetaContainer.cpp:180-224,891-1124): `LimitTables` = the limits (Qos,
PartitionResourceLimit; src/CraneCtld/Account/AccountDefs.h:27-49,163-175) and the usage maps
(MetaResource, AccountMetaContainer.h:30-80) at the start of the commit loop; `LimitJobs` = the keys of the
pending vector (JobScheduler.cpp:1492).  Pure plumbing: no admission logic here.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import abi

LIM_NONE = 0xFFFFFFFF
MAX_CHAIN = 6
UNLIMITED_JOBS = 0xFFFFFFFF
UNLIMITED_CPU_RAW = 1 << 53
MAX_JOB_MEMORY = 10737418240000
NOT_CANDIDATE = 255

# The reference's strings (AccountMetaContainer.cpp; CheckTres_'s prefix defaults to "Qos", AccountMetaContainer.h:179-181):
# codes 2 and 5 are different checks (max_cpus_per_user / a max_tres* cpu count) that report the same string.
LIMIT_REASON_STR = {
    0: "", 1: "QosEntryNotFound", 2: "QosCpuResourceLimit", 3: "QosJobsResourceLimit", 4: "QosWallTimeLimit",
    5: "QosCpuResourceLimit", 6: "QosMemResourceLimit", 7: "QosGresResourceLimit", 8: "PartitionEntryNotFound",
    9: "UserPartitionJobsLimit", 10: "UserPartitionWallTimeLimit", 11: "AccPartitionJobsLimit",
    12: "AccPartitionWallTimeLimit", 13: "PartitionCpuResourceLimit", 14: "PartitionMemResourceLimit",
    15: "PartitionGresResourceLimit", 255: "<not a candidate>",
}

# numpy images of the C structs (all naturally aligned, no padding holes)
TRES_DT = np.dtype([("cpu_raw", "<i8"), ("mem", "<u8"), ("name_mask", "<u4"), ("class_mask", "<u4"),
                    ("name_total", "<u8", (abi.MAX_GRES_NAMES,)), ("class_count", "<u8", (abi.MAX_GRES_CLASSES,))])
QOS_DT = np.dtype([("max_jobs_per_user", "<u4"), ("max_jobs_per_account", "<u4"), ("max_jobs", "<u4"),
                   ("reserved0", "<u4"), ("max_cpus_per_user_raw", "<i8"), ("max_wall_sec", "<i8"),
                   ("max_tres", TRES_DT), ("max_tres_per_user", TRES_DT), ("max_tres_per_account", TRES_DT)])
PART_LIMIT_DT = np.dtype([("max_jobs", "<u4"), ("reserved0", "<u4"), ("max_wall_sec", "<i8"), ("max_tres", TRES_DT)])
USAGE_DT = np.dtype([("cpu_raw", "<i8"), ("mem", "<u8"), ("wall_sec", "<i8"), ("jobs_count", "<u4"),
                     ("reserved0", "<u4"), ("name_total", "<u8", (abi.MAX_GRES_NAMES,)),
                     ("class_count", "<u8", (abi.MAX_GRES_CLASSES,))])
assert TRES_DT.itemsize == 120 and QOS_DT.itemsize == 392 and PART_LIMIT_DT.itemsize == 136 and USAGE_DT.itemsize == 128

_P = C.c_void_p


class CnsLimitTables(C.Structure):
    _fields_ = [("num_users", C.c_uint32), ("num_user_accts", C.c_uint32), ("num_accounts", C.c_uint32),
                ("num_qos", C.c_uint32), ("num_partitions", C.c_uint32), ("num_part_limits", C.c_uint32),
                ("qos", _P), ("acct_parent", _P), ("part_limits", _P), ("user_part_limit", _P),
                ("acct_part_limit", _P), ("user_qos", _P), ("user_qos_exists", _P), ("user_part", _P),
                ("user_part_exists", _P), ("acct_qos", _P), ("acct_qos_exists", _P), ("acct_part", _P),
                ("acct_part_exists", _P), ("qos_usage", _P)]


class CnsLimitJobSoa(C.Structure):
    _fields_ = [("num_jobs", C.c_uint64), ("select_index", _P), ("user", _P), ("user_acct", _P), ("account", _P),
                ("qos", _P), ("partition", _P), ("time_limit_sec", _P), ("skip", _P)]


class CnsLimitTiming(C.Structure):
    _fields_ = [("h2d_ms", C.c_double), ("prep_ms", C.c_double), ("admit_ms", C.c_double), ("d2h_ms", C.c_double),
                ("candidates", C.c_uint64), ("admitted", C.c_uint64), ("rounds", C.c_uint32),
                ("ordered_fallback", C.c_uint32)]


def unlimited_tres() -> np.ndarray:
    """The default ResourceView of a Qos / PartitionResourceLimit: IsUnlimitedTres_ (AccountMetaContainer.cpp:362)."""
    t = np.zeros((), TRES_DT)
    t["cpu_raw"], t["mem"] = UNLIMITED_CPU_RAW, MAX_JOB_MEMORY
    return t


def tres(cpu=None, mem=None, names: Optional[dict] = None, classes: Optional[dict] = None) -> np.ndarray:
    """cpu in cores (None = unlimited), names {name idx: total}, classes {class idx: count}."""
    t = unlimited_tres()
    if cpu is not None:
        t["cpu_raw"] = int(round(cpu * 256))
    if mem is not None:
        t["mem"] = mem
    for n, v in (names or {}).items():
        t["name_mask"] |= 1 << n
        t["name_total"][n] = v
    for g, v in (classes or {}).items():
        t["class_mask"] |= 1 << g
        t["class_count"][g] = v
    return t


def qos_limits(max_jobs_per_user=UNLIMITED_JOBS, max_jobs_per_account=UNLIMITED_JOBS, max_jobs=UNLIMITED_JOBS,
               max_cpus_per_user=None, max_wall_sec=0, max_tres=None, max_tres_per_user=None,
               max_tres_per_account=None) -> np.ndarray:
    q = np.zeros((), QOS_DT)
    q["max_jobs_per_user"], q["max_jobs_per_account"], q["max_jobs"] = max_jobs_per_user, max_jobs_per_account, max_jobs
    q["max_cpus_per_user_raw"] = UNLIMITED_CPU_RAW if max_cpus_per_user is None else int(round(max_cpus_per_user * 256))
    q["max_wall_sec"] = max_wall_sec
    q["max_tres"] = unlimited_tres() if max_tres is None else max_tres
    q["max_tres_per_user"] = unlimited_tres() if max_tres_per_user is None else max_tres_per_user
    q["max_tres_per_account"] = unlimited_tres() if max_tres_per_account is None else max_tres_per_account
    return q


def part_limit(max_jobs=UNLIMITED_JOBS, max_wall_sec=0, max_tres=None) -> np.ndarray:
    p = np.zeros((), PART_LIMIT_DT)
    p["max_jobs"], p["max_wall_sec"] = max_jobs, max_wall_sec
    p["max_tres"] = unlimited_tres() if max_tres is None else max_tres
    return p


@dataclass
class LimitTables:
    num_users: int
    num_user_accts: int
    num_partitions: int
    qos: np.ndarray                        # [Q] QOS_DT
    acct_parent: np.ndarray                # [A] u32, LIM_NONE for a root
    part_limits: np.ndarray = field(default_factory=lambda: np.zeros(0, PART_LIMIT_DT))
    user_part_limit: Optional[np.ndarray] = None   # [UA*Pn] u32
    acct_part_limit: Optional[np.ndarray] = None   # [A*Pn] u32
    user_qos: Optional[np.ndarray] = None          # [U*Q] USAGE_DT
    user_qos_exists: Optional[np.ndarray] = None   # u8
    user_part: Optional[np.ndarray] = None         # [UA*Pn]
    user_part_exists: Optional[np.ndarray] = None
    acct_qos: Optional[np.ndarray] = None          # [A*Q]
    acct_qos_exists: Optional[np.ndarray] = None
    acct_part: Optional[np.ndarray] = None         # [A*Pn]
    acct_part_exists: Optional[np.ndarray] = None
    qos_usage: Optional[np.ndarray] = None         # [Q]

    def __post_init__(self):
        self.qos = np.ascontiguousarray(self.qos, QOS_DT).reshape(-1)
        self.acct_parent = np.ascontiguousarray(self.acct_parent, np.uint32)
        self.part_limits = np.ascontiguousarray(self.part_limits, PART_LIMIT_DT).reshape(-1)
        Q, A, U, UA, Pn = len(self.qos), len(self.acct_parent), self.num_users, self.num_user_accts, self.num_partitions
        shapes = {"user_part_limit": (np.uint32, UA * Pn), "acct_part_limit": (np.uint32, A * Pn),
                  "user_qos": (USAGE_DT, U * Q), "user_qos_exists": (np.uint8, U * Q),
                  "user_part": (USAGE_DT, UA * Pn), "user_part_exists": (np.uint8, UA * Pn),
                  "acct_qos": (USAGE_DT, A * Q), "acct_qos_exists": (np.uint8, A * Q),
                  "acct_part": (USAGE_DT, A * Pn), "acct_part_exists": (np.uint8, A * Pn), "qos_usage": (USAGE_DT, Q)}
        for f, (dt, n) in shapes.items():
            v = getattr(self, f)
            if v is not None:
                v = np.ascontiguousarray(v, dt).reshape(-1)
                if len(v) != n:
                    raise ValueError(f"{f}: expected {n} records, got {len(v)}")
                setattr(self, f, v)

    @property
    def num_qos(self):
        return len(self.qos)

    @property
    def num_accounts(self):
        return len(self.acct_parent)

    def to_c(self) -> CnsLimitTables:
        p = lambda a: None if a is None or len(a) == 0 else a.ctypes.data
        return CnsLimitTables(self.num_users, self.num_user_accts, self.num_accounts, self.num_qos, self.num_partitions,
                              len(self.part_limits), p(self.qos), p(self.acct_parent), p(self.part_limits),
                              p(self.user_part_limit), p(self.acct_part_limit), p(self.user_qos), p(self.user_qos_exists),
                              p(self.user_part), p(self.user_part_exists), p(self.acct_qos), p(self.acct_qos_exists),
                              p(self.acct_part), p(self.acct_part_exists), p(self.qos_usage))

    def empty_usage(self) -> "Usage":
        Q, A, U, UA, Pn = self.num_qos, self.num_accounts, self.num_users, self.num_user_accts, self.num_partitions
        return Usage(np.zeros(U * Q, USAGE_DT), np.zeros(U * Q, np.uint8), np.zeros(UA * Pn, USAGE_DT),
                     np.zeros(UA * Pn, np.uint8), np.zeros(A * Q, USAGE_DT), np.zeros(A * Q, np.uint8),
                     np.zeros(A * Pn, USAGE_DT), np.zeros(A * Pn, np.uint8), np.zeros(Q, USAGE_DT))


@dataclass
class Usage:
    """Usage tables after an admission pass (cns_get_usage / the oracle's export)."""
    user_qos: np.ndarray
    user_qos_exists: np.ndarray
    user_part: np.ndarray
    user_part_exists: np.ndarray
    acct_qos: np.ndarray
    acct_qos_exists: np.ndarray
    acct_part: np.ndarray
    acct_part_exists: np.ndarray
    qos_usage: np.ndarray

    def pointers(self):
        return [a.ctypes.data_as(_P) for a in (self.user_qos, self.user_qos_exists, self.user_part,
                                               self.user_part_exists, self.acct_qos, self.acct_qos_exists,
                                               self.acct_part, self.acct_part_exists, self.qos_usage)]

    def same_as(self, o: "Usage") -> bool:
        return all(np.array_equal(getattr(self, f), getattr(o, f)) for f in self.__dataclass_fields__)


class LimitJobs:
    def __init__(self, user, user_acct, account, qos, partition, time_limit_sec, select_index=None, skip=None):
        a = lambda x, dt: np.ascontiguousarray(np.asarray(x, dtype=dt))
        self.user, self.user_acct, self.account = a(user, np.uint32), a(user_acct, np.uint32), a(account, np.uint32)
        self.qos, self.partition = a(qos, np.uint32), a(partition, np.uint32)
        self.time_limit_sec = a(time_limit_sec, np.int64)
        self.select_index = None if select_index is None else a(select_index, np.uint64)
        self.skip = None if skip is None else a(skip, np.uint8)
        self.num_jobs = len(self.user)
        for f in ("user_acct", "account", "qos", "partition", "time_limit_sec"):
            assert len(getattr(self, f)) == self.num_jobs, f

    def to_c(self) -> CnsLimitJobSoa:
        p = lambda x: None if x is None else x.ctypes.data
        return CnsLimitJobSoa(self.num_jobs, p(self.select_index), p(self.user), p(self.user_acct), p(self.account),
                              p(self.qos), p(self.partition), p(self.time_limit_sec), p(self.skip))
