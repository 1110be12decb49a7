"""ctypes mirror of include/crane_gpu/node_select.h (the C ABI of the engine).

Host-side containers (numpy SoA) for the three tables that cross the boundary of
CraneCtld's `SchedulerAlgo::NodeSelect` (reference: src/CraneCtld/JobScheduler.h:260-263):
the node snapshot (CranedMeta, src/CraneCtld/Node/NodeDefs.h:59-81), the running jobs
(RnJobInScheduler, JobScheduler.h:57-90) and the pending jobs (PdJobInScheduler,
JobScheduler.h:92-170), plus the placement result.  Pure plumbing: no scheduling logic here.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

CNS_ABI_VERSION = 4
RESV_NONE = 0xFFFFFFFF
MAX_GRES_CLASSES = 8
MAX_GRES_NAMES = 4
MAX_NODE_TYPES = 64
NODE_NONE = 0xFFFFFFFF

REASON_NONE, REASON_PRIORITY, REASON_RESOURCE, REASON_RESOURCE_RESERVED, \
    REASON_PARTITION_NOT_FOUND, REASON_SKIPPED, REASON_RESERVATION_NOT_FOUND = range(7)
REASON_ENGINE_REFUSED = 8   # (7: "Preempted", include/crane_gpu/preempt.h)
# cns_partition_status (cns_get_partition_status): why a partition's jobs came back with REASON_ENGINE_REFUSED
PART_SERVED, PART_REFUSED_NODE, PART_REFUSED_CPU, PART_REFUSED_TYPES, PART_REFUSED_WIDTH = range(5)
# cns_reason <-> the reference's reason strings (JobScheduler.cpp:6750-6831, JobScheduler.h:198)
REASON_STR = {
    REASON_NONE: "", REASON_PRIORITY: "Priority", REASON_RESOURCE: "Resource",
    REASON_RESOURCE_RESERVED: "Resource Reserved",
    REASON_PARTITION_NOT_FOUND: "Partition Not Found", REASON_SKIPPED: "<caller-set>",
    REASON_RESERVATION_NOT_FOUND: "Reservation Not Found",
    REASON_ENGINE_REFUSED: "<engine refused the partition: the caller's CPU scheduler takes the job>",
}

STATUS_STR = {0: "CNS_OK", -1: "CNS_ERR_INVALID_ARG", -2: "CNS_ERR_NO_DEVICE", -3: "CNS_ERR_HIP",
              -4: "CNS_ERR_UNSUPPORTED", -5: "CNS_ERR_STATE", -6: "CNS_ERR_DEVICE_FAULT"}


class CnsConfig(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("device", C.c_int32),
                ("scheduled_batch_size", C.c_uint64), ("max_job_num_per_node", C.c_uint32),
                ("kernel_pin", C.c_uint32), ("max_time_window_sec", C.c_int64)]


class CnsGresLayout(C.Structure):
    _fields_ = [("num_classes", C.c_uint32), ("class_name", C.c_uint8 * MAX_GRES_CLASSES),
                ("class_shift", C.c_uint8 * MAX_GRES_CLASSES),
                ("class_width", C.c_uint8 * MAX_GRES_CLASSES)]


_P = C.c_void_p


class CnsNodeSoa(C.Structure):
    _fields_ = [("num_nodes", C.c_uint32), ("num_partitions", C.c_uint32),
                ("cpu_total_raw", _P), ("mem_total", _P), ("core_lo", _P), ("core_hi", _P),
                ("gres_slots", _P), ("schedulable", _P), ("part_offsets", _P), ("part_nodes", _P),
                ("gres", CnsGresLayout), ("core_w2", _P), ("core_w3", _P), ("unsupported", _P)]


class CnsRunningSoa(C.Structure):
    _fields_ = [("num_jobs", C.c_uint32), ("num_allocs", C.c_uint32), ("end_sec", _P),
                ("alloc_offsets", _P), ("alloc_node", _P), ("alloc_cpu_raw", _P), ("alloc_mem", _P),
                ("alloc_core_lo", _P), ("alloc_core_hi", _P), ("alloc_gres", _P), ("reservation", _P),
                ("alloc_core_w2", _P), ("alloc_core_w3", _P)]


class CnsResvSoa(C.Structure):
    _fields_ = [("num_resv", C.c_uint32), ("num_allocs", C.c_uint32), ("start_sec", _P), ("end_sec", _P),
                ("alloc_offsets", _P), ("alloc_node", _P), ("alloc_cpu_raw", _P), ("alloc_mem", _P),
                ("alloc_core_lo", _P), ("alloc_core_hi", _P), ("alloc_gres", _P), ("alloc_core_w2", _P), ("alloc_core_w3", _P)]


class CnsJobSoa(C.Structure):
    _fields_ = [("num_jobs", C.c_uint64), ("partition", _P), ("time_limit_sec", _P),
                ("node_cpu_raw", _P), ("node_mem", _P), ("task_cpu_raw", _P), ("task_mem", _P),
                ("node_num", _P), ("ntasks", _P), ("ntasks_per_node_min", _P),
                ("ntasks_per_node_max", _P), ("exclusive", _P), ("gres_total", _P),
                ("gres_spec", _P), ("incl_offsets", _P), ("incl_nodes", _P), ("excl_offsets", _P),
                ("excl_nodes", _P), ("skip", _P), ("reservation", _P)]


class CnsPlacementSoa(C.Structure):
    _fields_ = [("place_capacity", C.c_uint64), ("start_sec", _P), ("reason", _P),
                ("place_offsets", _P), ("node_idx", _P), ("ntasks", _P), ("cpu_raw", _P),
                ("mem", _P), ("core_lo", _P), ("core_hi", _P), ("gres", _P), ("core_w2", _P), ("core_w3", _P)]


class CnsTiming(C.Structure):
    _fields_ = [("h2d_ms", C.c_double), ("init_ms", C.c_double), ("select_ms", C.c_double),
                ("d2h_ms", C.c_double), ("jobs_ordered", C.c_uint64),
                ("algorithmic_bytes", C.c_uint64)]


class CnsResultsOffsets(C.Structure):   # cns_results_offsets
    _fields_ = [("num_jobs", C.c_uint64), ("num_places", C.c_uint64), ("wide_cores", C.c_uint32), ("reserved0", C.c_uint32)] + \
               [(f, C.c_uint64) for f in ("start_sec", "cpu_raw", "mem", "core_lo", "core_hi", "gres", "node_idx", "ntasks", "reason",
                                          "core_w2", "core_w3", "total_bytes")]


class CnsGroupInfo(C.Structure):        # cns_group_info
    _fields_ = [("num_devices", C.c_uint32), ("gather_mode", C.c_uint32), ("shards_ms", C.c_double), ("max_select_ms", C.c_double),
                ("allgather_ms", C.c_double), ("download_ms", C.c_double), ("scatter_ms", C.c_double), ("slot_bytes", C.c_uint64)]


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_P)


def _arr(x, dtype, n=None):
    a = np.ascontiguousarray(x, dtype=dtype)
    if n is not None and a.shape[0] != n:
        raise ValueError(f"expected leading dim {n}, got {a.shape}")
    return a


@dataclass
class GresLayout:
    """(name,type) classes of the 64-bit GRES slot mask."""
    class_name: list = field(default_factory=list)   # name-group id per class
    class_shift: list = field(default_factory=list)
    class_width: list = field(default_factory=list)

    def to_c(self) -> CnsGresLayout:
        g = CnsGresLayout()
        g.num_classes = len(self.class_name)
        for i, (a, s, w) in enumerate(zip(self.class_name, self.class_shift, self.class_width)):
            g.class_name[i], g.class_shift[i], g.class_width[i] = a, s, w
        return g

    def class_mask(self, g: int) -> int:
        return ((1 << self.class_width[g]) - 1) << self.class_shift[g]


@dataclass
class Cluster:
    """Per-cycle node snapshot (cns_node_soa)."""
    cpu_total_raw: np.ndarray
    mem_total: np.ndarray
    core_lo: np.ndarray
    core_hi: np.ndarray
    gres_slots: np.ndarray
    part_offsets: np.ndarray
    part_nodes: np.ndarray
    gres: GresLayout = field(default_factory=GresLayout)
    schedulable: Optional[np.ndarray] = None
    core_w2: Optional[np.ndarray] = None    # core ids 128..191 / 192..255 (ABI 3); None = no node has any
    core_w3: Optional[np.ndarray] = None
    unsupported: Optional[np.ndarray] = None   # [N] uint8 (ABI 4): nodes the caller could not express (core id >= 256, too many GRES slots); None = none

    def __post_init__(self):
        n = len(self.cpu_total_raw)
        if self.unsupported is not None:
            self.unsupported = _arr(self.unsupported, np.uint8, n)
        if self.core_w2 is not None:
            self.core_w2 = _arr(self.core_w2, np.uint64, n)
        if self.core_w3 is not None:
            self.core_w3 = _arr(self.core_w3, np.uint64, n)
        self.cpu_total_raw = _arr(self.cpu_total_raw, np.int64)
        self.mem_total = _arr(self.mem_total, np.uint64, n)
        self.core_lo = _arr(self.core_lo, np.uint64, n)
        self.core_hi = _arr(self.core_hi, np.uint64, n)
        self.gres_slots = _arr(self.gres_slots, np.uint64, n)
        self.part_offsets = _arr(self.part_offsets, np.uint32)
        self.part_nodes = _arr(self.part_nodes, np.uint32)
        if self.schedulable is not None:
            self.schedulable = _arr(self.schedulable, np.uint8, n)

    @property
    def num_nodes(self):
        return len(self.cpu_total_raw)

    @property
    def num_partitions(self):
        return len(self.part_offsets) - 1

    @property
    def wide_cores(self) -> bool:
        """A node has a core id above 127: results carry the core_w2 / core_w3 planes (cns_placement_soa, packed buffer)."""
        return bool((self.core_w2 is not None and self.core_w2.any()) or (self.core_w3 is not None and self.core_w3.any()))

    def to_c(self) -> CnsNodeSoa:
        s = CnsNodeSoa()
        s.num_nodes, s.num_partitions = self.num_nodes, self.num_partitions
        s.cpu_total_raw, s.mem_total = _ptr(self.cpu_total_raw), _ptr(self.mem_total)
        s.core_lo, s.core_hi, s.gres_slots = _ptr(self.core_lo), _ptr(self.core_hi), _ptr(self.gres_slots)
        s.schedulable = _ptr(self.schedulable)
        s.part_offsets, s.part_nodes = _ptr(self.part_offsets), _ptr(self.part_nodes)
        s.gres = self.gres.to_c()
        s.core_w2, s.core_w3 = _ptr(self.core_w2), _ptr(self.core_w3)
        s.unsupported = _ptr(self.unsupported)
        return s


@dataclass
class Running:
    """Running jobs' allocations (cns_running_soa)."""
    end_sec: np.ndarray
    alloc_offsets: np.ndarray
    alloc_node: np.ndarray
    alloc_cpu_raw: np.ndarray
    alloc_mem: np.ndarray
    alloc_core_lo: np.ndarray
    alloc_core_hi: np.ndarray
    alloc_gres: np.ndarray
    reservation: Optional[np.ndarray] = None   # [R] reservation index or RESV_NONE
    alloc_core_w2: Optional[np.ndarray] = None
    alloc_core_w3: Optional[np.ndarray] = None

    def __post_init__(self):
        self.end_sec = _arr(self.end_sec, np.int64)
        if self.reservation is not None:
            self.reservation = _arr(self.reservation, np.uint32, len(self.end_sec))
        self.alloc_offsets = _arr(self.alloc_offsets, np.uint32, len(self.end_sec) + 1)
        m = int(self.alloc_offsets[-1]) if len(self.alloc_offsets) else 0
        self.alloc_node = _arr(self.alloc_node, np.uint32, m)
        self.alloc_cpu_raw = _arr(self.alloc_cpu_raw, np.int64, m)
        self.alloc_mem = _arr(self.alloc_mem, np.uint64, m)
        self.alloc_core_lo = _arr(self.alloc_core_lo, np.uint64, m)
        self.alloc_core_hi = _arr(self.alloc_core_hi, np.uint64, m)
        self.alloc_gres = _arr(self.alloc_gres, np.uint64, m)
        if self.alloc_core_w2 is not None:
            self.alloc_core_w2 = _arr(self.alloc_core_w2, np.uint64, m)
        if self.alloc_core_w3 is not None:
            self.alloc_core_w3 = _arr(self.alloc_core_w3, np.uint64, m)

    def to_c(self) -> CnsRunningSoa:
        s = CnsRunningSoa()
        s.num_jobs, s.num_allocs = len(self.end_sec), len(self.alloc_node)
        s.end_sec, s.alloc_offsets, s.alloc_node = _ptr(self.end_sec), _ptr(self.alloc_offsets), _ptr(self.alloc_node)
        s.alloc_cpu_raw, s.alloc_mem = _ptr(self.alloc_cpu_raw), _ptr(self.alloc_mem)
        s.alloc_core_lo, s.alloc_core_hi, s.alloc_gres = _ptr(self.alloc_core_lo), _ptr(self.alloc_core_hi), _ptr(self.alloc_gres)
        s.reservation = _ptr(self.reservation)
        s.alloc_core_w2, s.alloc_core_w3 = _ptr(self.alloc_core_w2), _ptr(self.alloc_core_w3)
        return s


@dataclass
class Reservations:
    """Reservations of the cycle (cns_resv_soa): start / end and the reserved resources per node."""
    start_sec: np.ndarray
    end_sec: np.ndarray
    alloc_offsets: np.ndarray
    alloc_node: np.ndarray
    alloc_cpu_raw: np.ndarray
    alloc_mem: np.ndarray
    alloc_core_lo: np.ndarray
    alloc_core_hi: np.ndarray
    alloc_gres: np.ndarray
    alloc_core_w2: Optional[np.ndarray] = None
    alloc_core_w3: Optional[np.ndarray] = None

    def __post_init__(self):
        self.start_sec = _arr(self.start_sec, np.int64)
        v = len(self.start_sec)
        self.end_sec = _arr(self.end_sec, np.int64, v)
        self.alloc_offsets = _arr(self.alloc_offsets, np.uint32, v + 1)
        m = int(self.alloc_offsets[-1]) if len(self.alloc_offsets) else 0
        self.alloc_node = _arr(self.alloc_node, np.uint32, m)
        self.alloc_cpu_raw = _arr(self.alloc_cpu_raw, np.int64, m)
        self.alloc_mem = _arr(self.alloc_mem, np.uint64, m)
        self.alloc_core_lo = _arr(self.alloc_core_lo, np.uint64, m)
        self.alloc_core_hi = _arr(self.alloc_core_hi, np.uint64, m)
        self.alloc_gres = _arr(self.alloc_gres, np.uint64, m)
        if self.alloc_core_w2 is not None:
            self.alloc_core_w2 = _arr(self.alloc_core_w2, np.uint64, m)
        if self.alloc_core_w3 is not None:
            self.alloc_core_w3 = _arr(self.alloc_core_w3, np.uint64, m)

    @property
    def num_resv(self):
        return len(self.start_sec)

    def to_c(self) -> CnsResvSoa:
        s = CnsResvSoa()
        s.num_resv, s.num_allocs = len(self.start_sec), len(self.alloc_node)
        s.start_sec, s.end_sec, s.alloc_offsets = _ptr(self.start_sec), _ptr(self.end_sec), _ptr(self.alloc_offsets)
        s.alloc_node, s.alloc_cpu_raw, s.alloc_mem = _ptr(self.alloc_node), _ptr(self.alloc_cpu_raw), _ptr(self.alloc_mem)
        s.alloc_core_lo, s.alloc_core_hi, s.alloc_gres = _ptr(self.alloc_core_lo), _ptr(self.alloc_core_hi), _ptr(self.alloc_gres)
        s.alloc_core_w2, s.alloc_core_w3 = _ptr(self.alloc_core_w2), _ptr(self.alloc_core_w3)
        return s


@dataclass
class Jobs:
    """Pending jobs in priority order (cns_job_soa)."""
    partition: np.ndarray
    time_limit_sec: np.ndarray
    node_mem: np.ndarray
    task_cpu_raw: np.ndarray
    task_mem: np.ndarray
    node_num: np.ndarray
    ntasks: np.ndarray
    ntasks_per_node_min: np.ndarray
    ntasks_per_node_max: np.ndarray
    node_cpu_raw: Optional[np.ndarray] = None
    exclusive: Optional[np.ndarray] = None
    gres_total: Optional[np.ndarray] = None   # [J, 4] uint8
    gres_spec: Optional[np.ndarray] = None    # [J, 8] uint8
    incl_offsets: Optional[np.ndarray] = None
    incl_nodes: Optional[np.ndarray] = None
    excl_offsets: Optional[np.ndarray] = None
    excl_nodes: Optional[np.ndarray] = None
    skip: Optional[np.ndarray] = None
    reservation: Optional[np.ndarray] = None   # [J] reservation index or RESV_NONE

    def __post_init__(self):
        j = len(self.partition)
        if self.reservation is not None:
            self.reservation = _arr(self.reservation, np.uint32, j)
        self.partition = _arr(self.partition, np.uint32)
        self.time_limit_sec = _arr(self.time_limit_sec, np.int64, j)
        self.node_mem = _arr(self.node_mem, np.uint64, j)
        self.task_cpu_raw = _arr(self.task_cpu_raw, np.int64, j)
        self.task_mem = _arr(self.task_mem, np.uint64, j)
        self.node_num = _arr(self.node_num, np.uint32, j)
        self.ntasks = _arr(self.ntasks, np.uint32, j)
        self.ntasks_per_node_min = _arr(self.ntasks_per_node_min, np.uint32, j)
        self.ntasks_per_node_max = _arr(self.ntasks_per_node_max, np.uint32, j)
        for name, dt in (("node_cpu_raw", np.int64), ("exclusive", np.uint8), ("skip", np.uint8)):
            v = getattr(self, name)
            if v is not None:
                setattr(self, name, _arr(v, dt, j))
        if self.gres_total is not None:
            self.gres_total = _arr(self.gres_total, np.uint8, j).reshape(j, MAX_GRES_NAMES)
        if self.gres_spec is not None:
            self.gres_spec = _arr(self.gres_spec, np.uint8, j).reshape(j, MAX_GRES_CLASSES)
        for off, lst in (("incl_offsets", "incl_nodes"), ("excl_offsets", "excl_nodes")):
            if getattr(self, off) is not None:
                setattr(self, off, _arr(getattr(self, off), np.uint64, j + 1))
                setattr(self, lst, _arr(getattr(self, lst), np.uint32))

    @property
    def num_jobs(self):
        return len(self.partition)

    def total_places(self) -> int:
        return int(self.node_num.astype(np.uint64).sum())

    def to_c(self) -> CnsJobSoa:
        s = CnsJobSoa()
        s.num_jobs = self.num_jobs
        for f, _ in CnsJobSoa._fields_[1:]:
            setattr(s, f, _ptr(getattr(self, f)))
        return s


class Placements:
    """Caller-allocated result arrays (cns_placement_soa)."""

    def __init__(self, num_jobs: int, capacity: int, alloc=None):
        """alloc(n, dtype, fill) -> array: where the arrays live (default: numpy; GpuNodeSelector.pinned: page-locked memory)."""
        self.num_jobs, self.capacity = num_jobs, capacity
        cap = max(capacity, 1)
        if alloc is None:
            alloc = lambda n, dt, fill=0: np.full(n, fill, dt) if fill else np.zeros(n, dt)
        self.start_sec = alloc(max(num_jobs, 1), np.int64, 0)
        self.reason = alloc(max(num_jobs, 1), np.uint8, 0)
        self.place_offsets = alloc(num_jobs + 1, np.uint64, 0)
        self.node_idx = alloc(cap, np.uint32, NODE_NONE)
        self.ntasks = alloc(cap, np.uint32, 0)
        self.cpu_raw = alloc(cap, np.int64, 0)
        self.mem = alloc(cap, np.uint64, 0)
        self.core_lo = alloc(cap, np.uint64, 0)
        self.core_hi = alloc(cap, np.uint64, 0)
        self.gres = alloc(cap, np.uint64, 0)
        self.core_w2 = alloc(cap, np.uint64, 0)
        self.core_w3 = alloc(cap, np.uint64, 0)

    def to_c(self) -> CnsPlacementSoa:
        s = CnsPlacementSoa()
        s.place_capacity = self.capacity
        for f, _ in CnsPlacementSoa._fields_[1:]:
            setattr(s, f, _ptr(getattr(self, f)))
        return s

    FIELDS = ("start_sec", "reason", "place_offsets", "node_idx", "ntasks", "cpu_raw", "mem",
              "core_lo", "core_hi", "gres", "core_w2", "core_w3")

    def trimmed(self):
        """Dict of result arrays cut to their logical lengths (for comparisons / hashing)."""
        j, c = self.num_jobs, self.capacity
        return {"start_sec": self.start_sec[:j], "reason": self.reason[:j],
                "place_offsets": self.place_offsets[:j + 1], "node_idx": self.node_idx[:c],
                "ntasks": self.ntasks[:c], "cpu_raw": self.cpu_raw[:c], "mem": self.mem[:c],
                "core_lo": self.core_lo[:c], "core_hi": self.core_hi[:c], "gres": self.gres[:c],
                "core_w2": self.core_w2[:c], "core_w3": self.core_w3[:c]}

    def diff(self, other: "Placements"):
        """First differing (field, index) between two results, or None if bit-identical."""
        a, b = self.trimmed(), other.trimmed()
        for k in self.FIELDS:
            if a[k].shape != b[k].shape:
                return (k, "shape", a[k].shape, b[k].shape)
            ne = np.nonzero(a[k] != b[k])[0]
            if len(ne):
                i = int(ne[0])
                return (k, i, a[k][i].item(), b[k][i].item(), f"{len(ne)} mismatches")
        return None


# ---------------------------------------------------------------------------------------------------------
# include/crane_gpu/preempt.h
# ---------------------------------------------------------------------------------------------------------
REASON_PREEMPTED = 7
PREEMPT_REF_PENDING = 0x80000000


class CnsPreemptSoa(C.Structure):
    _fields_ = [("enabled", C.c_uint32), ("num_qos", C.c_uint32), ("qos_preempt_offsets", C.c_void_p),
                ("qos_preempt", C.c_void_p), ("pd_job_id", C.c_void_p), ("pd_qos", C.c_void_p),
                ("pd_qos_priority", C.c_void_p), ("pd_priority", C.c_void_p), ("rn_job_id", C.c_void_p),
                ("rn_qos", C.c_void_p), ("rn_qos_priority", C.c_void_p), ("rn_start_sec", C.c_void_p),
                ("num_preempting", C.c_uint32), ("reserved0", C.c_uint32), ("preempting_job_ids", C.c_void_p)]


class CnsPreemptOut(C.Structure):
    _fields_ = [("capacity", C.c_uint64), ("offsets", C.c_void_p), ("preempted", C.c_void_p),
                ("cancel_capacity", C.c_uint32), ("num_cancelled", C.c_uint32), ("cancelled_job_ids", C.c_void_p),
                ("preempting_capacity", C.c_uint32), ("num_preempting", C.c_uint32), ("preempting_job_ids", C.c_void_p)]


@dataclass
class Preempt:
    """Preemption inputs of one cycle (cns_preempt_soa): QoS preempt lists, the fields TryPreempt_ reads of the pending
    and running jobs, and m_preempting_set_ as the previous cycle left it."""
    qos_preempt: list                      # qos id -> list of qos ids it may preempt
    pd_job_id: np.ndarray
    pd_qos: np.ndarray
    pd_qos_priority: np.ndarray
    pd_priority: np.ndarray
    rn_job_id: np.ndarray
    rn_qos: np.ndarray
    rn_qos_priority: np.ndarray
    rn_start_sec: np.ndarray
    preempting: np.ndarray = None
    enabled: bool = True

    def __post_init__(self):
        self.pd_job_id = _arr(self.pd_job_id, np.uint32); n = len(self.pd_job_id)
        self.pd_qos = _arr(self.pd_qos, np.uint32, n)
        self.pd_qos_priority = _arr(self.pd_qos_priority, np.uint32, n)
        self.pd_priority = _arr(self.pd_priority, np.float64, n)
        self.rn_job_id = _arr(self.rn_job_id, np.uint32); r = len(self.rn_job_id)
        self.rn_qos = _arr(self.rn_qos, np.uint32, r)
        self.rn_qos_priority = _arr(self.rn_qos_priority, np.uint32, r)
        self.rn_start_sec = _arr(self.rn_start_sec, np.int64, r)
        self.preempting = _arr(self.preempting if self.preempting is not None else [], np.uint32)
        off = [0]
        flat = []
        for lst in self.qos_preempt:
            flat.extend(int(x) for x in lst)
            off.append(len(flat))
        self._off = np.asarray(off, np.uint32)
        self._flat = np.asarray(flat if flat else [0], np.uint32)

    def to_c(self) -> CnsPreemptSoa:
        s = CnsPreemptSoa()
        s.enabled, s.num_qos = int(self.enabled), len(self.qos_preempt)
        s.qos_preempt_offsets, s.qos_preempt = _ptr(self._off), _ptr(self._flat)
        for f in ("pd_job_id", "pd_qos", "pd_qos_priority", "pd_priority", "rn_job_id", "rn_qos", "rn_qos_priority", "rn_start_sec"):
            setattr(s, f, _ptr(getattr(self, f)))
        s.num_preempting, s.preempting_job_ids = len(self.preempting), _ptr(self.preempting if len(self.preempting) else np.zeros(1, np.uint32))
        return s


class PreemptOut:
    """Caller-allocated preemption results (cns_preempt_out)."""

    def __init__(self, num_jobs: int, num_running: int, capacity: int = 0):
        self.num_jobs = num_jobs
        cap = max(capacity or (num_jobs + num_running) * 4, 1)
        self.capacity = cap
        self.offsets = np.zeros(num_jobs + 1, np.uint64)
        self.preempted = np.zeros(cap, np.uint32)
        self.cancelled = np.zeros(max(num_running, 1), np.uint32)
        self.preempting = np.zeros(max(2 * num_running, 1), np.uint32)
        self._c = None

    def to_c(self) -> CnsPreemptOut:
        s = CnsPreemptOut()
        s.capacity, s.offsets, s.preempted = self.capacity, _ptr(self.offsets), _ptr(self.preempted)
        s.cancel_capacity, s.cancelled_job_ids = len(self.cancelled), _ptr(self.cancelled)
        s.preempting_capacity, s.preempting_job_ids = len(self.preempting), _ptr(self.preempting)
        self._c = s
        return s

    def lists(self):
        """[(is_pending, index), ...] per pending job."""
        out = []
        for j in range(self.num_jobs):
            refs = self.preempted[int(self.offsets[j]):int(self.offsets[j + 1])]
            out.append([(bool(r & PREEMPT_REF_PENDING), int(r & 0x7FFFFFFF)) for r in refs])
        return out

    def cancelled_ids(self):
        return [int(x) for x in self.cancelled[:self._c.num_cancelled]]

    def preempting_ids(self):
        return [int(x) for x in self.preempting[:self._c.num_preempting]]
