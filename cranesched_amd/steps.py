"""Step-scheduler inputs / outputs (include/crane_gpu/steps.h) as numpy SoA + ctypes views.

Mirror of what `JobInCtld::SchedulePendingSteps` reads and writes (src/CraneCtld/CtldPublicDefs.cpp:2038-2159):
`StepJobs` = the jobs with pending steps and what is free inside their allocations (`step_res_avail_`), `Steps` = the
pending steps grouped by job in queue order, `StepResults` = craned ids / task maps / allocations + the availability
left.  Pure plumbing: no scheduling logic here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import abi

STEP_MAX_NODES = 64
_P = C.c_void_p


class CnsStepJobSoa(C.Structure):
    _fields_ = [("num_jobs", C.c_uint32), ("num_nodes", C.c_uint32), ("node_offsets", _P), ("node_idx", _P),
                ("avail_cpu_raw", _P), ("avail_mem", _P), ("avail_core_lo", _P), ("avail_core_hi", _P),
                ("avail_gres", _P), ("step_offsets", _P), ("avail_core_w2", _P), ("avail_core_w3", _P)]


class CnsStepSoa(C.Structure):
    _fields_ = [("num_steps", C.c_uint32), ("node_cpu_raw", _P), ("node_mem", _P), ("node_gres_total", _P),
                ("node_gres_spec", _P), ("task_cpu_raw", _P), ("task_mem", _P), ("task_gres_total", _P),
                ("task_gres_spec", _P), ("node_num", _P), ("ntasks", _P), ("ntasks_per_node_min", _P),
                ("ntasks_per_node_max", _P), ("incl_offsets", _P), ("incl_nodes", _P), ("excl_offsets", _P),
                ("excl_nodes", _P)]


class CnsStepResultSoa(C.Structure):
    _fields_ = [("scheduled", _P), ("place_offsets", _P), ("node_idx", _P), ("node_ntasks", _P), ("node_cpu_raw", _P),
                ("node_mem", _P), ("node_core_lo", _P), ("node_core_hi", _P), ("node_gres", _P), ("task_offsets", _P),
                ("task_node", _P), ("task_cpu_raw", _P), ("task_mem", _P), ("task_core_lo", _P), ("task_core_hi", _P),
                ("task_gres", _P), ("avail_cpu_raw", _P), ("avail_mem", _P), ("avail_core_lo", _P),
                ("avail_core_hi", _P), ("avail_gres", _P), ("node_core_w2", _P), ("node_core_w3", _P), ("task_core_w2", _P),
                ("task_core_w3", _P), ("avail_core_w2", _P), ("avail_core_w3", _P)]


def _a(x, dt):
    return np.ascontiguousarray(np.asarray(x, dtype=dt))


def _p(a):
    return None if a is None else a.ctypes.data


class StepJobs:
    def __init__(self, node_offsets, node_idx, avail_cpu_raw, avail_mem, avail_core_lo, avail_core_hi, avail_gres,
                 step_offsets, avail_core_w2=None, avail_core_w3=None):
        self.node_offsets, self.node_idx = _a(node_offsets, np.uint32), _a(node_idx, np.uint32)
        self.avail_cpu_raw, self.avail_mem = _a(avail_cpu_raw, np.int64), _a(avail_mem, np.uint64)
        self.avail_core_lo, self.avail_core_hi = _a(avail_core_lo, np.uint64), _a(avail_core_hi, np.uint64)
        self.avail_gres, self.step_offsets = _a(avail_gres, np.uint64), _a(step_offsets, np.uint32)
        self.avail_core_w2 = None if avail_core_w2 is None else _a(avail_core_w2, np.uint64)   # core ids 128..255 (ABI 3)
        self.avail_core_w3 = None if avail_core_w3 is None else _a(avail_core_w3, np.uint64)
        self.num_jobs, self.num_nodes = len(self.node_offsets) - 1, len(self.node_idx)
        assert self.node_offsets[-1] == self.num_nodes and len(self.step_offsets) == self.num_jobs + 1

    def to_c(self) -> CnsStepJobSoa:
        return CnsStepJobSoa(self.num_jobs, self.num_nodes, _p(self.node_offsets), _p(self.node_idx),
                             _p(self.avail_cpu_raw), _p(self.avail_mem), _p(self.avail_core_lo), _p(self.avail_core_hi),
                             _p(self.avail_gres), _p(self.step_offsets), _p(self.avail_core_w2), _p(self.avail_core_w3))


class Steps:
    def __init__(self, node_cpu_raw, node_mem, task_cpu_raw, task_mem, node_num, ntasks, tmin, tmax,
                 node_gres_total=None, node_gres_spec=None, task_gres_total=None, task_gres_spec=None,
                 incl_offsets=None, incl_nodes=None, excl_offsets=None, excl_nodes=None):
        self.node_cpu_raw, self.node_mem = _a(node_cpu_raw, np.int64), _a(node_mem, np.uint64)
        self.task_cpu_raw, self.task_mem = _a(task_cpu_raw, np.int64), _a(task_mem, np.uint64)
        self.node_num, self.ntasks = _a(node_num, np.uint32), _a(ntasks, np.uint32)
        self.tmin, self.tmax = _a(tmin, np.uint32), _a(tmax, np.uint32)
        S = self.num_steps = len(self.node_num)
        g = lambda x, w: None if x is None else _a(x, np.uint8).reshape(S, w)
        self.node_gres_total, self.node_gres_spec = g(node_gres_total, abi.MAX_GRES_NAMES), g(node_gres_spec, abi.MAX_GRES_CLASSES)
        self.task_gres_total, self.task_gres_spec = g(task_gres_total, abi.MAX_GRES_NAMES), g(task_gres_spec, abi.MAX_GRES_CLASSES)
        o = lambda x: None if x is None else _a(x, np.uint32)
        self.incl_offsets, self.incl_nodes, self.excl_offsets, self.excl_nodes = o(incl_offsets), o(incl_nodes), o(excl_offsets), o(excl_nodes)

    def to_c(self) -> CnsStepSoa:
        return CnsStepSoa(self.num_steps, _p(self.node_cpu_raw), _p(self.node_mem), _p(self.node_gres_total),
                          _p(self.node_gres_spec), _p(self.task_cpu_raw), _p(self.task_mem), _p(self.task_gres_total),
                          _p(self.task_gres_spec), _p(self.node_num), _p(self.ntasks), _p(self.tmin), _p(self.tmax),
                          _p(self.incl_offsets), _p(self.incl_nodes), _p(self.excl_offsets), _p(self.excl_nodes))


class StepResults:
    FIELDS = ("scheduled", "place_offsets", "node_idx", "node_ntasks", "node_cpu_raw", "node_mem", "node_core_lo",
              "node_core_hi", "node_gres", "task_offsets", "task_node", "task_cpu_raw", "task_mem", "task_core_lo",
              "task_core_hi", "task_gres", "avail_cpu_raw", "avail_mem", "avail_core_lo", "avail_core_hi", "avail_gres",
              "node_core_w2", "node_core_w3", "task_core_w2", "task_core_w3", "avail_core_w2", "avail_core_w3")

    def __init__(self, jobs: StepJobs, steps: Steps):
        S, places, tasks, n = steps.num_steps, int(steps.node_num.sum()), int(steps.ntasks.sum()), jobs.num_nodes
        z = lambda k, dt: np.zeros(max(k, 1), dt)
        self.scheduled = z(S, np.uint8)
        self.place_offsets, self.task_offsets = np.zeros(S + 1, np.uint64), np.zeros(S + 1, np.uint64)
        self.node_idx, self.node_ntasks = np.full(max(places, 1), abi.NODE_NONE, np.uint32), z(places, np.uint32)
        self.node_cpu_raw, self.node_mem = z(places, np.int64), z(places, np.uint64)
        self.node_core_lo, self.node_core_hi, self.node_gres = z(places, np.uint64), z(places, np.uint64), z(places, np.uint64)
        self.task_node = np.full(max(tasks, 1), abi.NODE_NONE, np.uint32)
        self.task_cpu_raw, self.task_mem = z(tasks, np.int64), z(tasks, np.uint64)
        self.task_core_lo, self.task_core_hi, self.task_gres = z(tasks, np.uint64), z(tasks, np.uint64), z(tasks, np.uint64)
        self.avail_cpu_raw, self.avail_mem = z(n, np.int64), z(n, np.uint64)
        self.avail_core_lo, self.avail_core_hi, self.avail_gres = z(n, np.uint64), z(n, np.uint64), z(n, np.uint64)
        self.node_core_w2, self.node_core_w3 = z(places, np.uint64), z(places, np.uint64)
        self.task_core_w2, self.task_core_w3 = z(tasks, np.uint64), z(tasks, np.uint64)
        self.avail_core_w2, self.avail_core_w3 = z(n, np.uint64), z(n, np.uint64)

    def to_c(self) -> CnsStepResultSoa:
        return CnsStepResultSoa(*[_p(getattr(self, f)) for f in self.FIELDS])

    def diff(self, other: "StepResults"):
        for f in self.FIELDS:
            a, b = getattr(self, f), getattr(other, f)
            if not np.array_equal(a, b):
                i = int(np.nonzero(a != b)[0][0])
                return (f, i, a[i], b[i])
        return None
