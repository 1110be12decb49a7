"""ctypes binding of the engine's C ABI (include/crane_gpu/node_select.h).

`GpuNodeSelector` is the Python mirror of the reference's `SchedulerAlgo` plugin surface
(src/CraneCtld/JobScheduler.h:233-263): one object per controller, `node_select(now, running,
pending)` once per scheduling cycle.  The work happens in the HIP shared library; this module
fails loudly if that library is missing or no MI355X is visible — there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
# torch first: its wheel bundles its own HIP/HSA runtime (SONAME libamdhip64.so.7).  Loading it before
# the engine makes the dynamic loader hand the SAME runtime to libcrane_gpu_nodeselect.so; two HIP
# runtimes in one process leave the second one without devices (hipGetDeviceCount -> "no ROCm-capable
# device").  torch is plumbing here (device memory for RCCL, torch.distributed), never on the hot path.
import torch  # noqa: F401  (must precede the CDLL below)

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CNS_ENGINE_LIB") or os.path.join(_HERE, "libcrane_gpu_nodeselect.so")
_LIB = None

# every symbol include/crane_gpu/node_select.h declares
ABI_SYMBOLS = ("cns_abi_version", "cns_last_error", "cns_create", "cns_destroy", "cns_set_nodes",
               "cns_set_reservations", "cns_set_running", "cns_set_host_threads", "cns_select", "cns_upload_jobs", "cns_run_resident", "cns_download",
               "cns_device_results", "cns_host_alloc", "cns_host_free", "cns_get_timing", "cns_debug_get_costs", "cns_debug_get_timeline", "cns_debug_get_timeline_cores",
               "cns_debug_last_kernel", "cns_debug_get_prof", "cns_debug_engine_partitions",
               # several devices (csrc/group_host.inc)
               "cns_get_partition_status",
               "cns_results_layout", "cns_comm_unique_id", "cns_comm_init_rank", "cns_comm_destroy", "cns_allgather_results",
               "cns_download_gathered", "cns_gather_timing", "cns_group_create", "cns_group_destroy", "cns_group_last_error",
               "cns_group_size", "cns_group_handle", "cns_group_set_nodes", "cns_group_set_reservations", "cns_group_set_running",
               "cns_group_select", "cns_group_get_info", "cns_group_device_of_partition", "cns_group_get_partition_status")
# ... and include/crane_gpu/priority.h
PRIORITY_ABI_SYMBOLS = ("cns_priority_order", "cns_priority_timing")
# ... and include/crane_gpu/run_limits.h
# ... and include/crane_gpu/steps.h
STEPS_ABI_SYMBOLS = ("cns_schedule_steps",)
# ... and include/crane_gpu/preempt.h
PREEMPT_ABI_SYMBOLS = ("cns_select_preempt",)
LIMITS_ABI_SYMBOLS = ("cns_set_run_limits", "cns_apply_run_limits", "cns_upload_limit_jobs", "cns_run_limits_resident",
                      "cns_download_limits", "cns_get_limit_timing", "cns_get_usage")


class EngineError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"{abi.STATUS_STR.get(status, status)}: {msg}")
        self.status = status


def lib():
    """Load the HIP engine. Raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: the HIP engine is not built and there is no CPU fallback. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'`.")
        L = C.CDLL(LIB_PATH)
        L.cns_last_error.restype = C.c_char_p
        L.cns_last_error.argtypes = [C.c_void_p]
        L.cns_destroy.restype = None
        L.cns_destroy.argtypes = [C.c_void_p]
        L.cns_debug_last_kernel.restype = C.c_char_p
        L.cns_debug_last_kernel.argtypes = [C.c_void_p]
        if not hasattr(L, "cns_group_create"):   # (an older build of the engine, loaded through CNS_ENGINE_LIB for an A/B run: tools/var_bench.py)
            _LIB = L
            return _LIB
        L.cns_group_last_error.restype = C.c_char_p
        L.cns_group_last_error.argtypes = [C.c_void_p]
        L.cns_group_destroy.restype = None
        L.cns_group_destroy.argtypes = [C.c_void_p]
        L.cns_group_handle.restype = C.c_void_p
        L.cns_group_handle.argtypes = [C.c_void_p, C.c_uint32]
        L.cns_group_size.restype = C.c_uint32
        L.cns_group_size.argtypes = [C.c_void_p]
        L.cns_group_device_of_partition.restype = C.c_uint32
        L.cns_group_device_of_partition.argtypes = [C.c_void_p, C.c_uint32]
        _LIB = L
    return _LIB


class GpuNodeSelector:
    """One engine handle on one MI355X (one process per GPU)."""

    def __init__(self, device: int = 0, scheduled_batch_size: int = 0, max_job_num_per_node: int = 0,
                 max_time_window_sec: int = 0, kernel_pin: int = 0):
        """kernel_pin: cns_kernel_pin — 0 the engine chooses per launch; 1 k_select, 2 k_pipe (one workgroup per partition: for a controller
        that shares its GPU with other processes, since k_wide's workgroups must all be resident at once)."""
        self._L = lib()
        self._h = C.c_void_p()
        cfg = abi.CnsConfig(abi.CNS_ABI_VERSION, device, scheduled_batch_size, max_job_num_per_node, kernel_pin,
                            max_time_window_sec)
        rc = self._L.cns_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            msg = self._L.cns_last_error(None)
            self._h = C.c_void_p()
            raise EngineError(rc, msg.decode() if msg else "")
        self._cluster = None
        self._jobs = None
        self._pinned = {}   # address -> array over a cns_host_alloc buffer (cns_destroy releases the memory: do not touch them after close())

    # -- plumbing -----------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != 0:
            msg = self._L.cns_last_error(self._h)
            raise EngineError(rc, msg.decode() if msg else "")

    # -- MultiFactorPriority (include/crane_gpu/priority.h) ---------------------------------------
    def priority_order(self, now: int, cfg, num_accounts: int, pending, running=None, limit: int | None = None):
        """IPrioritySorter::GetOrderedJobPtrVec for PriorityType multifactor (JobScheduler.cpp:7606-7631).
        Returns (order, priority, num_ordered): order[i] = input index of the i-th job by descending priority
        (ties: ascending index), priority[j] per input job, num_ordered = min(J, limit)."""
        J = pending.num_jobs
        order = np.empty(max(J, 1), np.uint32)
        prio = np.empty(max(J, 1), np.float64)
        nord = C.c_uint64(0)
        c_cfg, c_pd = cfg.to_c(), pending.to_c()
        c_rn = running.to_c() if running is not None else None
        self._check(self._L.cns_priority_order(
            self._h, C.c_int64(now), C.byref(c_cfg), C.c_uint32(num_accounts), C.byref(c_pd),
            C.byref(c_rn) if c_rn is not None else None, C.c_uint64(J if limit is None else limit),
            order.ctypes.data_as(C.c_void_p), prio.ctypes.data_as(C.c_void_p), C.byref(nord)))
        return order[:J], prio[:J], int(nord.value)

    def priority_timing(self):
        ms, nb = C.c_double(0), C.c_uint64(0)
        self._check(self._L.cns_priority_timing(self._h, C.byref(ms), C.byref(nb)))
        return {"kernels_ms": ms.value, "algorithmic_bytes": int(nb.value)}

    # -- run-limit admission (include/crane_gpu/run_limits.h) --------------------------------------
    def set_run_limits(self, tables):
        """Limits + usage at the start of the commit loop (AccountMetaContainer state); after set_nodes."""
        t = tables.to_c()
        self._check(self._L.cns_set_run_limits(self._h, C.byref(t)))
        self._lim_tables = tables

    def apply_run_limits(self, jobs):
        """CheckAndMallocMetaResource over the last node_select's results, in the order of `jobs`
        (JobScheduler.cpp:1492-1573).  Returns (limit_reason[J] u8, num_admitted)."""
        self.upload_limit_jobs(jobs)
        self.run_limits_resident()
        return self.download_limits()

    def upload_limit_jobs(self, jobs):
        cj = jobs.to_c()
        self._check(self._L.cns_upload_limit_jobs(self._h, C.byref(cj)))
        self._lim_jobs = jobs

    def run_limits_resident(self):
        self._check(self._L.cns_run_limits_resident(self._h))

    def download_limits(self):
        J = self._lim_jobs.num_jobs
        out = np.zeros(max(J, 1), np.uint8)
        adm = C.c_uint64(0)
        self._check(self._L.cns_download_limits(self._h, out.ctypes.data_as(C.c_void_p), C.byref(adm)))
        return out[:J], int(adm.value)

    def limit_timing(self) -> dict:
        from . import limits as lm
        t = lm.CnsLimitTiming()
        self._check(self._L.cns_get_limit_timing(self._h, C.byref(t)))
        return {f: getattr(t, f) for f, _ in lm.CnsLimitTiming._fields_}

    def usage(self):
        """Usage tables after the last admission (what DoMallocResource_ left)."""
        u = self._lim_tables.empty_usage()
        self._check(self._L.cns_get_usage(self._h, *u.pointers()))
        return u

    # -- step scheduler (include/crane_gpu/steps.h) ------------------------------------------------
    def schedule_steps(self, step_jobs, steps):
        """JobInCtld::SchedulePendingSteps for every job with pending steps (CtldPublicDefs.cpp:2038-2159); after
        set_nodes (GRES layout).  Returns (StepResults, kernel_ms)."""
        from . import steps as st
        out = st.StepResults(step_jobs, steps)
        cj, cs, co = step_jobs.to_c(), steps.to_c(), out.to_c()
        ms = C.c_double(0)
        self._check(self._L.cns_schedule_steps(self._h, C.byref(cj), C.byref(cs), C.byref(co), C.byref(ms)))
        return out, ms.value

    def close(self):
        if self._h:
            self._L.cns_destroy(self._h)
            self._h = C.c_void_p()
            self._pinned = {}   # (cns_destroy released them: arrays handed out by pinned() are dead from here on)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- per-cycle snapshot -----------------------------------------------------------------------
    def set_nodes(self, cluster: abi.Cluster):
        c = cluster.to_c()
        self._check(self._L.cns_set_nodes(self._h, C.byref(c)))
        self._cluster = cluster

    def partition_status(self) -> np.ndarray:
        """[P] cns_partition_status of the snapshot: 0 served; else the partition's group lies outside the engine's limits (its jobs come
        back with REASON_ENGINE_REFUSED and belong to the caller's CPU scheduler; every other partition is served)."""
        out = np.zeros(max(self._cluster.num_partitions, 1), np.uint8)
        self._check(self._L.cns_get_partition_status(self._h, out.ctypes.data_as(C.c_void_p), C.c_uint32(len(out))))
        return out[:self._cluster.num_partitions]

    def set_reservations(self, reservations: abi.Reservations | None):
        """Reservations of the cycle (JobScheduler.cpp:6619-6679): after set_nodes, before set_running."""
        if reservations is None:
            self._check(self._L.cns_set_reservations(self._h, None))
        else:
            r = reservations.to_c()
            self._check(self._L.cns_set_reservations(self._h, C.byref(r)))

    def set_running(self, running: abi.Running | None):
        if running is None:
            self._check(self._L.cns_set_running(self._h, None))
        else:
            r = running.to_c()
            self._check(self._L.cns_set_running(self._h, C.byref(r)))

    def set_host_threads(self, n: int):
        """Host threads of the engine's own pass over the queue inside cns_select / cns_upload_jobs (0: CNS_HOST_THREADS, else up to 16)."""
        self._check(self._L.cns_set_host_threads(self._h, C.c_uint32(n)))

    # -- NodeSelect --------------------------------------------------------------------------------
    def node_select(self, now: int, jobs: abi.Jobs, out: "abi.Placements | None" = None) -> abi.Placements:
        """SchedulerAlgo::NodeSelect(now, running_jobs, pending_jobs) in one call (`out`: result arrays to reuse)."""
        if out is None:
            out = abi.Placements(jobs.num_jobs, jobs.total_places())
        assert out.num_jobs == jobs.num_jobs and out.capacity >= jobs.total_places()
        cj, co = jobs.to_c(), out.to_c()
        self._check(self._L.cns_select(self._h, C.c_int64(now), C.byref(cj), C.byref(co)))
        self._jobs = jobs
        return out

    # page-locked host buffers (cns_host_alloc): the caller's job table and result arrays at full PCIe rate
    def pinned(self, n: int, dtype, fill=None) -> np.ndarray:
        """A numpy array of n elements in page-locked host memory of this handle.  It lives until free_pinned(a) or close();
        after close() the memory is gone and the array must not be touched."""
        dt = np.dtype(dtype)
        p = C.c_void_p()
        self._check(self._L.cns_host_alloc(self._h, C.c_uint64(max(n, 1) * dt.itemsize), C.byref(p)))
        a = np.frombuffer((C.c_char * (max(n, 1) * dt.itemsize)).from_address(p.value), dtype=dt)
        self._pinned[p.value] = a
        if fill is not None:
            a[:] = fill
        return a

    def free_pinned(self, a: np.ndarray):
        """Gives a buffer of pinned() (or any view of it) back (cns_host_free)."""
        base = a
        while isinstance(base.base, np.ndarray):
            base = base.base
        ptr = base.ctypes.data
        if ptr not in self._pinned:
            raise EngineError(-1, "free_pinned: not a buffer of pinned() on this handle")
        del self._pinned[ptr]
        self._check(self._L.cns_host_free(self._h, C.c_void_p(ptr)))

    def pinned_jobs(self, jobs: abi.Jobs, into: "abi.Jobs | None" = None) -> abi.Jobs:
        """The same job table with every array in page-locked memory (what an adapter that packs into such buffers hands over).
        Every call WITHOUT `into` allocates a fresh set of buffers: call it once and refill per cycle with
        `pj = eng.pinned_jobs(jobs, into=pj)` — arrays of `into` whose size still fits are overwritten in place, the others are
        freed and allocated anew; free_pinned_jobs(pj) gives a set back."""
        import dataclasses
        kw = {}
        for f in dataclasses.fields(jobs):
            v = getattr(jobs, f.name)
            old = getattr(into, f.name) if into is not None else None
            if v is None:
                if old is not None:
                    self.free_pinned(old)
                kw[f.name] = None
                continue
            if old is not None and old.dtype == v.dtype and old.size == v.size:
                old.reshape(-1)[:] = v.reshape(-1)
                kw[f.name] = old.reshape(v.shape)
                continue
            if old is not None:
                self.free_pinned(old)
            a = self.pinned(v.size, v.dtype)
            a[:v.size] = v.reshape(-1)
            kw[f.name] = a[:v.size].reshape(v.shape)
        return abi.Jobs(**kw)

    def free_pinned_jobs(self, pj: abi.Jobs):
        import dataclasses
        for f in dataclasses.fields(pj):
            v = getattr(pj, f.name)
            if v is not None:
                self.free_pinned(v)

    def pinned_placements(self, jobs: abi.Jobs) -> abi.Placements:
        """Result arrays in page-locked memory, to be reused across cycles: node_select(now, jobs, out=...)."""
        return abi.Placements(jobs.num_jobs, jobs.total_places(), alloc=self.pinned)

    # split form: inputs resident in HBM before the timed region (bench.py)
    def node_select_preempt(self, now: int, jobs: abi.Jobs, preempt: "abi.Preempt"):
        """NodeSelect with preemption (include/crane_gpu/preempt.h): TryPreempt_ between the res_total selection and the
        backfill; -> (Placements, PreemptOut).  The running set must have been handed in with set_running."""
        out = abi.Placements(jobs.num_jobs, jobs.total_places())
        pout = abi.PreemptOut(jobs.num_jobs, len(preempt.rn_job_id))
        cj, co, cp, cpo = jobs.to_c(), out.to_c(), preempt.to_c(), pout.to_c()
        self._check(self._L.cns_select_preempt(self._h, C.c_int64(now), C.byref(cj), C.byref(cp), C.byref(co), C.byref(cpo)))
        self._jobs = jobs
        return out, pout

    def upload_jobs(self, jobs: abi.Jobs):
        cj = jobs.to_c()
        self._check(self._L.cns_upload_jobs(self._h, C.byref(cj)))
        self._jobs = jobs

    def run_resident(self, now: int):
        self._check(self._L.cns_run_resident(self._h, C.c_int64(now)))

    def download(self) -> abi.Placements:
        out = abi.Placements(self._jobs.num_jobs, self._jobs.total_places())
        co = out.to_c()
        self._check(self._L.cns_download(self._h, C.byref(co)))
        return out

    def device_results(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._check(self._L.cns_device_results(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    # -- one process per device: the all-gather of the packed results over RCCL (csrc/group_host.inc) ----------------
    def results_layout(self) -> dict:
        """Byte offsets of the packed result buffer of the last upload (cns_results_layout)."""
        o = abi.CnsResultsOffsets()
        self._check(self._L.cns_results_layout(self._h, C.byref(o)))
        return {f: getattr(o, f) for f, _ in abi.CnsResultsOffsets._fields_}

    @staticmethod
    def comm_unique_id() -> bytes:
        """rank 0: the communicator's id (ship it to every rank: torch.distributed's store, MPI, a file)"""
        buf = (C.c_uint8 * 128)()
        rc = lib().cns_comm_unique_id(buf)
        if rc != 0:
            raise EngineError(rc, (lib().cns_last_error(None) or b"").decode())
        return bytes(buf)

    def comm_init_rank(self, nranks: int, rank: int, uid: bytes):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self._check(self._L.cns_comm_init_rank(self._h, C.c_uint32(nranks), C.c_uint32(rank), buf))

    def comm_destroy(self):
        self._check(self._L.cns_comm_destroy(self._h))

    def allgather_results(self, slot_bytes: int) -> int:
        """collective: every rank's packed results, rank r at [r * slot_bytes) of a device buffer of this engine -> its address"""
        p = C.c_void_p()
        self._check(self._L.cns_allgather_results(self._h, C.c_uint64(slot_bytes), C.byref(p)))
        return p.value

    def download_gathered(self, nbytes: int) -> np.ndarray:
        out = np.empty(nbytes, np.uint8)
        self._check(self._L.cns_download_gathered(self._h, out.ctypes.data_as(C.c_void_p), C.c_uint64(nbytes)))
        return out

    def gather_timing(self):
        ms, b = C.c_double(), C.c_uint64()
        self._L.cns_gather_timing(self._h, C.byref(ms), C.byref(b))
        return ms.value, b.value

    def timing(self) -> dict:
        t = abi.CnsTiming()
        self._check(self._L.cns_get_timing(self._h, C.byref(t)))
        return {f: getattr(t, f) for f, _ in abi.CnsTiming._fields_}

    def last_kernel(self) -> str:
        """Selection kernel of the last run: 'k_pipe<NPL>' (decoupled test / commit pipeline) or 'k_select<NPL>'."""
        return (self._L.cns_debug_last_kernel(self._h) or b"").decode()

    # -- parity helpers ------------------------------------------------------------------------------
    def costs(self) -> np.ndarray:
        c = np.zeros(len(self._cluster.part_nodes), np.float64)
        self._check(self._L.cns_debug_get_costs(self._h, c.ctypes.data_as(C.c_void_p)))
        return c

    def prof(self) -> np.ndarray:
        """[P, 32] cycle counters of the last run (zeros unless built with -DCNS_PROF)."""
        P = self._cluster.num_partitions
        out = np.zeros(P * 32, np.uint64)
        self._check(self._L.cns_debug_get_prof(self._h, out.ctypes.data_as(C.c_void_p), C.c_uint32(P * 32)))
        return out.reshape(P, 32)

    WIDE_STATS = ("looks_empty", "looks_consumed", "leader_polls_ring_full", "stops", "flushes", "redo_with_exclusion",
                  "serial_jobs", "resource_verdicts", "partitions_straddling_xcds",
                  "windows", "jobs_decided_in_windows", "windows_that_decided_nothing", "prefetch_wave_gave_up",   # (round 5: several jobs per exchange, wide_kernel.inc)
                  "home_workgroups")   # (round 6: summed over the partitions — 8 partitions x 3 homes = 24)

    def wide_stats(self) -> dict:
        """Always-on protocol counters of k_wide's last run, summed over the partitions (every build; zeros for the other
        kernels): who waited for whom (supervisor looks without / with a decision, leader polls in front of a full ring) and
        how often the chain was broken (stops, flushes) and mended (redo with an excluded candidate, serial jobs, "Resource")."""
        self._L.cns_debug_engine_partitions.restype = C.c_uint32
        P = int(self._L.cns_debug_engine_partitions(self._h)) or self._cluster.num_partitions   # (groups of partitions that share nodes + reservations)
        out = np.zeros(P * 48, np.uint64)
        self._check(self._L.cns_debug_get_prof(self._h, out.ctypes.data_as(C.c_void_p), C.c_uint32(P * 48)))
        st = out[P * 32:].reshape(P, 16).sum(axis=0)
        return {k: int(v) for k, v in zip(self.WIDE_STATS, st)}

    def timeline(self, node: int, cap: int = 1100):
        n = C.c_uint32(0)
        t = np.zeros(cap, np.int64); cpu = np.zeros(cap, np.int64)
        mem = np.zeros(cap, np.uint64); lo = np.zeros(cap, np.uint64)
        hi = np.zeros(cap, np.uint64); g = np.zeros(cap, np.uint64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._check(self._L.cns_debug_get_timeline(self._h, C.c_uint32(node), C.c_uint32(cap), C.byref(n),
                                                   p(t), p(cpu), p(mem), p(lo), p(hi), p(g)))
        k = min(n.value, cap)
        w2 = np.zeros(cap, np.uint64); w3 = np.zeros(cap, np.uint64)
        self._check(self._L.cns_debug_get_timeline_cores(self._h, C.c_uint32(node), C.c_uint32(cap), p(w2), p(w3)))
        return {"t": t[:k], "cpu_raw": cpu[:k], "mem": mem[:k], "core_lo": lo[:k], "core_hi": hi[:k], "gres": g[:k],
                "core_w2": w2[:k], "core_w3": w3[:k]}


class GpuNodeSelectorGroup:
    """ONE process, N devices (cns_group_*: what the C++ adapter's GpuNodeSelectionAlgo(std::vector<int> devices) drives): the groups of
    partitions dealt over the devices, every shard on its own host thread, the packed results all-gathered on the devices (RCCL when the
    devices are distinct) and merged back into queue order.  A repeated ordinal ([0, 0]) exercises the path on a one-GPU box."""

    def __init__(self, devices, scheduled_batch_size: int = 0, max_job_num_per_node: int = 0, max_time_window_sec: int = 0):
        self._L = lib()
        self._g = C.c_void_p()
        cfg = abi.CnsConfig(abi.CNS_ABI_VERSION, 0, scheduled_batch_size, max_job_num_per_node, 0, max_time_window_sec)
        dev = (C.c_int32 * len(devices))(*devices)
        rc = self._L.cns_group_create(C.byref(cfg), dev, C.c_uint32(len(devices)), C.byref(self._g))
        if rc != 0:
            msg = self._L.cns_group_last_error(None)
            self._g = C.c_void_p()
            raise EngineError(rc, msg.decode() if msg else "")
        self._cluster = None

    def _check(self, rc: int):
        if rc != 0:
            msg = self._L.cns_group_last_error(self._g)
            raise EngineError(rc, msg.decode() if msg else "")

    def close(self):
        if self._g:
            self._L.cns_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_nodes(self, cluster: abi.Cluster):
        c = cluster.to_c()
        self._check(self._L.cns_group_set_nodes(self._g, C.byref(c)))
        self._cluster = cluster

    def set_reservations(self, reservations):
        if reservations is None:
            self._check(self._L.cns_group_set_reservations(self._g, None))
        else:
            r = reservations.to_c()
            self._check(self._L.cns_group_set_reservations(self._g, C.byref(r)))

    def set_running(self, running):
        if running is None:
            self._check(self._L.cns_group_set_running(self._g, None))
        else:
            r = running.to_c()
            self._check(self._L.cns_group_set_running(self._g, C.byref(r)))

    def node_select(self, now: int, jobs: abi.Jobs, out: "abi.Placements | None" = None) -> abi.Placements:
        if out is None:
            out = abi.Placements(jobs.num_jobs, jobs.total_places())
        cj, co = jobs.to_c(), out.to_c()
        self._check(self._L.cns_group_select(self._g, C.c_int64(now), C.byref(cj), C.byref(co)))
        return out

    def info(self) -> dict:
        i = abi.CnsGroupInfo()
        self._check(self._L.cns_group_get_info(self._g, C.byref(i)))
        d = {f: getattr(i, f) for f, _ in abi.CnsGroupInfo._fields_}
        d["gather_mode"] = {1: "rccl", 2: "device-copies"}.get(d["gather_mode"], d["gather_mode"])
        return d

    def device_of_partition(self, p: int) -> int:
        return int(self._L.cns_group_device_of_partition(self._g, C.c_uint32(p)))

    def partition_status(self) -> np.ndarray:
        """cns_partition_status per partition of the caller (0: served), whichever device it was dealt to."""
        st = np.zeros(self._cluster.num_partitions, np.uint8)
        self._check(self._L.cns_group_get_partition_status(self._g, st.ctypes.data_as(C.c_void_p), C.c_uint32(len(st))))
        return st

    def last_kernels(self):
        n = self._L.cns_group_size(self._g)
        return [(self._L.cns_debug_last_kernel(self._L.cns_group_handle(self._g, d)) or b"").decode() for d in range(n)]

    def costs_of_device(self, d: int, n: int) -> np.ndarray:
        c = np.zeros(n, np.float64)
        h = self._L.cns_group_handle(self._g, d)
        rc = self._L.cns_debug_get_costs(C.c_void_p(h), c.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise EngineError(rc, (self._L.cns_last_error(C.c_void_p(h)) or b"").decode())
        return c
