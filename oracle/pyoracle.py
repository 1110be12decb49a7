"""ctypes driver of the CPU oracle (oracle/liboracle.so).

ORACLE = TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never from cranesched_amd/ (the product).  See
oracle/res_algebra.hpp for what it restates and how it is pinned to the reference's own code (oracle/_ref).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from cranesched_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MASK, LITERAL = 0, 1


class OraRes(C.Structure):
    _fields_ = [("cpu", C.c_int64), ("mem", C.c_uint64), ("clo", C.c_uint64), ("chi", C.c_uint64),
                ("gres", C.c_uint64), ("c2", C.c_uint64), ("c3", C.c_uint64)]   # c2 / c3: core ids 128..255

    def tup(self):
        return (self.cpu, self.mem, self.clo, self.chi, self.gres, self.c2, self.c3)


class OraReq(C.Structure):
    _fields_ = [("cpu", C.c_int64), ("mem", C.c_uint64), ("gtot", C.c_uint8 * abi.MAX_GRES_NAMES),
                ("gspec", C.c_uint8 * abi.MAX_GRES_CLASSES)]


def build(force: bool = False) -> str:
    path = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("crane_oracle.cpp", "sched_oracle.hpp", "res_algebra.hpp", "prio_oracle.hpp", "limits_oracle.hpp", "steps_oracle.hpp")]
    srcs.append(os.path.join(_HERE, "..", "include", "crane_gpu", "node_select.h"))
    srcs.append(os.path.join(_HERE, "..", "include", "crane_gpu", "run_limits.h"))
    srcs.append(os.path.join(_HERE, "..", "include", "crane_gpu", "steps.h"))
    srcs.append(os.path.join(_HERE, "..", "include", "crane_gpu", "preempt.h"))
    stale = force or not os.path.exists(path) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(path) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    return path


def lib(backend: str = "oracle"):
    """`oracle`: the restatement (liboracle.so).  `ref` / `ref_hash`: the REFERENCE's own sliced sources behind the
    same entry points (oracle/_ref/libcrane_ref*.so, oracle/ref_build/); see ref_available()."""
    global _LIB
    if backend != "oracle":
        return ref_lib(backend)
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.ora_seconds.restype = C.c_double
        _LIB.ora_jobs_ordered.restype = C.c_uint64
        _LIB.ora_free.restype = None
    return _LIB


_REF_LIBS: dict = {}
_REF_FILES = {"ref": "libcrane_ref.so", "ref_hash": "libcrane_ref_hash.so"}
REFERENCE_ROOT = "/root/reference"


def build_ref(force: bool = False) -> bool:
    """Build oracle/_ref/*.so from the reference's own sources (needs /root/reference).  Returns availability."""
    paths = [os.path.join(_HERE, "_ref", f) for f in _REF_FILES.values()]
    if os.path.isdir(REFERENCE_ROOT):
        srcs = [os.path.join(_HERE, "ref_build", f) for f in
                ("extract.py", "ref_harness.cpp", "shim/absl_shim.h", "shim/crane_shim.h", "shim/crane_shim_ctld.h", "shim/fpm/fixed.hpp")]
        stale = force or any(not os.path.exists(p) for p in paths) or any(
            os.path.getmtime(s) > min(os.path.getmtime(p) for p in paths) for s in srcs)
        if stale:
            subprocess.run(["make", "-C", _HERE, "-B", "_ref"], check=True, capture_output=True)
    return all(os.path.exists(p) for p in paths)


def ref_available() -> bool:
    """True when the build of the reference's sliced sources is there (built here from /root/reference, or shipped
    prebuilt to a box without the reference)."""
    try:
        return build_ref()
    except subprocess.CalledProcessError:
        return False


def ref_lib(backend: str = "ref"):
    if backend not in _REF_LIBS:
        if not build_ref():
            raise RuntimeError("oracle/_ref is not built and /root/reference is not present")
        L = C.CDLL(os.path.join(_HERE, "_ref", _REF_FILES[backend]))
        L.ora_seconds.restype = C.c_double
        L.ora_jobs_ordered.restype = C.c_uint64
        L.ora_free.restype = None
        L.ref_last_error.restype = C.c_char_p
        _REF_LIBS[backend] = L
    return _REF_LIBS[backend]


class RefAsserted(RuntimeError):
    """A CRANE_ASSERT / ABSL_ASSERT of the reference's own code failed on this input."""


class RefUnsupported(RuntimeError):
    """The call asks for a constant the reference fixes at compile time (kAlgoMaxJobNumPerNode, kAlgoMaxTimeWindow)."""


def make_req(cpu=0, mem=0, gtot=(), gspec=()) -> OraReq:
    q = OraReq()
    q.cpu, q.mem = cpu, mem
    for i, v in enumerate(gtot):
        q.gtot[i] = v
    for i, v in enumerate(gspec):
        q.gspec[i] = v
    return q


def make_res(cpu=0, mem=0, clo=0, chi=0, gres=0, c2=0, c3=0) -> OraRes:
    return OraRes(cpu, mem, clo, chi, gres, c2, c3)


def feasible(layout: abi.GresLayout, algebra: int, req: OraReq, avail: OraRes, backend: str = "oracle"):
    out = OraRes()
    gl = layout.to_c()
    ok = lib(backend).ora_feasible(C.byref(gl), algebra, C.byref(req), C.byref(avail), C.byref(out))
    return (bool(ok), out.tup() if ok else None)


def binop(layout: abi.GresLayout, algebra: int, op: str, a: OraRes, b: OraRes, backend: str = "oracle"):
    code = {"ckmin": 0, "add": 1, "sub": 2, "le": 3}[op]
    out = OraRes()
    gl = layout.to_c()
    ret = lib(backend).ora_binop(C.byref(gl), algebra, code, C.byref(a), C.byref(b), C.byref(out))
    return bool(ret) if op == "le" else out.tup()


class OracleRun:
    """Result of one oracle cycle; keeps the C++ state alive for cost / timeline queries."""

    def __init__(self, handle, placements, cluster, backend: str = "oracle"):
        self._h, self.placements, self._cluster, self._backend = handle, placements, cluster, backend

    @property
    def seconds(self) -> float:
        return lib(self._backend).ora_seconds(self._h)

    @property
    def jobs_ordered(self) -> int:
        return lib(self._backend).ora_jobs_ordered(self._h)

    def costs(self) -> np.ndarray:
        c = np.zeros(len(self._cluster.part_nodes), np.float64)
        lib(self._backend).ora_get_costs(self._h, c.ctypes.data_as(C.c_void_p))
        return c

    def timeline(self, node: int, cap: int = 1100):
        n = C.c_uint32(0)
        t = np.zeros(cap, np.int64); cpu = np.zeros(cap, np.int64)
        mem = np.zeros(cap, np.uint64); lo = np.zeros(cap, np.uint64)
        hi = np.zeros(cap, np.uint64); g = np.zeros(cap, np.uint64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        lib(self._backend).ora_get_timeline(self._h, C.c_uint32(node), C.c_uint32(cap), C.byref(n), p(t), p(cpu), p(mem),
                               p(lo), p(hi), p(g))
        k = n.value
        w2 = np.zeros(cap, np.uint64); w3 = np.zeros(cap, np.uint64)
        lib(self._backend).ora_get_timeline_cores(self._h, C.c_uint32(node), C.c_uint32(cap), p(w2), p(w3))
        return {"t": t[:k], "cpu_raw": cpu[:k], "mem": mem[:k], "core_lo": lo[:k], "core_hi": hi[:k], "gres": g[:k],
                "core_w2": w2[:k], "core_w3": w3[:k]}

    def close(self):
        if self._h:
            lib(self._backend).ora_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def select(cluster: abi.Cluster, jobs: abi.Jobs, now: int, running: abi.Running | None = None,
           algebra: int = MASK, scheduled_batch_size: int = 0, max_job_num_per_node: int = 0,
           max_time_window_sec: int = 0, reservations: abi.Reservations | None = None,
           preempt: "abi.Preempt | None" = None, backend: str = "oracle") -> OracleRun:
    cfg = abi.CnsConfig(abi.CNS_ABI_VERSION, 0, scheduled_batch_size, max_job_num_per_node, 0,
                        max_time_window_sec)
    out = abi.Placements(jobs.num_jobs, jobs.total_places())
    cn, cj, co = cluster.to_c(), jobs.to_c(), out.to_c()
    cr = running.to_c() if running is not None else None
    cv = reservations.to_c() if reservations is not None else None
    h = C.c_void_p()
    L = lib(backend)

    def check(rc, what):
        if rc == 0:
            return
        if backend != "oracle":
            msg = (L.ref_last_error() or b"").decode()
            if rc == -3:
                raise RefAsserted(msg)
            if rc == -5:
                raise RefUnsupported(msg)
            raise RuntimeError(f"{what} ({backend}) failed: {rc}: {msg}")
        raise RuntimeError(f"{what} failed: {rc}")

    if preempt is not None:   # include/crane_gpu/preempt.h
        pout = abi.PreemptOut(jobs.num_jobs, len(running.end_sec) if running is not None else 0)
        cp, cpo = preempt.to_c(), pout.to_c()
        rc = L.ora_select_preempt(C.byref(cfg), C.byref(cn), C.byref(cr) if cr is not None else None,
                                      C.byref(cv) if cv is not None else None, C.c_int64(now), C.byref(cj), C.byref(cp),
                                      C.byref(co), C.byref(cpo), algebra, C.byref(h))
        check(rc, "ora_select_preempt")
        run = OracleRun(h, out, cluster, backend)
        run.preempt_out = pout
        return run
    rc = L.ora_select_resv(C.byref(cfg), C.byref(cn), C.byref(cr) if cr is not None else None,
                               C.byref(cv) if cv is not None else None,
                               C.c_int64(now), C.byref(cj), C.byref(co), algebra, C.byref(h))
    check(rc, "ora_select")
    return OracleRun(h, out, cluster, backend)


def priority_order(now: int, cfg, num_accounts: int, pending, running=None, backend: str = "oracle"):
    """CPU restatement of MultiFactorPriority (oracle/prio_oracle.hpp).  Returns (order, priority).
    backend `ref`: the reference's own MultiFactorPriority::GetOrderedJobPtrVec (JobScheduler.cpp:7606-7819)."""
    L = lib(backend)
    J = pending.num_jobs
    R = running.num_jobs if running is not None else 0
    order = np.empty(max(J, 1), np.uint32)
    prio = np.empty(max(J, 1), np.float64)
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    z = lambda: None
    rc = L.ora_priority_order(
        C.c_int64(now), C.c_uint64(cfg.max_age_sec), C.c_uint32(cfg.weight_age), C.c_uint32(cfg.weight_fair_share),
        C.c_uint32(cfg.weight_job_size), C.c_uint32(cfg.weight_partition), C.c_uint32(cfg.weight_qos),
        C.c_uint32(1 if cfg.favor_small else 0), C.c_uint32(num_accounts), C.c_uint32(J), p(pending.submit_sec),
        p(pending.qos_priority), p(pending.partition_priority), p(pending.node_num), p(pending.total_cpu_raw),
        p(pending.total_mem), p(pending.account), p(pending.cached_priority), C.c_uint32(R),
        p(running.start_sec) if R else z(), p(running.qos_priority) if R else z(),
        p(running.partition_priority) if R else z(), p(running.node_num) if R else z(),
        p(running.alloc_cpu_raw) if R else z(), p(running.alloc_mem) if R else z(), p(running.account) if R else z(),
        p(order), p(prio))
    if rc != 0:
        raise ValueError("ora_priority_order: account id out of range")
    return order[:J], prio[:J]


def run_limits(layout: abi.GresLayout, tables, jobs, placements: abi.Placements, backend: str = "oracle"):
    """CPU restatement of the commit loop's run-limit admission (oracle/limits_oracle.hpp).

    `placements` are a NodeSelect result (of the oracle or of the engine).  Returns (reason[J] u8, admitted, Usage).
    backend `ref`: the reference's own AccountMetaContainer::CheckAndMallocMetaResource (AccountMetaContainer.cpp:180-224
    with CheckRunLimits_ :891-1028, the per-entity checks :508-687, CheckTres_ / CheckGres_ :345-365,1030-1050 and
    DoMallocResource_ :1067-1124), compiled from /root/reference (oracle/_ref).  Its reasons are strings; the code
    returned is the FIRST code of include/crane_gpu/run_limits.h with that string (codes 2 and 5 share
    "QosCpuResourceLimit"): compare through limits.REASON_STRINGS.
    """
    from cranesched_amd import limits as lm
    L = lib(backend)
    reason = np.full(max(jobs.num_jobs, 1), 0, np.uint8)
    adm = C.c_uint64(0)
    usage = tables.empty_usage()
    gl, ct, cj, cp = layout.to_c(), tables.to_c(), jobs.to_c(), placements.to_c()
    rc = L.ora_run_limits(C.byref(gl), C.byref(ct), C.byref(cj), C.byref(cp), reason.ctypes.data_as(C.c_void_p),
                          C.byref(adm), *usage.pointers())
    if rc == -1:
        raise ValueError("ora_run_limits: key index out of range")
    if rc != 0:
        raise RuntimeError(f"ora_run_limits ({backend}) failed: {rc}: {L.ref_last_error().decode() if backend != 'oracle' else ''}")
    return reason[:jobs.num_jobs], adm.value, usage


def schedule_steps(layout: abi.GresLayout, step_jobs, steps, algebra: int = MASK, backend: str = "oracle"):
    """CPU restatement of JobInCtld::SchedulePendingSteps for every job (oracle/steps_oracle.hpp).  Returns StepResults.
    backend `ref`: the reference's own JobInCtld::SchedulePendingSteps (CtldPublicDefs.cpp:2038-2159), compiled from
    /root/reference (oracle/_ref); a job's nodes must be given in ascending node index (the order its map walk has there)."""
    from cranesched_amd import steps as st
    out = st.StepResults(step_jobs, steps)
    gl, cj, cs, co = layout.to_c(), step_jobs.to_c(), steps.to_c(), out.to_c()
    L = lib(backend)
    rc = L.ora_schedule_steps(C.byref(gl), C.byref(cj), C.byref(cs), C.byref(co), C.c_int(algebra))
    if rc != 0:
        raise RuntimeError(f"ora_schedule_steps ({backend}) failed: {rc}: {L.ref_last_error().decode() if backend != 'oracle' else ''}")
    return out


def license_check(total, used, reserved, last_deficit, requests, is_or, backend: str = "ref"):
    """THE REFERENCE'S OWN LicenseManager::CheckLicenseCountSufficient (LicenseManager.cpp:167-221, sliced at build time) on
    a license table (four uint32 columns) and the ordered jobs' requests: requests[j] = [(license index, count), ...] in
    request order (an index >= len(total): a license the table does not know).  -> (rejected [J] bool, actual [J] lists of
    (license index, count) sorted by index).  There is no restated oracle of this pass: the product's host pass is compared
    with the reference build directly (tests/test_ref_pin.py)."""
    L = ref_lib(backend)
    u32 = lambda a: np.ascontiguousarray(a, np.uint32)
    total, used, reserved, last_deficit = u32(total), u32(used), u32(reserved), u32(last_deficit)
    J = len(requests)
    off = np.zeros(J + 1, np.uint32)
    off[1:] = np.cumsum([len(r) for r in requests])
    flat = [x for r in requests for x in r]
    rl = u32([x[0] for x in flat] or [0]); rc = u32([x[1] for x in flat] or [0])
    io = np.ascontiguousarray(is_or, np.uint8)
    rej = np.zeros(max(J, 1), np.uint8)
    aoff = np.zeros(J + 1, np.uint32)
    al = np.zeros(max(len(flat), 1), np.uint32); ac = np.zeros(max(len(flat), 1), np.uint32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc_ = L.ref_license_check(C.c_uint32(len(total)), p(total), p(used), p(reserved), p(last_deficit), C.c_uint32(J), p(off), p(rl), p(rc), p(io),
                              p(rej), p(aoff), p(al), p(ac))
    if rc_ != 0:
        raise RuntimeError(f"ref_license_check: {L.ref_last_error().decode()}")
    return rej[:J].astype(bool), [[(int(al[x]), int(ac[x])) for x in range(aoff[j], aoff[j + 1])] for j in range(J)]
