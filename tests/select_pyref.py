"""A SECOND, independent restatement of the node-selection cycle in plain Python — test infrastructure only.

It shares no code with oracle/ (C++) or with the device code: written from the reference lines cited below and from
SURVEY.md Appendix A, with Python ints, sets and dicts (literal containers: core ids and GRES slots are sets of ids,
time maps are sorted lists).  tests/test_select_pyref.py runs it against the C++ oracle on the hand-derived scenarios and
on random small clusters; agreement of two restatements that share nothing is what "parity unpinned" can be narrowed to
while the reference itself cannot be compiled here.

Reference:  src/CraneCtld/JobScheduler.cpp:6127-6376 (select / backfill), :6507-6836 (cycle),
            src/CraneCtld/JobScheduler.h:41-55 (cost), :272-460 (NodeState), :492-595 (selector), :678-865 (backfill),
            src/Utilities/PublicHeader/PublicHeader.cpp:519-599 (GetFeasibleResourceInNode), :781-827, :886-890.
Canonicalisations (same as everywhere in this repo): cost ties break on the dense node index; a job's nodes are reported
in ascending index; unordered_map walks (GRES names / types) go in ascending id.
"""
from __future__ import annotations

import copy

INF = (1 << 63) - 1
MAX_WINDOW = 7 * 24 * 3600
MAX_JOBS_PER_NODE = 1000


class Res:
    """ResourceInNodeV3: cpu raw (x256), mem bytes, core id set, GRES: {(name, type): set(slot)}."""
    __slots__ = ("cpu", "mem", "cores", "gres")

    def __init__(self, cpu=0, mem=0, cores=(), gres=None):
        self.cpu, self.mem, self.cores = cpu, mem, set(cores)
        self.gres = {k: set(v) for k, v in (gres or {}).items() if v}

    def copy(self):
        return Res(self.cpu, self.mem, self.cores, self.gres)

    def key(self):
        return (self.cpu, self.mem, tuple(sorted(self.cores)), tuple(sorted((k, tuple(sorted(v))) for k, v in self.gres.items() if v)))


def res_sub(a: Res, b: Res):       # PublicHeader.cpp:789-796, CpuSet -= :758-766 (tolerant erase)
    a.cpu -= b.cpu
    a.mem -= b.mem
    a.cores -= b.cores
    for k, v in b.gres.items():
        if k in a.gres:
            a.gres[k] -= v
            if not a.gres[k]:
                del a.gres[k]


def res_add(a: Res, b: Res):       # :781-787
    a.cpu += b.cpu
    a.mem += b.mem
    a.cores |= b.cores
    for k, v in b.gres.items():
        a.gres.setdefault(k, set()).update(v)


def res_le(a: Res, b: Res) -> bool:  # :886-890 — core ids are NOT compared; requested slots must be a subset
    if a.cpu > b.cpu or a.mem > b.mem:
        return False
    return all(v <= b.gres.get(k, set()) for k, v in a.gres.items())


def res_ckmin(a: Res, b: Res):     # :815-827
    a.cpu = min(a.cpu, b.cpu)
    a.mem = min(a.mem, b.mem)
    if a.cores and b.cores:
        a.cores &= b.cores
    for k in list(a.gres):
        a.gres[k] &= b.gres.get(k, set())
        if not a.gres[k]:
            del a.gres[k]


class Req:
    """ResourceView of a request: cpu raw, mem, GRES per name: total count and {type: specified count}."""

    def __init__(self, cpu=0, mem=0, gtot=None, gspec=None):
        self.cpu, self.mem = cpu, mem
        self.gtot = dict(gtot or {})        # name -> total
        self.gspec = dict(gspec or {})      # (name, type) -> count


def compose(node: Req, task_cpu, task_mem, n) -> Req:   # req_node + req_task * n (:473-481, :601-611); tasks carry no GRES
    return Req(node.cpu + task_cpu * n, node.mem + task_mem * n, node.gtot, node.gspec)


def feasible(req: Req, avail: Res, types_of):   # GetFeasibleResourceInNode, :519-599 -> allocation or None
    if req.cpu > avail.cpu or req.mem > avail.mem:           # :522-523 (mem_sw is not tested)
        return None
    out = Res(req.cpu, req.mem)
    n = req.cpu // 256 if req.cpu >= 0 else -((-req.cpu) // 256)
    if n * 256 == req.cpu and avail.cores:                   # :528-530 integer request and the node tracks core ids
        if len(avail.cores) < n:                             # :534
            return None
        out.cores = set(sorted(avail.cores)[:n])             # the n lowest ids, :535-537
    names = sorted(set(req.gtot) | {k[0] for k in req.gspec})
    for name in names:
        total = req.gtot.get(name, 0)
        spec = {t: c for (nm, t), c in req.gspec.items() if nm == name and c}
        if total == 0 and not spec:
            continue
        if not any(k[0] == name and v for k, v in avail.gres.items()):   # :550-551 no slot of that name at all
            return None
        untyped = max(total - sum(spec.values()), 0)         # :556-559
        for t in sorted(spec):                               # specified types (canonical: ascending type id)
            slots = sorted(avail.gres.get((name, t), set()))
            if len(slots) < spec[t]:                         # :566-569
                return None
            take = slots[:spec[t]]
            rest = slots[spec[t]:]
            extra = rest[:untyped]                           # the same type serves the untyped remainder first, :577-578
            untyped -= len(extra)
            out.gres.setdefault((name, t), set()).update(take + extra)
        if untyped > 0:                                      # :582-592 the other types of the name, ascending
            for t in types_of(name):
                if t in spec or untyped == 0:
                    continue
                slots = sorted(avail.gres.get((name, t), set()))
                extra = slots[:untyped]
                untyped -= len(extra)
                if extra:
                    out.gres.setdefault((name, t), set()).update(extra)
        if untyped != 0:                                     # :594
            return None
    return out


# ---- std::priority_queue<node_info> as libstdc++ builds it (bits/stl_heap.h: __push_heap / __adjust_heap) ---------------
# comp(a, b) = a < b  <=>  a.ntasks > b.ntasks  (node_info::operator<, JobScheduler.cpp:6157-6163): the TOP is the smallest
# ntasks_on_node; which of several equal entries is evicted is an artefact of the heap layout.
def _comp(a, b):
    return a[0] > b[0]


def heap_push(h, x):
    h.append(x)
    hole = len(h) - 1
    parent = (hole - 1) // 2
    while hole > 0 and _comp(h[parent], x):
        h[hole] = h[parent]
        hole = parent
        parent = (hole - 1) // 2
    h[hole] = x


def heap_pop(h):
    top = h[0]
    value = h.pop()
    n = len(h)
    if n == 0:
        return top
    hole, second = 0, 0          # __adjust_heap(first, 0, n, value)
    while second < (n - 1) // 2:
        second = 2 * (second + 1)
        if _comp(h[second], h[second - 1]):
            second -= 1
        h[hole] = h[second]
        hole = second
    if (n & 1) == 0 and second == (n - 2) // 2:
        second = 2 * (second + 1)
        h[hole] = h[second - 1]
        hole = second - 1
    parent = (hole - 1) // 2     # __push_heap(first, hole, 0, value)
    while hole > 0 and _comp(h[parent], value):
        h[hole] = h[parent]
        hole = parent
        parent = (hole - 1) // 2
    h[hole] = value
    return top


class Node:
    def __init__(self, idx, total: Res):
        self.idx, self.total = idx, total
        self.avail0 = total.copy()
        self.allocated = []   # (end, Res) in arrival order
        self.reserved = []    # (start, end, Res): future reservations on the node
        self.tmap = []        # sorted [[t, Res]]

    def init_map(self, now, end=INF):   # InitTimeAvailResMap, JobScheduler.h:301-338
        # resource_changes: (time, allocate?, res); reserved_res first (- at its start, + at its end), then allocated_res
        # (+ at its end, and out of res_avail); sorted by time, "release before allocate" at equal times (h:317-322; the sort
        # is not stable in the reference — equal (time, flag) entries commute: all additions or all subtractions)
        changes = []
        for st, en, r in getattr(self, "reserved", []):
            changes.append((st, True, r))
            changes.append((en, False, r))
        for e, r in self.allocated:
            changes.append((e, False, r))
            res_sub(self.avail0, r)
        changes.sort(key=lambda x: (x[0], x[1]))
        self.tmap = [[now, self.avail0.copy()]]
        for t, alloc, r in changes:
            if t != self.tmap[-1][0]:
                self.tmap.append([t, self.tmap[-1][1].copy()])
            if alloc:
                res_sub(self.tmap[-1][1], r)
            else:
                res_add(self.tmap[-1][1], r)
        if self.tmap[-1][0] == end:
            self.tmap[-1][1] = Res()
        else:
            self.tmap.append([end, Res()])

    def release(self, start, end, res: Res):   # UpdateResourceInNode(..., is_release = true), JobScheduler.h:340-459
        ts = [e[0] for e in self.tmap]
        ib = max(i for i, t in enumerate(ts) if t <= start)
        if ts[ib] != start:
            self.tmap.insert(ib + 1, [start, self.tmap[ib][1].copy()])
            ib += 1
        ts = [e[0] for e in self.tmap]
        ie = max(i for i, t in enumerate(ts) if t <= end)
        if ts[ie] != end:
            self.tmap.insert(ie + 1, [end, self.tmap[ie][1].copy()])   # the copy is taken BEFORE the addition below
            ie += 1
        for i in range(ib, ie):
            res_add(self.tmap[i][1], res)

    def commit(self, start, end, res: Res):   # UpdateResourceInNode (allocate), JobScheduler.h:340-459
        ts = [e[0] for e in self.tmap]
        ib = max(i for i, t in enumerate(ts) if t <= start)
        if ts[ib] != start:
            self.tmap.insert(ib + 1, [start, self.tmap[ib][1].copy()])
            ib += 1
        ts = [e[0] for e in self.tmap]
        ie = max(i for i, t in enumerate(ts) if t <= end)
        if ts[ie] != end:
            self.tmap.insert(ie + 1, [end, self.tmap[ie][1].copy()])   # copies the value BEFORE the subtraction below
            ie += 1
        for i in range(ib, ie):
            res_sub(self.tmap[i][1], res)


class Cycle:
    """One NodeSelect cycle (JobScheduler.cpp:6507-6836) without reservations / preemption / licenses."""

    def __init__(self, now, totals, part_nodes, schedulable=None, types_of=None, max_jobs_per_node=MAX_JOBS_PER_NODE,
                 max_window=MAX_WINDOW):
        self.now, self.max_jobs, self.max_window = now, max_jobs_per_node, max_window
        self.types_of = types_of or (lambda name: [])
        n = len(totals)
        sched = schedulable if schedulable is not None else [1] * n
        in_part = set(x for p in part_nodes for x in p)
        self.nodes = {i: Node(i, totals[i].copy()) for i in range(n) if sched[i] and i in in_part}   # cpp:6584-6606
        self.parts = [[i for i in p if i in self.nodes] for p in part_nodes]
        self.cost = None

    def add_running(self, end, allocs):          # cpp:6513-6514, :6681-6709
        end = max(end, self.now + 1)
        for node, res in allocs:
            if node in self.nodes:
                self.nodes[node].allocated.append((end, res))

    def start(self):
        for nd in self.nodes.values():
            nd.init_map(self.now)
        # one cost per (partition, node): NodeRater ctor, JobScheduler.h:498-511, in allocated_res order
        self.cost = []
        for p in self.parts:
            c = {}
            for i in p:
                nd, v = self.nodes[i], 0.0
                for st, en, r in nd.reserved:                       # reserved_res first, over [start, end) (h:502-506)
                    v += float(en - st) * ((float(r.cpu) / 256.0) / (float(nd.total.cpu) / 256.0))
                for e, r in nd.allocated:
                    v += float(e - self.now) * ((float(r.cpu) / 256.0) / (float(nd.total.cpu) / 256.0))
                c[i] = v
            self.cost.append(c)

    # -- get_max_tasks, cpp:6171-6186
    def max_tasks(self, job, res: Res):
        f = feasible(job["min_view"], res, self.types_of)
        if f is None:
            return 0
        left = res.copy()
        res_sub(left, f)
        t = job["tmin"]
        one = Req(job["tcpu"], job["tmem"])
        while t < job["tmax"]:
            f = feasible(one, left, self.types_of)
            if f is None:
                break
            t += 1
            res_sub(left, f)
        return t

    # -- GetNodesAndTrySchedule_, cpp:6147-6369: returns ("now", picks) | ("later", picks) | None
    def try_schedule(self, job):
        p = job["part"]
        E = self.now + job["L"]
        k, ntasks = job["k"], job["ntasks"]
        h_total, sum_total, h_avail, sum_avail = [], 0, [], 0
        for i in sorted(self.parts[p], key=lambda x: (self.cost[p][x], x)):     # :6188 ascending (cost, node)
            nd = self.nodes[i]
            if len(nd.tmap) >= self.max_jobs:                                    # :6194
                continue
            if job["incl"] and i not in job["incl"]:                             # :6202-6220
                continue
            if i in job["excl"]:
                continue
            tt = self.max_tasks(job, nd.total)                                   # :6222
            if tt == 0:
                continue
            if len(h_total) < k or sum_total < ntasks:                           # :6233-6242
                sum_total += tt
                heap_push(h_total, (tt, nd.total, i))
                if len(h_total) > k:
                    sum_total -= h_total[0][0]
                    heap_pop(h_total)
            if job["exclusive"]:                                                 # :6249-6271
                if any(not res_le(nd.total, r) for t, r in nd.tmap if t < E):
                    continue
                cand = (tt, nd.total, i)
            else:                                                                # :6272-6299
                if feasible(job["min_view"], nd.avail0, self.types_of) is None:  # cycle-start res_avail, never updated
                    continue
                m = nd.avail0.copy()
                for t, r in nd.tmap:
                    if t >= E:
                        break
                    res_ckmin(m, r)
                ta = self.max_tasks(job, m)
                if ta == 0:
                    continue
                cand = (ta, m, i)
            sum_avail += cand[0]
            heap_push(h_avail, cand)
            if len(h_avail) > k:
                sum_avail -= h_avail[0][0]
                heap_pop(h_avail)
            if len(h_avail) == k and sum_avail >= ntasks:                        # :6294-6297
                break

        def distribute(h):                                                       # :6304-6325 / :6345-6367
            rest = ntasks - k
            picks = []
            while h:
                cap, res, i = h[0]               # top = smallest ntasks_on_node
                t = min(rest, cap - 1) + 1
                if job["exclusive"]:
                    alloc = res.copy()
                else:
                    alloc = feasible(compose(job["node_view"], job["tcpu"], job["tmem"], t), res, self.types_of)
                    assert alloc is not None
                picks.append((i, t, alloc))
                rest -= t - 1
                heap_pop(h)
            return picks

        if len(h_avail) == k and sum_avail >= ntasks:
            return "now", distribute(h_avail)
        if len(h_total) < k or sum_total < ntasks:                               # :6335-6343
            return None
        return "later", distribute(h_total)

    # -- Backfill_ / EarliestStartSubsetSelector on exactly the k nodes (JobScheduler.h:792-865): the earliest time at
    # which every node satisfies `alloc <= entry` for a whole time limit; candidate starts are `now` and the map keys
    def earliest_start(self, job, picks):
        L = job["L"]
        cands = sorted({self.now} | {t for i, _, _ in picks for t, _ in self.nodes[i].tmap if t > self.now})
        for s in cands:
            if s == INF or s - self.now > self.max_window:                       # h:815
                return None
            ok = True
            for i, _, alloc in picks:
                tm = self.nodes[i].tmap
                for x, (t, r) in enumerate(tm):
                    nxt = tm[x + 1][0] if x + 1 < len(tm) else None
                    if nxt is not None and nxt <= s:
                        continue                 # entry ends before the window
                    if t >= s + L:
                        break                    # entry starts after the window
                    if not res_le(alloc, r):
                        ok = False
                        break
                if not ok:
                    break
            if ok:
                return s
        return None

    def run_job(self, job):
        """Returns (reason, start, [(node, ntasks, alloc Res)] ascending node).  reason: 0 none, 1 Priority, 2 Resource."""
        r = self.try_schedule(job)
        if r is None:
            return 2, 0, []
        kind, picks = r
        if kind == "now":
            start = self.now
        else:
            start = self.earliest_start(job, picks)
            if start is None:
                return 2, 0, []
        end = start + job["L"]
        p = job["part"]
        for i, t, alloc in picks:                                                # :6795 + NodeSelector::AllocateResource h:567-575
            nd = self.nodes[i]
            nd.commit(start, end, alloc)
            self.cost[p][i] += float(end - start) * ((float(alloc.cpu) / 256.0) / (float(nd.total.cpu) / 256.0))
        reason = 0
        if start != self.now:                                                    # :6797-6833 (no reservations here)
            reason = 1
            if any(not res_le(alloc, self.nodes[i].avail0) for i, _, alloc in picks):
                reason = 2
        return reason, start, sorted(picks, key=lambda x: x[0])


# ---------------------------------------------------------------------------------------------------------------------
# Reservations inside NodeSelect (JobScheduler.cpp:6619-6679 prologue, :6692-6707 running jobs inside a reservation,
# :6715-6719 time maps up to the reservation's end, :6729-6732 one scheduler per reservation, :6754-6760 dispatch,
# :6797-6830 reasons).  Written from those lines; shares nothing with oracle/sched_oracle.hpp.  Canonicalisation: the
# reservations (an unordered map in the reference) are taken in ascending index.
# ---------------------------------------------------------------------------------------------------------------------
REASON_RESV_NOT_FOUND = 6
REASON_RESOURCE_RESERVED = 3


class ResvCycle(Cycle):
    def __init__(self, *a, reservations=(), pending_resv=(), **kw):
        """reservations: dicts start, end, allocs = [(node, Res)]; pending_resv: reservation ids named by pending jobs."""
        super().__init__(*a, **kw)
        self.first_resv = {}          # craned_id_first_resv_map
        self.resv_sched = {}          # reservation id -> {node: Node} (active AND named by a pending job, :6652-6668)
        self.resv_end = {}
        pend = set(pending_resv)
        for v, rv in enumerate(reservations):
            if self.now >= rv["end"]:                                   # expired but not cleaned up (:6631-6634)
                continue
            for node, _ in rv["allocs"]:                                # :6635-6642
                self.first_resv[node] = min(self.first_resv.get(node, INF), rv["start"])
            if self.now >= rv["start"]:                                 # active: its share is allocated until its end (:6643-6651)
                for node, res in rv["allocs"]:
                    if node in self.nodes:
                        self.nodes[node].allocated.append((rv["end"], res))
                if v not in pend:
                    continue                                            # no pending jobs, skip (:6652-6655)
                self.resv_end[v] = rv["end"]
                self.resv_sched[v] = {node: Node(node, res.copy()) for node, res in rv["allocs"]}
            else:                                                       # future: a dip in the node's map (:6669-6678)
                for node, res in rv["allocs"]:
                    if node in self.nodes:
                        self.nodes[node].reserved.append((rv["start"], rv["end"], res))

    def add_running(self, end, allocs, resv=None):                      # :6681-6709
        if resv is None:
            return super().add_running(end, allocs)
        if resv not in self.resv_sched:
            return                                                      # "not found in resv_node_state_map": ignored
        end = max(end, self.now + 1)
        for node, res in allocs:
            if node in self.resv_sched[resv]:
                self.resv_sched[resv][node].allocated.append((end, res))

    def start(self):
        super().start()
        self.resv_cost = {}
        for v, nodes in self.resv_sched.items():                        # :6715-6719, :6729-6732
            c = {}
            for i, nd in nodes.items():
                nd.init_map(self.now, self.resv_end[v])
                val = 0.0
                for e, r in nd.allocated:
                    val += float(e - self.now) * ((float(r.cpu) / 256.0) / (float(nd.total.cpu) / 256.0))
                c[i] = val
            self.resv_cost[v] = c

    def run_job(self, job):
        v = job.get("resv")
        if v is None:
            reason, start, picks = super().run_job(job)
            if reason in (1, 2) and any(self.first_resv.get(i, INF) < self.now + job["L"] for i, _, _ in picks):
                reason = REASON_RESOURCE_RESERVED                       # :6798-6806 comes before the res_avail test
            return reason, start, picks
        if v not in self.resv_sched:
            return REASON_RESV_NOT_FOUND, 0, []                         # :6754-6759
        saved = (self.nodes, self.parts, self.cost)
        try:
            self.nodes, self.parts, self.cost = self.resv_sched[v], [sorted(self.resv_sched[v])], [self.resv_cost[v]]
            return Cycle.run_job(self, dict(job, part=0))               # reasons against the reservation's own res_avail (:6818-6829)
        finally:
            self.nodes, self.parts, self.cost = saved


# ---------------------------------------------------------------------------------------------------------------------
# Preemption: TryPreempt_ (JobScheduler.cpp:6378-6505), PreemptSegTree (JobScheduler.h:867-980), the bookkeeping of
# NodeSelect around it (:6545-6559, :6779-6795; JobScheduler.h:630-670).  Written from those lines; shares nothing with
# oracle/sched_oracle.hpp.  Canonicalisation: candidates that the reference's comparator leaves unordered (it sorts the
# iteration order of a hash set) are taken pending-first, then by ascending index.
# ---------------------------------------------------------------------------------------------------------------------
TICKS = 4_000_000_000   # absl::Duration resolves a quarter of a nanosecond; `(ed - st) / 2` truncates there


def res_is_zero(r: Res) -> bool:   # ResourceInNodeV3::IsZero, PublicHeader.cpp:798-801
    return r.cpu == 0 and r.mem == 0 and not r.cores and not any(r.gres.values())


class SegTree:
    """A lazily split segment tree over [st, ed): every node carries a resource, a `satisfied` flag and two pending tags."""

    def __init__(self, st, ed, target: Res):
        self.target = target
        self.root = self._node(st, ed, False, Res())

    @staticmethod
    def _node(st, ed, sat, res):
        return {"st": st, "ed": ed, "ls": None, "rs": None, "sat": sat, "res": res, "add": Res(), "sub": Res()}

    def _plus(self, n, r):      # add_res_: the flag is recomputed from THIS node's resource, whatever its children hold
        res_add(n["res"], r)
        n["sat"] = res_le(self.target, n["res"])
        if n["ls"] is not None:
            res_add(n["add"], r)

    def _minus(self, n, r):     # sub_res_
        res_sub(n["res"], r)
        n["sat"] = res_le(self.target, n["res"])
        if n["ls"] is not None:
            res_add(n["sub"], r)

    def _down(self, n):         # push_down_: the first visit splits the node, later visits hand the tags on
        if n["ls"] is None:
            mid = n["st"] + (n["ed"] - n["st"]) // 2
            n["ls"] = self._node(n["st"], mid, n["sat"], n["res"].copy())
            n["rs"] = self._node(mid, n["ed"], n["sat"], n["res"].copy())
            return
        if not res_is_zero(n["add"]):
            self._plus(n["ls"], n["add"]); self._plus(n["rs"], n["add"])
            n["add"] = Res()
        if not res_is_zero(n["sub"]):
            self._minus(n["ls"], n["sub"]); self._minus(n["rs"], n["sub"])
            n["sub"] = Res()

    def _walk(self, n, st, ed, r, plus):
        if n["ed"] <= st or ed <= n["st"]:
            return
        if st <= n["st"] and n["ed"] <= ed:
            (self._plus if plus else self._minus)(n, r)
            return
        self._down(n)
        self._walk(n["ls"], st, ed, r, plus)
        self._walk(n["rs"], st, ed, r, plus)
        n["sat"] = n["ls"]["sat"] and n["rs"]["sat"]      # push_up_

    def add(self, st, ed, r):
        self._walk(self.root, st, ed, r, True)

    def sub(self, st, ed, r):
        self._walk(self.root, st, ed, r, False)

    @property
    def satisfied(self):
        return self.root["sat"]


class PreemptCycle(ResvCycle):
    """Cycle (with reservations) + preemption.  Jobs are referred to as ("pd", index) / ("rn", index).  Every scheduler —
    a partition's or a reservation's — sees the qos_job_map of ITS NodeStates: `node_jobs` for the real nodes,
    `resv_node_jobs[v]` for the reservation's own (JobScheduler.cpp:6688 / :6705)."""

    def __init__(self, *a, qos_preempt=(), preempting=(), **kw):
        super().__init__(*a, **kw)
        self.qos_preempt = [list(x) for x in qos_preempt]
        self.preempting = set(preempting)       # m_preempting_set_ (job ids)
        self.cancelled = []
        self.rn = []                            # dicts: id, qos, qprio, start, end, allocs {node: Res}
        self.pd = {}                            # index -> dict: qos, qprio, prio, start, end, allocs, reason, nodes
        self.node_jobs = {}                     # node -> {qos: set(ref)}     (NodeState::qos_job_map)
        self.resv_node_jobs = {}                # reservation -> the same for its own NodeStates

    def add_running_job(self, job_id, qos, qprio, start, end, allocs, resv=None):
        end = max(end, self.now + 1)                                             # :6513-6514
        self.rn.append(dict(id=job_id, qos=qos, qprio=qprio, start=start, end=end, allocs=allocs, resv=resv))

    def start(self):
        ids = {r["id"] for r in self.rn}
        self.preempting &= ids                                                   # :6550-6558
        for r in self.rn:
            if r["id"] in self.preempting:
                r["end"] = self.now + 1
        for x, r in enumerate(self.rn):                                          # :6681-6690
            merged = {}
            v = r.get("resv")
            for node, res in r["allocs"]:
                if node in merged:
                    res_add(merged[node], res)
                else:
                    merged[node] = res.copy()
                if v is None:
                    if node in self.nodes:
                        self.nodes[node].allocated.append((r["end"], res))
                        self.node_jobs.setdefault(node, {}).setdefault(r["qos"], set()).add(("rn", x))
                elif v in self.resv_sched and node in self.resv_sched[v]:           # :6692-6707
                    self.resv_sched[v][node].allocated.append((r["end"], res))
                    self.resv_node_jobs.setdefault(v, {}).setdefault(node, {}).setdefault(r["qos"], set()).add(("rn", x))
            r["allocs"] = merged
        super().start()

    def try_preempt(self, job, jinfo, picks):
        """picks: the (node, ntasks, alloc) list of the res_total branch, in nodes_to_sched order.  -> list of refs or None."""
        plist = self.qos_preempt[jinfo["qos"]] if jinfo["qos"] < len(self.qos_preempt) else []
        if not plist:
            return None
        cand = set()
        for i, _, _ in picks:
            for q in plist:
                cand |= self.node_jobs.get(i, {}).get(q, set())
        if not cand:
            return None

        def key(ref):
            kind, x = ref
            if kind == "rn":
                r = self.rn[x]
                return (0 if r["id"] in self.preempting else 1, 1, r["qprio"], -r["start"], x)
            d = self.pd[x]
            return (1, 0, d["qprio"], d["prio"], x)
        order = sorted(cand, key=key)
        tk = lambda t: (t - self.now) * TICKS
        seg_end = tk(self.now + job["L"])
        trees = {}
        for i, _, alloc in picks:
            tr = SegTree(0, seg_end, alloc)
            tm = self.nodes[i].tmap
            for x, (t, r) in enumerate(tm):
                ed = tk(tm[x + 1][0]) if x + 1 < len(tm) else seg_end
                tr.add(tk(t), ed, r)
                if ed >= seg_end:
                    break
            trees[i] = tr

        def span(ref):
            d = self.rn[ref[1]] if ref[0] == "rn" else self.pd[ref[1]]
            return tk(d["start"]), tk(d["end"]), d["allocs"]

        def apply(ref, plus):
            st, ed, allocs = span(ref)
            for node, res in allocs.items():
                if node in trees:
                    (trees[node].add if plus else trees[node].sub)(st, ed, res)
        ok = lambda: all(t.satisfied for t in trees.values())
        last = -1
        for x, ref in enumerate(order):
            if ok():
                break
            apply(ref, True)
            last = x
        if not ok():
            return None
        chosen = [order[last]] if last >= 0 else []
        for x in range(last - 1, -1, -1):
            apply(order[x], False)
            if not ok():
                apply(order[x], True)
                chosen.append(order[x])
        return chosen

    def run_job_p(self, idx, job, jinfo):
        """-> (reason, start, picks ascending node, preempted refs).  A job inside a reservation runs on that reservation's
        scheduler (its nodes, costs and job lists), or gets "Reservation Not Found"."""
        v = job.get("resv")
        if v is None:
            reason, start, picks, pre = self._run_job_p(idx, job, jinfo)
            if reason in (1, 2) and any(self.first_resv.get(i, INF) < self.now + job["L"] for i, _, _ in picks):
                reason = REASON_RESOURCE_RESERVED                                 # :6798-6806 (before the res_avail test)
                self.pd[idx]["reason"] = reason
            return reason, start, picks, pre
        if v not in self.resv_sched:
            self.pd[idx] = dict(qos=jinfo["qos"], qprio=jinfo["qprio"], prio=jinfo["prio"], start=0, end=0, allocs={}, reason=REASON_RESV_NOT_FOUND, nodes=[])
            return REASON_RESV_NOT_FOUND, 0, [], []
        saved = (self.nodes, self.parts, self.cost, self.node_jobs)
        try:
            self.nodes, self.parts, self.cost = self.resv_sched[v], [sorted(self.resv_sched[v])], [self.resv_cost[v]]
            self.node_jobs = self.resv_node_jobs.setdefault(v, {})
            return self._run_job_p(idx, dict(job, part=0), jinfo)
        finally:
            self.nodes, self.parts, self.cost, self.node_jobs = saved

    def _run_job_p(self, idx, job, jinfo):
        self.pd[idx] = dict(qos=jinfo["qos"], qprio=jinfo["qprio"], prio=jinfo["prio"], start=0, end=0, allocs={}, reason=None, nodes=[])
        r = self.try_schedule(job)
        if r is None:
            self.pd[idx]["reason"] = 2
            return 2, 0, [], []
        kind, picks = r
        pre = []
        if kind == "now":
            start = self.now
        else:
            chosen = self.try_preempt(job, jinfo, picks)                          # :6140-6143
            if chosen is not None:
                start, pre = self.now, chosen
            else:
                start = self.earliest_start(job, picks)
                if start is None:
                    self.pd[idx]["reason"] = 2
                    return 2, 0, [], []
        end = start + job["L"]
        p = job["part"]
        me = self.pd[idx]
        me.update(start=start, end=end, allocs={i: a for i, _, a in picks}, nodes=[i for i, _, _ in picks])
        for ref in pre:                                                           # :6779-6792, JobScheduler.h:645-670
            d = self.rn[ref[1]] if ref[0] == "rn" else self.pd[ref[1]]
            st = self.now if ref[0] == "rn" else d["start"]
            for node, res in d["allocs"].items():
                if node in self.cost[p]:
                    nd = self.nodes[node]
                    nd.release(st, d["end"], res)
                    self.cost[p][node] -= float(d["end"] - st) * ((float(res.cpu) / 256.0) / (float(nd.total.cpu) / 256.0))
            if ref[0] == "rn":
                for node in d["allocs"]:
                    if node in self.cost[p]:
                        self.node_jobs.get(node, {}).get(d["qos"], set()).discard(ref)
                if d["id"] not in self.preempting:
                    self.preempting.add(d["id"])
                    self.cancelled.append(d["id"])
            else:
                if d["reason"] == 0:                                              # is_scheduled() at that moment
                    for node in d["nodes"]:
                        if node in self.cost[p]:
                            self.node_jobs.get(node, {}).get(d["qos"], set()).discard(ref)
                d["reason"] = 7                                                   # "Preempted"
        for i, t, alloc in picks:                                                 # :6795
            nd = self.nodes[i]
            nd.commit(start, end, alloc)
            self.cost[p][i] += float(end - start) * ((float(alloc.cpu) / 256.0) / (float(nd.total.cpu) / 256.0))
            self.node_jobs.setdefault(i, {}).setdefault(jinfo["qos"], set()).add(("pd", idx))   # reason still empty (h:636-642)
        reason = 0
        if start != self.now:
            reason = 1
            if any(not res_le(alloc, self.nodes[i].avail0) for i, _, alloc in picks):
                reason = 2
        me["reason"] = reason
        return reason, start, sorted(picks, key=lambda x: x[0]), pre
