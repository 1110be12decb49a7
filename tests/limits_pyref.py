"""A second, independent restatement of the commit loop's run-limit admission in plain Python — dicts keyed like the
reference's maps, no shared code with oracle/limits_oracle.hpp — used only to cross-check the C++ oracle on random
inputs (tests/test_run_limits.py).  Follows AccountMetaContainer.cpp:180-224 (CheckAndMallocMetaResource), :891-1028
(CheckRunLimits_), :508-688 (entity checks), :345-365,1030-1050 (CheckTres_ / CheckGres_), :1067-1124 (DoMallocResource_).
The allocation view of a job comes from its placements (ResourceV3::View, PublicHeader.cpp:946-952)."""
import numpy as np

from cranesched_amd import limits as lm

NONE = lm.LIM_NONE


def _view(cpu, mem, names, classes):
    return {"cpu": int(cpu), "mem": int(mem), "gres": {n: {"total": t, "spec": dict(classes.get(n, {}))} for n, t in names.items()}}


def _tres_view(t, layout):
    gres = {}
    for n in range(4):
        if int(t["name_mask"]) >> n & 1:
            gres[n] = {"total": int(t["name_total"][n]), "spec": {}}
    for g in range(len(layout.class_name)):
        n = layout.class_name[g]
        if (int(t["class_mask"]) >> g & 1) and n in gres:
            gres[n]["spec"][g] = int(t["class_count"][g])
    return {"cpu": int(t["cpu_raw"]), "mem": int(t["mem"]), "gres": gres}


def _usage_meta(u, layout):
    gres = {}
    for n in range(4):
        if int(u["name_total"][n]):
            gres[n] = {"total": int(u["name_total"][n]), "spec": {}}
    for g in range(len(layout.class_name)):
        if int(u["class_count"][g]):
            gres.setdefault(layout.class_name[g], {"total": 0, "spec": {}})["spec"][g] = int(u["class_count"][g])
    return {"res": {"cpu": int(u["cpu_raw"]), "mem": int(u["mem"]), "gres": gres}, "jobs": int(u["jobs_count"]), "wall": int(u["wall_sec"])}


def _add_view(a, b):
    a["cpu"] += b["cpu"]; a["mem"] += b["mem"]
    for n, gc in b["gres"].items():
        d = a["gres"].setdefault(n, {"total": 0, "spec": {}})
        d["total"] += gc["total"]
        for t, c in gc["spec"].items():
            d["spec"][t] = d["spec"].get(t, 0) + c


def _copy_view(v):
    return {"cpu": v["cpu"], "mem": v["mem"], "gres": {n: {"total": g["total"], "spec": dict(g["spec"])} for n, g in v["gres"].items()}}


def _check_gres(req, total):                       # :1030-1050, ascending name / type = the canonical order
    for n in sorted(req):
        if n not in total:
            return True
        if req[n]["total"] > total[n]["total"]:
            return False
        for t in sorted(req[n]["spec"]):
            if t not in total[n]["spec"]:
                return True
            if req[n]["spec"][t] > total[n]["spec"][t]:
                return False
    return True


def _check_tres(req, total, prefix):               # :345-360
    if req["cpu"] > total["cpu"]:
        return 5 + prefix
    if req["mem"] > total["mem"]:
        return 6 + prefix
    if not _check_gres(req["gres"], total["gres"]):
        return 7 + prefix
    return 0


def _unlimited(v):                                 # :362-365
    return v["cpu"] == lm.UNLIMITED_CPU_RAW and v["mem"] == lm.MAX_JOB_MEMORY and not v["gres"]


def run(layout, t, lj, pl):
    """Returns (reason[J] u8, admitted).  layout: abi.GresLayout, t: LimitTables, lj: LimitJobs, pl: abi.Placements"""
    Q, Pn = t.num_qos, t.num_partitions
    qos = [{"jpu": int(q["max_jobs_per_user"]), "jpa": int(q["max_jobs_per_account"]), "jobs": int(q["max_jobs"]),
            "cpu_x": int(q["max_cpus_per_user_raw"]), "wall": int(q["max_wall_sec"]),
            "tres": _tres_view(q["max_tres"], layout), "tpu": _tres_view(q["max_tres_per_user"], layout),
            "tpa": _tres_view(q["max_tres_per_account"], layout)} for q in t.qos]
    plim = [{"jobs": int(p["max_jobs"]), "wall": int(p["max_wall_sec"]), "tres": _tres_view(p["max_tres"], layout)} for p in t.part_limits]

    def table(usage, exists, n):
        d = {}
        for i in range(n):
            if exists is None or exists[i]:
                d[i] = _usage_meta(usage[i], layout) if usage is not None else {"res": {"cpu": 0, "mem": 0, "gres": {}}, "jobs": 0, "wall": 0}
        return d
    uq = table(t.user_qos, t.user_qos_exists, t.num_users * Q)
    up = table(t.user_part, t.user_part_exists, t.num_user_accts * Pn)
    aq = table(t.acct_qos, t.acct_qos_exists, t.num_accounts * Q)
    ap = table(t.acct_part, t.acct_part_exists, t.num_accounts * Pn)
    qg = table(t.qos_usage, None, Q)

    def entity_qos(tab, key, q, is_user, alloc, tl):      # :508-540
        if key not in tab:
            return 1
        val = tab[key]
        use = _copy_view(alloc); _add_view(use, val["res"])
        if is_user:
            if use["cpu"] > q["cpu_x"]: return 2
            if val["jobs"] + 1 > q["jpu"]: return 3
            if q["wall"] > 0 and val["wall"] + tl > q["wall"]: return 4
            return _check_tres(use, q["tpu"], 0)
        if val["jobs"] + 1 > q["jpa"]: return 3
        if q["wall"] > 0 and val["wall"] + tl > q["wall"]: return 4
        return _check_tres(use, q["tpa"], 0)

    def entity_part(tab, key, lim_id, q, is_user, alloc, tl):   # :542-670
        if lim_id == NONE:
            return 0
        pl_ = plim[lim_id]
        if key not in tab:
            return 8
        val = tab[key]
        if (q["jpu"] if is_user else q["jpa"]) == lm.UNLIMITED_JOBS and val["jobs"] + 1 > pl_["jobs"]:
            return 9 if is_user else 11
        if q["wall"] == 0 and pl_["wall"] > 0 and val["wall"] + tl > pl_["wall"]:
            return 10 if is_user else 12
        if _unlimited(q["tpu"] if is_user else q["tpa"]):
            use = _copy_view(alloc); _add_view(use, val["res"])
            return _check_tres(use, pl_["tres"], 8)
        return 0

    J = lj.num_jobs
    reason = np.zeros(J, np.uint8)
    adm = 0
    po = pl.place_offsets
    for i in range(J):
        s = int(lj.select_index[i]) if lj.select_index is not None else i
        if pl.reason[s] != 0 or (lj.skip is not None and lj.skip[i]):
            reason[i] = 255
            continue
        cpu = mem = 0
        names, classes = {}, {}
        for r in range(int(po[s]), int(po[s + 1])):
            if pl.node_idx[r] == 0xFFFFFFFF:
                continue
            cpu += int(pl.cpu_raw[r]); mem += int(pl.mem[r])
            for g in range(len(layout.class_name)):
                c = bin(int(pl.gres[r]) & layout.class_mask(g)).count("1")
                if c:
                    n = layout.class_name[g]
                    names[n] = names.get(n, 0) + c
                    classes.setdefault(n, {})[g] = classes.get(n, {}).get(g, 0) + c
        alloc = _view(cpu, mem, names, classes)
        u, x, a0, qi, p, tl = int(lj.user[i]), int(lj.user_acct[i]), int(lj.account[i]), int(lj.qos[i]), int(lj.partition[i]), int(lj.time_limit_sec[i])
        q = qos[qi]
        chain = []
        a = a0
        while a != NONE:
            chain.append(a); a = int(t.acct_parent[a])
        upl = int(t.user_part_limit[x * Pn + p]) if t.user_part_limit is not None else NONE
        r = entity_qos(uq, u * Q + qi, q, True, alloc, tl) or entity_part(up, x * Pn + p, upl, q, True, alloc, tl)
        if not r:
            for a in chain:
                apl = int(t.acct_part_limit[a * Pn + p]) if t.acct_part_limit is not None else NONE
                r = entity_qos(aq, a * Q + qi, q, False, alloc, tl) or entity_part(ap, a * Pn + p, apl, q, False, alloc, tl)
                if r:
                    break
        if not r:                                        # :985-1025
            val = qg[qi]
            use = _copy_view(alloc); _add_view(use, val["res"])
            if val["jobs"] + 1 > q["jobs"]: r = 3
            elif q["wall"] > 0 and val["wall"] + tl > q["wall"]: r = 4
            else: r = _check_tres(use, q["tres"], 0)
        reason[i] = r
        if r:
            continue
        adm += 1                                         # :1067-1124
        for tab, key in [(uq, u * Q + qi), (up, x * Pn + p), (qg, qi)] + [(aq, a * Q + qi) for a in chain] + [(ap, a * Pn + p) for a in chain]:
            m = tab.setdefault(key, {"res": {"cpu": 0, "mem": 0, "gres": {}}, "jobs": 0, "wall": 0})
            _add_view(m["res"], alloc); m["jobs"] += 1; m["wall"] += tl
    return reason, adm
