"""MultiFactorPriority once more, in plain Python, written from the reference alone
(/root/reference/src/CraneCtld/JobScheduler.cpp:7606-7819) and sharing no code with oracle/prio_oracle.hpp: a second,
independent restatement, as tests/select_pyref.py is for the selection path and tests/limits_pyref.py for the run limits.
Python floats are IEEE doubles and every expression below keeps the reference's operand types and evaluation order
(`1.0 * (u32 - u32) / (u32 - u32)`: the product is a double before the division; the five weighted factors are summed left
to right), so the priorities can be compared as bit patterns.  Test infrastructure only."""
from __future__ import annotations

U64 = (1 << 64) - 1
U32_MAX = (1 << 32) - 1
DBL_MAX = 1.7976931348623157e308


def bounds(now, max_age, pending, running):
    """CalculateFactorBound_ (:7633-7752).  pending: dicts with submit, qos, part, nodes, cpu_raw, mem, account;
    running: dicts with start, qos, part, nodes, cpu_raw, mem, account."""
    b = dict(age_max=0, age_min=U64, qos_max=0, qos_min=U32_MAX, part_max=0, part_min=U32_MAX, nodes_max=0, nodes_min=U32_MAX,
             mem_max=0, mem_min=U64, cpus_max=0.0, cpus_min=DBL_MAX, sv_max=0.0, sv_min=float(U32_MAX), acc={})
    for j in pending:                                                  # :7663-7691
        age = min((now - j["submit"]) & U64, max_age)                  # uint64_t age = ToInt64Seconds(...); min with MaxAge
        b["acc"][j["account"]] = 0.0
        b["age_min"] = min(age, b["age_min"]); b["age_max"] = max(age, b["age_max"])
        b["nodes_min"] = min(j["nodes"], b["nodes_min"]); b["nodes_max"] = max(j["nodes"], b["nodes_max"])
        b["mem_min"] = min(j["mem"], b["mem_min"]); b["mem_max"] = max(j["mem"], b["mem_max"])
        cpus = j["cpu_raw"] / 256.0
        b["cpus_min"] = min(cpus, b["cpus_min"]); b["cpus_max"] = max(cpus, b["cpus_max"])
        b["qos_min"] = min(j["qos"], b["qos_min"]); b["qos_max"] = max(j["qos"], b["qos_max"])
        b["part_min"] = min(j["part"], b["part_min"]); b["part_max"] = max(j["part"], b["part_max"])
    for r in running:                                                  # :7693-7713
        b["nodes_min"] = min(r["nodes"], b["nodes_min"]); b["nodes_max"] = max(r["nodes"], b["nodes_max"])
        b["mem_min"] = min(r["mem"], b["mem_min"]); b["mem_max"] = max(r["mem"], b["mem_max"])
        cpus = r["cpu_raw"] / 256.0
        b["cpus_min"] = min(cpus, b["cpus_min"]); b["cpus_max"] = max(cpus, b["cpus_max"])
        b["qos_min"] = min(r["qos"], b["qos_min"]); b["qos_max"] = max(r["qos"], b["qos_max"])
        b["part_min"] = min(r["part"], b["part_min"]); b["part_max"] = max(r["part"], b["part_max"])
    for r in running:                                                  # :7715-7745
        sv = 0.0
        if b["cpus_max"] > b["cpus_min"]:
            sv += 1.0 * (r["cpu_raw"] / 256.0 - b["cpus_min"]) / (b["cpus_max"] - b["cpus_min"])
        else:
            sv += 1.0
        if b["nodes_max"] > b["nodes_min"]:
            sv += 1.0 * ((r["nodes"] - b["nodes_min"]) & U32_MAX) / (b["nodes_max"] - b["nodes_min"])
        else:
            sv += 1.0
        if b["mem_max"] > b["mem_min"]:
            sv += 1.0 * float(r["mem"] - b["mem_min"]) / float(b["mem_max"] - b["mem_min"])
        else:
            sv += 1.0
        run_time = (now - r["start"]) & U64
        b["acc"][r["account"]] = b["acc"].get(r["account"], 0.0) + sv * float(run_time)
    for v in b["acc"].values():                                        # :7747-7751
        b["sv_min"] = min(v, b["sv_min"]); b["sv_max"] = max(v, b["sv_max"])
    return b


def priority(now, cfg, b, j):
    """CalculatePriority_ (:7754-7817); cfg: max_age, w_age, w_fair, w_size, w_part, w_qos, favor_small."""
    age = min((now - j["submit"]) & U64, cfg["max_age"])
    qos_f = age_f = part_f = size_f = fair_f = 0.0
    if b["age_max"] > b["age_min"]:
        age_f = 1.0 * float(age - b["age_min"]) / float(b["age_max"] - b["age_min"])
    if b["qos_max"] > b["qos_min"]:
        qos_f = 1.0 * ((j["qos"] - b["qos_min"]) & U32_MAX) / (b["qos_max"] - b["qos_min"])
    if b["part_max"] > b["part_min"]:
        part_f = 1.0 * ((j["part"] - b["part_min"]) & U32_MAX) / (b["part_max"] - b["part_min"])
    if b["cpus_max"] > b["cpus_min"]:
        size_f += 1.0 * (j["cpu_raw"] / 256.0 - b["cpus_min"]) / (b["cpus_max"] - b["cpus_min"])
    if b["nodes_max"] > b["nodes_min"]:
        size_f += 1.0 * ((j["nodes"] - b["nodes_min"]) & U32_MAX) / (b["nodes_max"] - b["nodes_min"])
    if b["mem_max"] > b["mem_min"]:
        size_f += 1.0 * float(j["mem"] - b["mem_min"]) / float(b["mem_max"] - b["mem_min"])
    if cfg["favor_small"]:
        size_f = 1.0 - size_f / 3
    else:
        size_f /= 3.0
    if b["sv_max"] > b["sv_min"]:
        fair_f = 1.0 - (b["acc"][j["account"]] - b["sv_min"]) / (b["sv_max"] - b["sv_min"])
    return cfg["w_age"] * age_f + cfg["w_part"] * part_f + cfg["w_size"] * size_f + cfg["w_fair"] * fair_f + cfg["w_qos"] * qos_f


def ordered(now, cfg, pending, running):
    """GetOrderedJobPtrVec (:7606-7631): cached non-zero priorities are kept (:7616); descending priority, and — where the
    reference's unstable sort leaves the order open — ascending queue index (the canonical order of SURVEY 8f-2)."""
    b = bounds(now, cfg["max_age"], pending, running)
    prio = [j["cached"] if j.get("cached", 0.0) != 0.0 else priority(now, cfg, b, j) for j in pending]
    order = sorted(range(len(pending)), key=lambda i: (-prio[i], i))
    return order, prio
