"""JobInCtld::SchedulePendingSteps once more, in plain Python, written from the reference alone
(/root/reference/src/CraneCtld/CtldPublicDefs.cpp:2038-2159) on top of the resource algebra and the libstdc++ heap of
tests/select_pyref.py (GetFeasibleResourceInNode, `-=`, `+=`, std::priority_queue as bits/stl_heap.h builds it); it shares no
code with oracle/steps_oracle.hpp.  One canonicalisation, the oracle's: the job's nodes are walked in the order given (the
reference walks an unordered_map).  Test infrastructure only."""
from __future__ import annotations

from tests import select_pyref as pr


def schedule_pending_steps(nodes, avail, steps, types_of):
    """nodes: the job's node indices; avail: their ResourceInNodeV3 (step_res_avail_, pr.Res, updated in place);
    steps: dicts node_view / task_view (pr.Req), k, ntasks, tmin, tmax, incl, excl (sets of node indices), in queue order.
    Returns per step None (not scheduled: it and everything behind it stays pending, :2104-2106) or
    (places, tasks): places = [(node, ntasks_on_node, step_alloc_res of the node)] in pop order, tasks = [(node, task_res)]
    by task id."""
    outs = [None] * len(steps)
    for si, step in enumerate(steps):
        cand = []                                                      # std::priority_queue<NodeInfo>, top = fewest tasks (:2056-2063)
        sum_ntasks = 0
        for pos, node in enumerate(nodes):                             # :2066-2102
            if node in step["excl"]:
                continue
            if step["incl"] and node not in step["incl"]:
                continue
            fr = pr.feasible(step["node_view"], avail[pos], types_of)
            if fr is None:
                continue
            res_avail = avail[pos].copy()
            pr.res_sub(res_avail, fr)
            n_on = 0
            while n_on < step["tmax"]:
                fr = pr.feasible(step["task_view"], res_avail, types_of)
                if fr is None:
                    break
                n_on += 1
                pr.res_sub(res_avail, fr)
            if n_on < step["tmin"]:
                continue
            pr.heap_push(cand, (n_on, pos))
            sum_ntasks += n_on
            if len(cand) > step["k"]:
                sum_ntasks -= cand[0][0]
                pr.heap_pop(cand)
            if len(cand) == step["k"] and sum_ntasks >= step["ntasks"]:
                break
        if len(cand) < step["k"] or sum_ntasks < step["ntasks"]:       # :2104-2106
            break
        rest = step["ntasks"] - step["k"]                              # :2107
        places, tasks = [], []
        while cand:                                                    # :2109-2128
            n_cap, pos = cand[0]
            res_avail = avail[pos]
            node_sum = pr.Res()
            fr = pr.feasible(step["node_view"], res_avail, types_of)
            pr.res_sub(res_avail, fr)
            pr.res_add(node_sum, fr)
            n_on = min(rest, n_cap - 1) + 1
            for _ in range(n_on):
                fr = pr.feasible(step["task_view"], res_avail, types_of)
                pr.res_sub(res_avail, fr)
                tasks.append((nodes[pos], fr))
                pr.res_add(node_sum, fr)
            rest -= n_on - 1
            places.append((nodes[pos], n_on, node_sum))
            pr.heap_pop(cand)
        outs[si] = (places, tasks)
    return outs
