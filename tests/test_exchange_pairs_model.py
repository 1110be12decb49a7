"""A model of DESIGN.md 10, item 1(b) — ONE exchange for TWO consecutive one-node, start-now jobs — checked against the
sequential rule it has to reproduce (min cost first, ties by node index: MinCpuTimeRatioFirst; JobScheduler.cpp:6188-6333).
Not product code: it pins down what every wave has to publish so that all waves resolve both jobs from the same granules.

The node rows are spread over W waves.  For the pair (j, j + 1) a wave publishes
  A   its cheapest node that can start job j now,
  B0  its cheapest node for job j + 1 on its rows as they are (right if the wave does NOT win job j),
  B1  the same with the row of A changed the way job j would change it (right if it DOES win job j)
and every wave then knows: job j goes to the least A (w* its wave), job j + 1 to the least of {B0 of the others, B1 of w*}.
Only w*'s row changes between the two jobs, a cost only grows and a free resource only shrinks, so nothing else can move."""
import random

import pytest


def fits(node, job):
    return node["cpu"] >= job["cpu"] and node["mem"] >= job["mem"]


def place(node, job):
    node["cost"] += job["L"] * (job["cpu"] / node["total"])      # UpdateCost: ratio first, then x seconds (h:47-53)
    node["cpu"] -= job["cpu"]; node["mem"] -= job["mem"]


def best(nodes, idx, job, override=None):
    """Least (cost, index) among the nodes idx that fit; `override`: (index, node) to look at instead of nodes[index]."""
    b = None
    for i in idx:
        n = override[1] if override and override[0] == i else nodes[i]
        if fits(n, job) and (b is None or (n["cost"], i) < b):
            b = (n["cost"], i)
    return b


def sequential(nodes, jobs):
    out = []
    for job in jobs:
        b = best(nodes, range(len(nodes)), job)
        out.append(None if b is None else b[1])
        if b is not None:
            place(nodes[b[1]], job)
    return out


def paired(nodes, jobs, W):
    rows = [list(range(w, len(nodes), W)) for w in range(W)]      # wave w holds the nodes w, w + W, ...
    out, j = [], 0
    while j < len(jobs):
        j0 = jobs[j]
        j1 = jobs[j + 1] if j + 1 < len(jobs) else None
        A, B0, B1 = [], [], []
        for w in range(W):
            a = best(nodes, rows[w], j0)
            A.append(a)
            if j1 is None:
                continue
            B0.append(best(nodes, rows[w], j1))
            if a is None:
                B1.append(None)
            else:
                hyp = dict(nodes[a[1]]); place(hyp, j0)
                B1.append(best(nodes, rows[w], j1, override=(a[1], hyp)))
        # ---- what every wave computes from the 3 W published entries ----
        cand = [(a, w) for w, a in enumerate(A) if a is not None]
        if not cand:            # nobody can start job j now: the pair protocol does not apply (backfill / "Resource"); one job resolved
            out.append(None); j += 1
            continue
        a, wstar = min(cand)
        out.append(a[1]); place(nodes[a[1]], j0)
        if j1 is None:
            j += 1
            continue
        c1 = [b for w, b in enumerate(B0) if w != wstar and b is not None] + ([B1[wstar]] if B1[wstar] is not None else [])
        if c1:
            b = min(c1)
            out.append(b[1]); place(nodes[b[1]], j1)
        else:
            out.append(None)
        j += 2
    return out


@pytest.mark.parametrize("seed", range(60))
def test_one_exchange_resolves_two_jobs_like_the_sequential_rule(seed):
    rng = random.Random(seed)
    N, W = rng.choice([5, 16, 64, 257]), rng.choice([1, 2, 4, 8])
    def cluster():
        r = random.Random(seed + 1)
        return [dict(cost=r.choice([0.0, 0.0, r.random() * 100]), total=64.0, cpu=r.randrange(0, 65), mem=r.randrange(0, 257)) for _ in range(N)]
    jobs = [dict(cpu=rng.choice([1, 1, 2, 4, 8, 32]), mem=rng.choice([1, 2, 16, 64]), L=rng.choice([600, 600, 1200, 3600])) for _ in range(rng.randrange(1, 400))]
    a, b = sequential(cluster(), jobs), paired(cluster(), jobs, W)
    assert a == b
