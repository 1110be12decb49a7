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


# ---------------------------------------------------------------------------------------------------------------------
# General B (round 5, first design — built, bit-exact on the GPU, measured at 4 800 cycles per job and REPLACED by the pool rule at the
# end of this file: profiles/r05_window_hypotheticals_*.txt): ONE exchange for a WINDOW of up to B consecutive one-node jobs.  Per job i of the window every wave publishes i + 1 entries, all computed on its rows AS THEY ARE at
# the window's start:
#   E_i^0        "I have won nothing earlier in this window": its start-now argmin A, else its res_total argmin T (a
#                backfill), else nothing;
#   E_i^(1+j)    j < i, "I have won exactly job j (with E_j^0)": only the row R_j of E_j^0 differs from the tile (costlier;
#                emptier if job j starts there), so this is E_i^0 unless E_i^0 sits on R_j; then it is the lesser of the
#                wave's SECOND best start-now row and the changed R_j if job i still fits it.  If no start-now row is left
#                the entry falls back to the T argmin, which is only a LOWER BOUND when it sits on R_j itself.
# All waves resolve the jobs in order from the same granules: a wave's entry for job i is E_i^0 while it has won nothing,
# E_i^(1+j) after one win (exact unless flagged), and E_i^0 as a mere lower bound after two or more (a touched row only gets
# costlier and emptier: JobScheduler.h:526-538,567-575; a hypothesis "won exactly job j2" is NOT one then: the wave may have
# won j2 on another row than the one that hypothesis changes).  The least entry decides job i if it is exact; a least entry that is only a
# lower bound CLOSES the window in front of job i, which opens the next one.  A window always resolves its first job.
# ---------------------------------------------------------------------------------------------------------------------
def fits_total(node, job):
    return node["total"] >= job["cpu"] and node["mtotal"] >= job["mem"]


def best_total(nodes, idx, job):
    b = None
    for i in idx:
        if fits_total(nodes[i], job) and (b is None or (nodes[i]["cost"], i) < b):
            b = (nodes[i]["cost"], i)
    return b


def bump(node, job):
    node["cost"] += job["L"] * (job["cpu"] / node["total"])


def sequential_bf(nodes, jobs):
    """start now on the least (cost, index) that fits; else reserve (cost only) on the least whose TOTAL fits; else nothing."""
    out = []
    for job in jobs:
        b = best(nodes, range(len(nodes)), job)
        if b is not None:
            out.append(("start", b[1])); place(nodes[b[1]], job)
            continue
        t = best_total(nodes, range(len(nodes)), job)
        if t is not None:
            out.append(("backfill", t[1])); bump(nodes[t[1]], job)
        else:
            out.append(("none", None))
    return out


def best2(nodes, idx, job):
    """the two least (cost, index) among the rows that fit, on DISTINCT rows"""
    c = sorted((nodes[i]["cost"], i) for i in idx if fits(nodes[i], job))
    return (c[0] if c else None), (c[1] if len(c) > 1 else None)


def windowed(nodes, jobs, W, B, owner, stats):
    rows = [[i for i in range(len(nodes)) if owner(i) == w] for w in range(W)]
    out, j0 = [], 0
    while j0 < len(jobs):
        n = min(B, len(jobs) - j0)
        # ---- what every wave publishes: E[w][i][h] = (kind 0 A / 1 T, cost, node, lower_bound) or None ----
        E = []
        for w in range(W):
            Ew, base = [], []
            for i in range(n):
                job = jobs[j0 + i]
                a1, a2 = best2(nodes, rows[w], job)
                t1 = best_total(nodes, rows[w], job)
                e0 = (0,) + a1 + (False,) if a1 else ((1,) + t1 + (False,) if t1 else None)
                ent = [e0]
                for j in range(i):
                    ej = base[j]
                    if ej is None:                   # the wave cannot have won job j
                        ent.append(e0); continue
                    Rj = ej[2]
                    hyp = dict(nodes[Rj])
                    place(hyp, jobs[j0 + j]) if ej[0] == 0 else bump(hyp, jobs[j0 + j])
                    if a1 is None: ah = None
                    elif a1[1] != Rj: ah = a1
                    else:
                        r = (hyp["cost"], Rj) if fits(hyp, job) else None
                        ah = min([x for x in (a2, r) if x is not None], default=None)
                    if ah is not None: ent.append((0,) + ah + (False,))
                    elif t1 is not None: ent.append((1,) + t1 + (t1[1] == Rj,))
                    else: ent.append(None)
                Ew.append(ent); base.append(e0)
            E.append(Ew)
        # ---- what every wave computes from them ----
        wins = [[] for _ in range(W)]
        done = 0
        for i in range(n):
            cand = []
            for w in range(W):
                e = E[w][i][1 + wins[w][0]] if len(wins[w]) == 1 else E[w][i][0]   # (two wins: no published hypothesis covers the wave; E^0 bounds it from below)
                if e is None: continue
                lb = e[3] or len(wins[w]) >= 2
                cand.append((e[:3], w, lb))
            if not cand:
                out.append(("none", None)); done += 1
                continue
            key, w, lb = min(cand)
            if lb:
                break
            job = jobs[j0 + i]
            if key[0] == 0:
                out.append(("start", key[2])); place(nodes[key[2]], job)
            else:
                out.append(("backfill", key[2])); bump(nodes[key[2]], job)
            wins[w].append(i)
            done += 1
        assert done >= 1
        stats["windows"] += 1; stats["jobs"] += done
        j0 += done
    return out


@pytest.mark.parametrize("seed", range(40))
def test_one_exchange_resolves_a_window_of_jobs_like_the_sequential_rule(seed):
    rng = random.Random(1000 + seed)
    N, W, B = rng.choice([5, 16, 64, 257, 600]), rng.choice([1, 2, 4, 8, 16]), rng.choice([2, 3, 4, 8])
    interleaved = rng.random() < 0.5           # node -> wave: i % W (consecutive nodes on different waves) or blocks of consecutive nodes
    per = (N + W - 1) // W
    owner = (lambda i: i % W) if interleaved else (lambda i: i // per)
    def cluster():
        r = random.Random(seed + 1)
        out = []
        for _ in range(N):
            tot = r.choice([16.0, 64.0, 64.0])
            out.append(dict(cost=r.choice([0.0, 0.0, r.random() * 100]), total=tot, mtotal=256, cpu=r.randrange(0, int(tot) + 1), mem=r.randrange(0, 257)))
        return out
    jobs = [dict(cpu=rng.choice([1, 1, 2, 4, 8, 32, 70]), mem=rng.choice([1, 2, 16, 64, 300]), L=rng.choice([600, 600, 1200, 3600])) for _ in range(rng.randrange(1, 500))]
    stats = dict(windows=0, jobs=0)
    a, b = sequential_bf(cluster(), jobs), windowed(cluster(), jobs, W, B, owner, stats)
    assert a == b
    assert stats["jobs"] == len(jobs)


def test_window_fill_on_a_cold_and_on_a_warm_cluster():
    """How full the windows get (the model's own statistics, B = 4, 64 waves).  Cold cluster (all costs 0): ties go to the lowest node
    index, consecutive jobs take consecutive nodes; blocks of consecutive nodes per wave make one wave win again and again (its
    third win in a window is only a lower bound: ~2 jobs per window), node i on wave i % W resolves every window whole."""
    N, W, B = 4096, 64, 4
    jobs = [dict(cpu=64, mem=1, L=600) for _ in range(1024)]
    def cluster():
        return [dict(cost=0.0, total=64.0, mtotal=256, cpu=64, mem=256) for _ in range(N)]
    s_blk, s_int = dict(windows=0, jobs=0), dict(windows=0, jobs=0)
    a = windowed(cluster(), jobs, W, B, lambda i: i // (N // W), s_blk)
    b = windowed(cluster(), jobs, W, B, lambda i: i % W, s_int)
    assert a == b == sequential_bf(cluster(), jobs)
    assert s_int["windows"] == 1024 // B
    assert s_blk["windows"] == 1024 // 2
    # warm: random costs, small jobs — a window closes early only when one wave holds three of its four winners
    rng = random.Random(5)
    def warm():
        r = random.Random(6)
        return [dict(cost=r.random() * 1000, total=64.0, mtotal=256, cpu=64, mem=256) for _ in range(N)]
    jobs = [dict(cpu=rng.choice([1, 2, 4, 8]), mem=rng.choice([2, 4, 8, 16]), L=600 * rng.randrange(1, 25)) for _ in range(4000)]
    s = dict(windows=0, jobs=0)
    assert windowed(warm(), jobs, W, B, lambda i: i % W, s) == sequential_bf(warm(), jobs)
    assert s["jobs"] / s["windows"] > 3.9


def test_a_wave_that_won_twice_is_bounded_by_its_untouched_entry_only():
    """Small clusters on two waves, jobs of very different sizes: a wave wins job j1 (small) and job j2 (big) whose best rows coincide,
    the second one on its SECOND best row.  The hypothesis "won exactly j2" then describes a tile that never existed and may lie ABOVE
    the wave's true entry (157 of 30 000 such cases resolve wrongly with it); only E_i^0 bounds a two-win wave from below."""
    for seed in range(4000):
        rng = random.Random(seed)
        N, W, B = rng.choice([3, 4, 6]), 2, 4
        def cluster():
            r = random.Random(seed + 7)
            return [dict(cost=r.choice([0.0, 1.0, 5.0, r.random() * 10]), total=64.0, mtotal=256, cpu=r.randrange(8, 65), mem=256) for _ in range(N)]
        jobs = [dict(cpu=rng.choice([1, 2, 16, 32]), mem=1, L=rng.choice([60, 600, 6000])) for _ in range(8)]
        assert windowed(cluster(), jobs, W, B, lambda i: i % W, dict(windows=0, jobs=0)) == sequential_bf(cluster(), jobs), seed


# ---------------------------------------------------------------------------------------------------------------------
# The POOL rule (round 5, what k_wide runs: wide_kernel.inc, "A WINDOW OF JOBS PER EXCHANGE").  Every job moves exactly one row,
# and only towards "costlier and emptier" — so the next several decisions fall among the few cheapest rows.  A window
# exchanges the candidate POOL once: every wave publishes its M cheapest rows WITH THEIR STATE and the key of its next row
# (the bound).  Every wave then decides job after job on its copy of the pool, with no further communication: the least
# (cost, index) among the pool rows the job fits, applied to the copy (and by the owner to its tile).  A decision is exact
# while its key lies below every wave's bound: no row outside the pool can come before it (their keys are >= the bounds and
# do not move).  The first job that finds no pool row, or whose key is not below the bound, closes the window and goes
# through the single-job exchange.
# ---------------------------------------------------------------------------------------------------------------------
def pooled(nodes, jobs, W, M, JMAX, owner, stats):
    rows = [[i for i in range(len(nodes)) if owner(i) == w] for w in range(W)]
    out, j = [], 0
    def single(job):
        out.extend(sequential_bf(nodes, [job]))
    while j < len(jobs):
        pool, bound = {}, None
        for w in range(W):
            srt = sorted((nodes[i]["cost"], i) for i in rows[w])
            for c, i in srt[:M]:
                pool[i] = dict(nodes[i])
            if len(srt) > M and (bound is None or srt[M] < bound): bound = srt[M]
        done = 0
        while j < len(jobs) and done < JMAX:
            job = jobs[j]
            cand = [(r["cost"], i) for i, r in pool.items() if fits(r, job)]
            if not cand: break
            key = min(cand)
            if bound is not None and not key < bound: break
            place(pool[key[1]], job)           # every wave: its copy of the row
            place(nodes[key[1]], job)          # the owner: its tile
            out.append(("start", key[1]))
            j += 1; done += 1
        stats["windows"] += 1; stats["jobs"] += done
        if done == 0:
            single(jobs[j]); j += 1; stats["single"] += 1
    return out


@pytest.mark.parametrize("seed", range(120))
def test_jobs_decided_on_an_exchanged_pool_follow_the_sequential_rule(seed):
    rng = random.Random(7000 + seed)
    N, W, M = rng.choice([5, 16, 64, 257, 600]), rng.choice([1, 2, 4, 8, 16]), rng.choice([1, 2, 3])
    per = (N + W - 1) // W
    owner = (lambda i: i % W) if rng.random() < 0.5 else (lambda i: i // per)
    def cluster():
        r = random.Random(seed + 1)
        return [dict(cost=r.choice([0.0, 0.0, r.random() * 100]), total=r.choice([16.0, 64.0]), mtotal=256, cpu=r.randrange(0, 17), mem=r.randrange(0, 257)) for _ in range(N)]
    jobs = [dict(cpu=rng.choice([1, 1, 2, 4, 8, 32]), mem=rng.choice([1, 2, 16, 64, 300]), L=rng.choice([600, 1200, 3600])) for _ in range(rng.randrange(1, 400))]
    st = dict(windows=0, jobs=0, single=0)
    assert pooled(cluster(), jobs, W, M, rng.choice([4, 16]), owner, st) == sequential_bf(cluster(), jobs)


def test_pool_windows_fill_on_a_partition_like_c5():
    """8 192 nodes x 64 cores on 64 waves, jobs of 1..8 cpus (the first 6 000 of a C5-like queue, cold start included): two rows per wave
    fill a window of 16 almost always; one row per wave already decides ~11 jobs per exchange."""
    rng = random.Random(1)
    jobs = [dict(cpu=c, mem=2 * c, L=675 * rng.randrange(1, 33)) for c in (rng.choice([1, 2, 4, 8]) for _ in range(6000))]
    fill = {}
    for M in (1, 2):
        nodes = [dict(cost=0.0, total=64.0, mtotal=256, cpu=64, mem=256) for _ in range(8192)]
        st = dict(windows=0, jobs=0, single=0)
        pooled(nodes, jobs, 64, M, 16, lambda i: i % 64, st)
        fill[M] = st["jobs"] / st["windows"]
    assert fill[1] > 9 and fill[2] > 15
