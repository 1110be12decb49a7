"""Bit-level models (plain Python) of two pieces of device logic whose correctness argument is not obvious from the
code, checked against brute force on random inputs.  They pin the ALGORITHMS; the kernels themselves are checked
against the oracle under `-m gpu`.

1. `next_fit_regs` (cranesched_amd/csrc/select_kernels.hip): the lane-parallel earliest-start search must equal the
   reference's walk over the time map (EarliestStartSubsetSelector, JobScheduler.h:792-865, on one node).
2. the three-valued CheckGres_ of `k_par_eval` (cranesched_amd/csrc/limits_kernels.hip): "certainly passes" /
   "certainly fails" over an interval of usage vectors must hold for EVERY vector of the interval
   (AccountMetaContainer.cpp:1030-1050 evaluated exhaustively)."""
import itertools

import numpy as np
import pytest

INF = 1 << 62


def next_fit_walk(t, sat, L, t0):
    """the sequential form (the previous next_fit_regs, itself a restatement of the run walk)"""
    n = len(t)
    idx = max(i for i in range(n) if t[i] <= t0)
    s = t0
    while True:
        if not sat[idx]:
            above = [i for i in range(idx + 1, n) if sat[i]]
            if not above:
                return INF
            idx = above[0]
            s = t[idx]
        unsat = [i for i in range(idx + 1, n) if not sat[i]]
        if not unsat:
            return s
        ue = unsat[0]
        if t[ue] - s >= L:
            return s
        idx = ue


def next_fit_lanes(t, sat, L, t0):
    """what every lane computes, then the lowest qualifying lane wins"""
    n = len(t)
    idx0 = sum(1 for i in range(n) if t[i] <= t0) - 1
    ok = []
    for lane in range(n):
        here = sat[lane]
        below = lane > 0 and sat[lane - 1]
        starts = here and (lane == idx0 or (lane > idx0 and not below))
        s = t0 if lane == idx0 else t[lane]
        above = [i for i in range(lane + 1, n) if not sat[i]]
        endt = t[above[0]] if above else None
        ok.append(starts and (endt is None or endt - s >= L))
    if not any(ok):
        return INF
    w = ok.index(True)
    return t0 if w == idx0 else t[w]


@pytest.mark.parametrize("seed", range(5))
def test_next_fit_lane_parallel_equals_walk(seed):
    rng = np.random.default_rng(seed)
    for _ in range(4000):
        n = int(rng.integers(1, 20))
        t = (1000 + np.concatenate([[0], np.cumsum(rng.integers(1, 50, n - 1))])).astype(int).tolist()
        sat = (rng.random(n) < rng.choice([0.2, 0.5, 0.8])).tolist()
        L = int(rng.integers(1, 120))
        t0 = int(rng.integers(1000, t[-1] + 30))
        assert next_fit_lanes(t, sat, L, t0) == next_fit_walk(t, sat, L, t0), (t, sat, L, t0)


# ---- CheckGres_ over an interval ------------------------------------------------------------------------------------
def check_gres_exact(use, has, lim):
    """components in walk order; an entry exists in `use` when its count is > 0 (:1030-1050)"""
    for c in range(len(use)):
        if use[c] == 0:
            continue
        if not has[c]:
            return True            # `return true` at the first requested entry the limit lacks
        if use[c] > lim[c]:
            return False
    return True


def check_gres_interval(lo, hi, has, lim):
    """the ballots of k_par_eval: returns (certain_pass, certain_fail)"""
    n = len(lo)
    sC = [not has[c] and lo[c] > 0 for c in range(n)]
    sM = [not has[c] and hi[c] > 0 for c in range(n)]
    gC = [has[c] and lo[c] > lim[c] for c in range(n)]
    gM = [has[c] and hi[c] > lim[c] for c in range(n)]
    m1 = [sC[c] or gC[c] or gM[c] for c in range(n)]
    m2 = [m1[c] or sM[c] for c in range(n)]
    g_pass = (not any(m1)) or sC[m1.index(True)]
    g_fail = any(m2) and gC[m2.index(True)]
    return g_pass, g_fail


@pytest.mark.parametrize("seed", range(4))
def test_three_valued_gres_walk_is_sound(seed):
    rng = np.random.default_rng(100 + seed)
    seen_pass = seen_fail = seen_unknown = 0
    for _ in range(3000):
        n = int(rng.integers(1, 6))
        has = (rng.random(n) < 0.6).tolist()
        lim = rng.integers(0, 4, n).tolist()
        lo = rng.integers(0, 4, n).tolist()
        hi = [lo[c] + int(rng.integers(0, 3)) for c in range(n)]
        g_pass, g_fail = check_gres_interval(lo, hi, has, lim)
        assert not (g_pass and g_fail)
        outcomes = {check_gres_exact(u, has, lim) for u in itertools.product(*[range(lo[c], hi[c] + 1) for c in range(n)])}
        if g_pass:
            assert outcomes == {True}, (lo, hi, has, lim)
            seen_pass += 1
        elif g_fail:
            assert outcomes == {False}, (lo, hi, has, lim)
            seen_fail += 1
        else:
            seen_unknown += 1
        if lo == hi:      # a zero-width interval is always decided: the progress guarantee of the rounds
            assert g_pass or g_fail
    assert seen_pass and seen_fail and seen_unknown


# ---- the bracketing rounds as an algorithm ---------------------------------------------------------------------------
def greedy(keys, add, lim):
    """admission in order: job j is admitted iff every one of its keys stays within its limit"""
    use = {}
    out = []
    for j in range(len(add)):
        ok = all(use.get(k, 0) + add[j] <= lim[k] for k in keys[j])
        out.append(ok)
        if ok:
            for k in keys[j]:
                use[k] = use.get(k, 0) + add[j]
    return out


def bracketing(keys, add, lim):
    """state 0 undecided / 1 admitted / 2 rejected; per round: sums over earlier admitted (L) and earlier not-rejected
    (U) jobs of each key; passes under U -> admitted, fails under L -> rejected (k_par_tails / k_par_eval / k_par_update)"""
    n = len(add)
    state = [0] * n
    rounds = 0
    while 0 in state:
        rounds += 1
        sumL, sumU = {}, {}
        new = list(state)
        for j in range(n):
            if state[j] == 0:
                pass_u = all(sumU.get(k, 0) + add[j] <= lim[k] for k in keys[j])
                fail_l = any(sumL.get(k, 0) + add[j] > lim[k] for k in keys[j])
                if pass_u:
                    new[j] = 1
                elif fail_l:
                    new[j] = 2
            for k in keys[j]:
                if state[j] == 1:
                    sumL[k] = sumL.get(k, 0) + add[j]
                if state[j] != 2:
                    sumU[k] = sumU.get(k, 0) + add[j]
        assert new != state, "a round must decide at least the first undecided job"
        state = new
    return [s == 1 for s in state], rounds


@pytest.mark.parametrize("seed", range(6))
def test_bracketing_rounds_equal_greedy_admission(seed):
    rng = np.random.default_rng(500 + seed)
    worst = 0
    for _ in range(300):
        n = int(rng.integers(1, 60))
        nk = int(rng.integers(1, 6))
        keys = [tuple({int(rng.integers(0, nk)), nk + int(rng.integers(0, 2)), 2 * nk + 7}) for _ in range(n)]   # user-like, account-like, global
        add = rng.integers(1, 9, n).tolist()
        lim = {k: int(rng.integers(4, 60)) for k in range(3 * nk + 10)}
        want = greedy(keys, add, lim)
        got, rounds = bracketing(keys, add, lim)
        assert got == want
        worst = max(worst, rounds)
    assert worst <= 60


# ---- 4. k_wide's launch layout (cranesched_amd/csrc/wide_kernel.inc: blockIdx -> partition, member; engine.hip: launch_wide) -------------
@pytest.mark.parametrize("group,max_parts", [(17, 8), (9, 24), (5, 48), (3, 80)])
def test_wide_launch_layout(group, max_parts):
    """The four builds of k_wide give a partition 1 + 16 / 8 / 4 / 2 workgroups.  Block b lands on XCD b % 8 (observed; speed only) and an
    XCD has 32 CUs with one workgroup each, so the layout must (a) give every (partition, member) pair of the launch exactly one
    block, (b) keep all workgroups of a partition on one value of b % 8, (c) put at most 32 workgroups on any XCD for up to
    `max_parts` partitions, (d) fit the device's 256 CUs — for every partition count the build serves."""
    for nparts in range(1, max_parts + 1):
        groups = (nparts + 7) // 8
        grid = 8 * groups * group
        seen = {}
        per_xcd = [0] * 8
        for b in range(grid):
            bslot = b >> 3
            pidx = (bslot // group) * 8 + (b & 7)
            m = bslot % group
            if pidx >= nparts:          # the kernel returns at once: no CU held
                continue
            assert (pidx, m) not in seen
            seen[(pidx, m)] = b
            per_xcd[b % 8] += 1
            assert b % 8 == pidx % 8    # (b): a partition's workgroups share the XCD
        assert len(seen) == nparts * group                       # (a)
        assert max(per_xcd) <= 32, (nparts, per_xcd)              # (c)
        assert grid <= 256                                       # (d) what engine.hip's fits() checks before the launch
    assert (32 // group) * 8 == max_parts                        # kWMaxParts of that build


# ---------------------------------------------------------------------------------------------------------------------
# 5. k_select's dips (cranesched_amd/csrc/select_kernels.hip: post_dip, pack_dip, the dip test in the scanners' row loop).
#    The claim: once the worker has posted, under a rejected start-now candidate, either the first entry of the window that
#    cannot host the job or the WINDOW MINIMUM dated at the last entry inside the window, the scanners' packed test never drops
#    a node on which a LATER job would pass the exact test — whatever was committed on the node in between (entries only shrink
#    and split within a cycle without releases).  Resources here: (cpus, GiB); the GRES counts go through the same comparison.
# ---------------------------------------------------------------------------------------------------------------------
def _pack_dip(dt, cpus_up, gib_up):
    """pack_dip: 16 s units rounded UP (0xFFFF: none / too far), whole cpus and GiB capped at 255"""
    t16 = 0xFFFF if dt >= 0xFFFF0 else (dt + 15) >> 4
    return t16, min(cpus_up, 255), min(gib_up, 255)


def _packed_drops(packed, L, req_cpu256, req_mib):
    """the scanners' test: True = the node is NOT proposed to a job with time limit L and this request"""
    t16, c8, m8 = packed
    lq = min(L >> 4, 0xFFFF)
    rc8, rm8 = min(req_cpu256 >> 8, 255), min(req_mib >> 10, 255)       # the request rounded DOWN, capped like the dip
    fits = rc8 <= c8 and rm8 <= m8
    return not (t16 >= lq or fits)


def _window_min(avail0, tmap, L):
    m = list(avail0)
    for t, (c, g) in tmap:
        if t < L:                                                      # entry t seconds after now lies in the window iff t < L (:6279)
            m = [min(m[0], c), min(m[1], g)]
    return m


@pytest.mark.parametrize("seed", range(200))
def test_k_select_dip_never_drops_a_node_that_passes(seed):
    rng = np.random.default_rng(9000 + seed)
    # a node's time map: entry 0 at `now`, later entries at increasing offsets (seconds), resources in 1/256 cpus and MiB
    n = int(rng.integers(1, 12))
    times = [0] + sorted(int(x) for x in rng.choice(np.arange(1, 400_000), size=n - 1, replace=False)) if n > 1 else [0]
    tot = (int(rng.integers(1, 300)) * 256, int(rng.integers(1, 600)) * 1024)
    tmap = [(t, (int(rng.integers(0, tot[0] // 128 + 1)) * 128, int(rng.integers(0, tot[1] // 512 + 1)) * 512)) for t in times]   # (half cpus, half GiB: the rounding edges come up)
    avail0 = tmap[0][1]
    exact_ok = lambda L, rc, rm: all(a >= b for a, b in zip(_window_min(avail0, tmap, L), (rc, rm)))
    for _ in range(40):
        L = int(rng.integers(1, 500_000)); rc = int(rng.integers(1, tot[0] + 300)); rm = int(rng.integers(1, tot[1] + 2000))
        if exact_ok(L, rc, rm):
            continue
        # post_dip for the rejected job (L, rc, rm)
        inw = [i for i, (t, _) in enumerate(tmap) if i == 0 or t < L]
        single = [i for i in inw if i >= 1 and (tmap[i][1][0] < rc or tmap[i][1][1] < rm)]
        if single:
            i = single[0]; r = tmap[i][1]
        else:
            i = inw[-1]; r = _window_min(avail0, tmap, L)
        dt = tmap[i][0]
        packed = _pack_dip(dt, -(-max(r[0], 0) // 256), -(-r[1] // 1024))      # dip_cm_of: cpus and GiB rounded UP
        # ... later commits: entries shrink, new entries split old ones (a copy of the covering entry, then shrunk)
        later = list(tmap)
        for _ in range(int(rng.integers(0, 4))):
            s = int(rng.integers(0, 400_000)); e = s + int(rng.integers(1, 100_000)); dc = int(rng.integers(0, 3)) * 256
            for cut in (s, e):
                if all(t != cut for t, _ in later):
                    cover = max((x for x in later if x[0] <= cut), key=lambda x: x[0])
                    later.append((cut, cover[1]))
            later.sort()
            later = [(t, (max(c - dc, 0), g)) if s <= t < e else (t, (c, g)) for t, (c, g) in later]
        a0 = later[0][1] if later[0][0] == 0 else avail0
        for k in range(60):
            L2 = int(rng.integers(1, 500_000)); rc2 = int(rng.integers(1, tot[0] + 300)); rm2 = int(rng.integers(1, tot[1] + 2000))
            if k % 2:   # at the edges: a window that ends around the dip's date, a request that just passes the exact test
                L2 = max(1, dt + int(rng.integers(-40, 41)))
                m2 = _window_min(avail0, later, L2)
                rc2 = max(1, m2[0] - int(rng.integers(0, 2)) * int(rng.integers(0, 600))); rm2 = max(1, m2[1] - int(rng.integers(0, 2)) * int(rng.integers(0, 3000)))
            passes = all(a >= b for a, b in zip(_window_min(avail0, later, L2), (rc2, rm2)))
            if _packed_drops(packed, L2, rc2, rm2):
                assert not passes, (seed, tmap, (L, rc, rm), (dt, r), packed, later, (L2, rc2, rm2))
        assert _packed_drops(packed, max(L, ((dt + 15) >> 4 << 4) + 16), rc + 256 * 300, rm) or packed[1] == 255 or packed[0] == 0xFFFF


def test_pack_dip_none_and_caps():
    assert _pack_dip(0xFFFFFFFF, 0, 0)[0] == 0xFFFF                       # "no dip" is never inside a window
    assert not _packed_drops(_pack_dip(0xFFFFFFFF, 0, 0), 10**9, 10**6, 10**6)
    assert not _packed_drops(_pack_dip(100, 300, 1), 10**6, 400 * 256, 512)   # a dip with >= 255 cpus fits any cpu request (capped both sides)
    assert _packed_drops(_pack_dip(100, 3, 1), 10**6, 4 * 256, 512)
    assert not _packed_drops(_pack_dip(100, 3, 1), 112, 4 * 256, 512)         # window [0, 112) in 16 s units = 7 = the dip's unit: may not contain it
