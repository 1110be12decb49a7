"""Run-limit admission = the QoS / account / partition post-filter of the commit loop (SURVEY.md §8(f)-1;
JobScheduler.cpp:1492-1573 -> AccountMetaContainer::CheckAndMallocMetaResource, AccountMetaContainer.cpp:180-224,
508-688,891-1124).

The reference has no test of this path, so the expectations below are hand-derived from the cited lines (written
before running anything), checked against the CPU oracle here and against the HIP engine under `-m gpu`; plus
random cases engine-vs-oracle (reasons and every usage record after the pass)."""
import numpy as np
import pytest

from cranesched_amd import abi, limits as lm
from oracle import pyoracle
from tests import helpers, kat

NOW = kat.NOW
GIB = 1 << 30
NONE = lm.LIM_NONE


def _cluster():
    # 8 nodes x 64 cores / 256 GiB; nodes 0-3: 4 a100 + 4 h100 (name 0 "gpu"), nodes 4-7: 8 a910 (name 1 "npu")
    lay = helpers.multi_type_layout()
    return kat.cluster([64] * 8, mem_gib=[256] * 8, gres=[0xFF] * 4 + [0xFF00] * 4, layout=lay), lay


def _tables(qos, parent, U, ua_pairs, Pn=1, **kw):
    return lm.LimitTables(num_users=U, num_user_accts=len(ua_pairs), num_partitions=Pn, qos=np.array(qos, lm.QOS_DT),
                          acct_parent=np.array(parent, np.uint32), **kw)


def _limjobs(keys, ua_pairs, L):
    """keys: (ua index, qos, partition) per job"""
    ua = [k[0] for k in keys]
    return lm.LimitJobs(user=[ua_pairs[x][0] for x in ua], user_acct=ua, account=[ua_pairs[x][1] for x in ua],
                        qos=[k[1] for k in keys], partition=[k[2] for k in keys], time_limit_sec=L)


def scenario_jobs_and_cpu_per_user():
    # one QoS: max_jobs_per_user 3, max_cpus_per_user 8.  CheckQosRunLimitsForEntity_ (:521-531) tests, for a user,
    # cpu (QosCpuResourceLimit) BEFORE the job count.  A rejected job adds nothing (:208-209 returns before :217).
    #   u0: 4c ok (4) | 3c ok (7) | 2c -> 9 > 8 QosCpu | 1c ok (8, 3 jobs) | 1c -> 9 > 8: QosCpu wins over the job count
    #   u1: 8c ok | 8c -> QosCpu | then three 0-cpu-impossible... use 0.5c: 8.5 > 8 QosCpu
    ua = [(0, 0), (1, 0)]
    specs = [dict(cpu=4), dict(cpu=3), dict(cpu=2), dict(cpu=1), dict(cpu=1), dict(cpu=8), dict(cpu=8), dict(cpu=0.5)]
    keys = [(0, 0, 0)] * 5 + [(1, 0, 0)] * 3
    t = _tables([lm.qos_limits(max_jobs_per_user=3, max_cpus_per_user=8)], [NONE], 2, ua)
    exp = [0, 0, 2, 0, 2, 0, 2, 2]
    return specs, keys, ua, t, exp, dict(uq_jobs=[3, 1], uq_cpu=[8 * 256, 8 * 256], qos_jobs=[4])


def scenario_jobs_per_user_only():
    # max_jobs_per_user 2: the third job of u0 is QosJobsResourceLimit (:526), u1 is not affected
    ua = [(0, 0), (1, 0)]
    keys = [(0, 0, 0), (0, 0, 0), (1, 0, 0), (0, 0, 0), (1, 0, 0)]
    t = _tables([lm.qos_limits(max_jobs_per_user=2)], [NONE], 2, ua)
    return [dict(cpu=1)] * 5, keys, ua, t, [0, 0, 0, 3, 0], dict(uq_jobs=[2, 2], qos_jobs=[4])


def scenario_account_chain():
    # accounts: a0 root, a1 and a2 its children; max_tres_per_account cpu = 10 applies to EVERY account of the chain
    # (CheckRunLimits_ :946-982 walks job.account_chain).  a1: 6c ok (a1 6, a0 6) | a2: 6c: a2 6 ok, a0 12 > 10 ->
    # CpuResourceLimit | a2: 4c ok (a2 4, a0 10) | a1: 1c: a1 7 ok, a0 11 > 10 -> CpuResourceLimit
    ua = [(0, 1), (1, 2)]
    keys = [(0, 0, 0), (1, 0, 0), (1, 0, 0), (0, 0, 0)]
    specs = [dict(cpu=6), dict(cpu=6), dict(cpu=4), dict(cpu=1)]
    t = _tables([lm.qos_limits(max_tres_per_account=lm.tres(cpu=10))], [NONE, 0, 0], 2, ua)
    return specs, keys, ua, t, [0, 5, 0, 5], dict(aq_cpu=[10 * 256, 6 * 256, 4 * 256])


def scenario_wall():
    # Qos max_wall 1000 caps the SUM of time limits per user, per account and globally (:528-530,535-537,1004-1013)
    #   u0 L=600 ok | u0 L=500: 1100 > 1000 at the user -> QosWall | u1 L=500: user 500 ok, account 600+500 -> QosWall
    #   u1 L=400: user 400, account 1000, global 1000: ok | u2 L=1: account 1001 -> QosWall
    ua = [(0, 0), (1, 0), (2, 0)]
    keys = [(0, 0, 0), (0, 0, 0), (1, 0, 0), (1, 0, 0), (2, 0, 0)]
    specs = [dict(cpu=1, L=600), dict(cpu=1, L=500), dict(cpu=1, L=500), dict(cpu=1, L=400), dict(cpu=1, L=1)]
    t = _tables([lm.qos_limits(max_wall_sec=1000)], [NONE], 3, ua)
    return specs, keys, ua, t, [0, 4, 4, 0, 4], dict(qos_jobs=[2])


def scenario_gres_quirks():
    # per-user TRES limit: gpu total 3, a100 1; h100 and npu have no entry in the limit.
    # CheckGres_ (:1030-1050) returns TRUE at the first entry of `use` the limit lacks.  Canonical walk: gpu total,
    # a100, h100, npu total, a910.
    #   A: 1 a100            use gpu 1 <= 3, a100 1 <= 1                              -> ok
    #   B: 1 a100            use gpu 2, a100 2 > 1                                     -> GresResourceLimit
    #   C: 1 h100            use gpu 2 <= 3, a100 1 ok, h100 not in the limit -> true  -> ok
    #   D: 2 h100            use gpu 4 > 3                                             -> GresResourceLimit
    #   E: 8 a910 (npu)      use gpu 2 ok, a100 ok, h100 -> true before npu is looked at -> ok
    #   F (u1): 8 a910       use npu only: name not in the limit -> true               -> ok
    ua = [(0, 0), (1, 0)]
    g = lambda tot, spec: dict(cpu=1, gtot=tot, gspec=spec)
    specs = [g([1], [1]), g([1], [1]), g([1], [0, 1]), g([2], [0, 2]), g([0, 8], [0, 0, 8]), g([0, 8], [0, 0, 8])]
    keys = [(0, 0, 0)] * 5 + [(1, 0, 0)]
    t = _tables([lm.qos_limits(max_tres_per_user=lm.tres(names={0: 3}, classes={0: 1}))], [NONE], 2, ua)
    return specs, keys, ua, t, [0, 7, 0, 7, 0, 0], {}


def scenario_gres_stop_hides_failure():
    # per-account limit: npu total 2 only.  Account a0 already uses 1 gpu: `use` starts with the gpu name, which the
    # limit lacks -> CheckGres_ returns true and 8 npu > 2 is never seen.  Account a1 uses npu only -> fails.
    ua = [(0, 0), (1, 1)]
    g = lambda tot, spec: dict(cpu=1, gtot=tot, gspec=spec)
    specs = [g([1], [1]), g([0, 8], [0, 0, 8]), g([0, 8], [0, 0, 8]), g([0, 2], [0, 0, 2])]
    keys = [(0, 0, 0), (0, 0, 0), (1, 0, 0), (1, 0, 0)]
    t = _tables([lm.qos_limits(max_tres_per_account=lm.tres(names={1: 2}))], [NONE, NONE], 2, ua)
    return specs, keys, ua, t, [0, 0, 7, 0], {}


def scenario_partition_limits():
    # two limit partitions.  part_limits: 0 = {max_jobs 1} for (u0,a0) x p0; 1 = {max_tres cpu 4} for a0 x p1;
    # 2 = {max_wall 100} for a0 x p0.  QoS 0 caps nothing; QoS 1 caps jobs per user (5), so the PARTITION job cap is
    # not looked at for it (:577).
    #   0: u0 q0 p0 L=60: ok            (user part jobs 1, a0 p0 wall 60)
    #   1: u0 q0 p0 L=10: user partition jobs 2 > 1                      -> UserPartitionJobsLimit
    #   2: u0 q1 p0 L=10: QoS 1 caps jobs -> partition cap skipped; a0 p0 wall 70 ok -> ok
    #   3: u1 q0 p0 L=40: (u1,a0) has no user limit; a0 p0 wall 70+40 > 100 -> AccPartitionWallTimeLimit
    #   4: u1 q0 p1 3c : a0 p1 cpu 3 <= 4 ok
    #   5: u1 q0 p1 2c : a0 p1 cpu 5 > 4                                  -> PartitionCpuResourceLimit
    #   6: u1 q1 p1 1c : cpu 4 <= 4 ok (QoS 1's per-account TRES is unlimited too)
    ua = [(0, 0), (1, 0)]
    specs = [dict(cpu=1, L=60), dict(cpu=1, L=10), dict(cpu=1, L=10), dict(cpu=1, L=40), dict(cpu=3), dict(cpu=2), dict(cpu=1)]
    keys = [(0, 0, 0), (0, 0, 0), (0, 1, 0), (1, 0, 0), (1, 0, 1), (1, 0, 1), (1, 1, 1)]
    pl = np.array([lm.part_limit(max_jobs=1), lm.part_limit(max_tres=lm.tres(cpu=4)), lm.part_limit(max_wall_sec=100)], lm.PART_LIMIT_DT)
    t = _tables([lm.qos_limits(), lm.qos_limits(max_jobs_per_user=5)], [NONE], 2, ua, Pn=2, part_limits=pl,
                user_part_limit=[0, NONE, NONE, NONE], acct_part_limit=[2, 1])
    return specs, keys, ua, t, [0, 9, 0, 12, 0, 13, 0], {}


def scenario_missing_entries():
    # (u1, q0) has no entry in the user's qos map -> QosEntryNotFound (:514-517).  a0 x p0 has a limit but no entry ->
    # PartitionEntryNotFound (:619-622) for every job of partition 0; partition 1 has no limit: the missing entry is
    # created by DoMallocResource_ (:1078-1123).
    ua = [(0, 0), (1, 0)]
    specs = [dict(cpu=1)] * 4
    keys = [(1, 0, 1), (0, 0, 0), (0, 0, 1), (0, 0, 1)]
    pl = np.array([lm.part_limit(max_jobs=100)], lm.PART_LIMIT_DT)
    t = _tables([lm.qos_limits()], [NONE], 2, ua, Pn=2, part_limits=pl, acct_part_limit=[0, NONE],
                user_qos_exists=[1, 0], acct_part_exists=[0, 0], user_part_exists=[0, 0, 0, 0])
    return specs, keys, ua, t, [1, 8, 0, 0], dict(ap_exists=[0, 1], up_exists=[0, 1, 0, 0])


SCENARIOS = {f.__name__[9:]: f for f in (scenario_jobs_and_cpu_per_user, scenario_jobs_per_user_only, scenario_account_chain,
                                          scenario_wall, scenario_gres_quirks, scenario_gres_stop_hides_failure,
                                          scenario_partition_limits, scenario_missing_entries)}


def _check_extra(usage, extra):
    if "uq_jobs" in extra:
        assert list(usage.user_qos["jobs_count"]) == extra["uq_jobs"]
    if "uq_cpu" in extra:
        assert list(usage.user_qos["cpu_raw"]) == extra["uq_cpu"]
    if "qos_jobs" in extra:
        assert list(usage.qos_usage["jobs_count"]) == extra["qos_jobs"]
    if "aq_cpu" in extra:
        assert list(usage.acct_qos["cpu_raw"]) == extra["aq_cpu"]
    if "ap_exists" in extra:
        assert list(usage.acct_part_exists) == extra["ap_exists"]
    if "up_exists" in extra:
        assert list(usage.user_part_exists) == extra["up_exists"]


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_oracle_kat(name):
    specs, keys, ua, t, exp, extra = SCENARIOS[name]()
    cluster, lay = _cluster()
    jobs = kat.jobs(specs)
    sel = pyoracle.select(cluster, jobs, NOW)
    assert not sel.placements.reason[:jobs.num_jobs].any(), "KAT jobs must all start now"
    lj = _limjobs(keys, ua, jobs.time_limit_sec)
    reason, adm, usage = pyoracle.run_limits(lay, t, lj, sel.placements)
    assert list(reason) == exp, [lm.LIMIT_REASON_STR[int(r)] for r in reason]
    assert adm == exp.count(0)
    _check_extra(usage, extra)


def test_oracle_non_candidates_and_order():
    # a job NodeSelect left pending (reason != "") never reaches the check (:1507-1510); `skip` mirrors the other
    # `continue`s; the pass runs in the order of the limit-job table, not of the select table
    cluster, lay = _cluster()
    jobs = kat.jobs([dict(cpu=64, k=8, ntasks=8), dict(cpu=1), dict(cpu=1), dict(cpu=1)])  # job 0 fills the cluster
    sel = pyoracle.select(cluster, jobs, NOW)
    assert list(sel.placements.reason[:4]) == [0, abi.REASON_PRIORITY, abi.REASON_PRIORITY, abi.REASON_PRIORITY] or \
        sel.placements.reason[1] != 0
    ua = [(0, 0)]
    t = _tables([lm.qos_limits(max_jobs_per_user=1)], [NONE], 1, ua)
    lj = lm.LimitJobs(user=[0] * 4, user_acct=[0] * 4, account=[0] * 4, qos=[0] * 4, partition=[0] * 4,
                      time_limit_sec=jobs.time_limit_sec[[3, 2, 1, 0]], select_index=[3, 2, 1, 0], skip=[0, 0, 0, 0])
    reason, adm, _ = pyoracle.run_limits(lay, t, lj, sel.placements)
    assert list(reason) == [255, 255, 255, 0] and adm == 1
    lj.skip = np.array([0, 0, 0, 1], np.uint8)
    reason, adm, _ = pyoracle.run_limits(lay, t, lj, sel.placements)
    assert list(reason) == [255, 255, 255, 255] and adm == 0


def random_limit_case(seed, J=600, N=96, tight=True):
    """Random cluster / queue (tests.helpers) + a random account tree, QoS set, partition limits, initial usage."""
    rng = np.random.default_rng(seed + 7000)
    cluster, jobs, now, _ = helpers.random_case(seed, N=N, J=J, P=2, running=0)
    lay = cluster.gres
    U, A, Q, Pn = 7, 6, 3, 2
    parent = [NONE, 0, 0, 1, 3, NONE][:A]           # two trees, depth up to 4 (a4 -> a3 -> a1 -> a0)
    ua_pairs = [(u, int(a)) for u in range(U) for a in rng.choice(A, rng.integers(1, 3), replace=False)]
    UA = len(ua_pairs)
    big = 10 ** 6

    def rtres(scale):
        if rng.random() < 0.3:
            return lm.unlimited_tres()
        names = {int(n): int(rng.integers(1, 6 * scale)) for n in range(2) if rng.random() < 0.5}
        classes = {int(g): int(rng.integers(1, 4 * scale)) for g in range(3) if rng.random() < 0.4}
        return lm.tres(cpu=int(rng.integers(8, 64 * scale)) if rng.random() < 0.7 else None,
                       mem=int(rng.integers(32, 256 * scale)) * GIB if rng.random() < 0.5 else None, names=names, classes=classes)

    s = 1 if tight else 50
    qos = [lm.qos_limits(max_jobs_per_user=int(rng.integers(2, 12 * s)) if rng.random() < 0.6 else lm.UNLIMITED_JOBS,
                         max_jobs_per_account=int(rng.integers(5, 40 * s)) if rng.random() < 0.5 else lm.UNLIMITED_JOBS,
                         max_jobs=int(rng.integers(30, 150 * s)) if rng.random() < 0.5 else lm.UNLIMITED_JOBS,
                         max_cpus_per_user=int(rng.integers(8, 100 * s)) if rng.random() < 0.5 else None,
                         max_wall_sec=int(rng.integers(20000, 200000 * s)) if rng.random() < 0.4 else 0,
                         max_tres=rtres(8 * s), max_tres_per_user=rtres(s), max_tres_per_account=rtres(3 * s))
           for _ in range(Q)]
    pls = np.array([lm.part_limit(max_jobs=int(rng.integers(1, 15 * s)) if rng.random() < 0.6 else lm.UNLIMITED_JOBS,
                                  max_wall_sec=int(rng.integers(5000, 100000 * s)) if rng.random() < 0.5 else 0,
                                  max_tres=rtres(2 * s)) for _ in range(5)], lm.PART_LIMIT_DT)
    pick = lambda n: np.where(rng.random(n) < 0.4, rng.integers(0, len(pls), n), NONE).astype(np.uint32)

    def rusage(n, p=0.3):
        u = np.zeros(n, lm.USAGE_DT)
        on = rng.random(n) < p
        u["cpu_raw"] = np.where(on, rng.integers(0, 8, n) * 256, 0)
        u["mem"] = np.where(on, rng.integers(0, 16, n) * GIB, 0)
        u["wall_sec"] = np.where(on, rng.integers(0, 5000, n), 0)
        u["jobs_count"] = np.where(on, rng.integers(0, 3, n), 0)
        for g in range(3):
            c = np.where(on & (rng.random(n) < 0.3), rng.integers(0, 3, n), 0)
            u["class_count"][:, g] = c
            u["name_total"][:, lay.class_name[g]] += c.astype(np.uint64)
        return u

    ex = lambda n: (rng.random(n) > 0.03).astype(np.uint8)
    t = lm.LimitTables(num_users=U, num_user_accts=UA, num_partitions=Pn, qos=np.array(qos, lm.QOS_DT),
                       acct_parent=np.array(parent, np.uint32), part_limits=pls, user_part_limit=pick(UA * Pn),
                       acct_part_limit=pick(A * Pn), user_qos=rusage(U * Q), user_qos_exists=ex(U * Q),
                       user_part=rusage(UA * Pn), user_part_exists=ex(UA * Pn), acct_qos=rusage(A * Q),
                       acct_qos_exists=ex(A * Q), acct_part=rusage(A * Pn), acct_part_exists=ex(A * Pn),
                       qos_usage=rusage(Q, 1.0))
    Jn = jobs.num_jobs
    order = rng.permutation(Jn).astype(np.uint64)          # commit-loop order != NodeSelect order
    uax = rng.integers(0, UA, Jn)
    lj = lm.LimitJobs(user=[ua_pairs[x][0] for x in uax], user_acct=uax, account=[ua_pairs[x][1] for x in uax],
                      qos=rng.integers(0, Q, Jn), partition=rng.integers(0, Pn, Jn),
                      time_limit_sec=jobs.time_limit_sec[order.astype(np.int64)], select_index=order,
                      skip=(rng.random(Jn) < 0.02).astype(np.uint8))
    return cluster, jobs, now, lay, t, lj


@pytest.mark.parametrize("seed", [1, 2])
def test_oracle_random_properties(seed):
    """Admission is monotone: replaying the admitted set alone admits exactly the same jobs; and jobs_count grows by
    the number of admitted jobs in every table a job touches."""
    cluster, jobs, now, lay, t, lj = random_limit_case(seed, J=400)
    sel = pyoracle.select(cluster, jobs, now)
    reason, adm, usage = pyoracle.run_limits(lay, t, lj, sel.placements)
    assert adm == int((reason == 0).sum()) and 0 < adm
    assert set(np.unique(reason)) - {0, 255} , "the random limits should reject something"
    base = t.qos_usage["jobs_count"].astype(np.int64).sum()
    assert usage.qos_usage["jobs_count"].astype(np.int64).sum() == base + adm
    lj2 = lm.LimitJobs(lj.user, lj.user_acct, lj.account, lj.qos, lj.partition, lj.time_limit_sec, lj.select_index,
                       skip=((reason != 0) | (lj.skip != 0)).astype(np.uint8))
    reason2, adm2, usage2 = pyoracle.run_limits(lay, t, lj2, sel.placements)
    assert adm2 == adm and np.array_equal(reason2 == 0, reason == 0) and usage2.same_as(usage)


def test_limits_abi_symbols(built):
    import re, os
    from cranesched_amd import engine
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "crane_gpu", "run_limits.h")).read()
    declared = set(re.findall(r"\b(cns_[a-z_]+)\s*\(", hdr))
    assert declared == set(engine.LIMITS_ABI_SYMBOLS)
    for s in declared:
        getattr(engine.lib(), s)
    # struct sizes the ctypes / numpy mirrors rely on
    assert lm.TRES_DT.itemsize == 120 and lm.USAGE_DT.itemsize == 128


# ---------------------------------------------------------------- GPU ----------------------------------------------------
@pytest.fixture(params=["parallel", "ordered"])
def mode(request, monkeypatch):
    """Both device paths: the bracketing rounds over sorted items (default) and the ordered single-wave kernel."""
    if request.param == "ordered":
        monkeypatch.setenv("CNS_LIMITS_MODE", "seq")
    else:
        monkeypatch.delenv("CNS_LIMITS_MODE", raising=False)
    return request.param


def _gpu_vs_oracle(engine_default, cluster, jobs, now, lay, t, lj, tag="", expect_fallback=None):
    eng = engine_default(device=0)
    try:
        eng.set_nodes(cluster)
        got = eng.node_select(now, jobs)
        ref = pyoracle.select(cluster, jobs, now)
        assert got.diff(ref.placements) is None, tag
        eng.set_run_limits(t)
        reason, adm = eng.apply_run_limits(lj)
        usage = eng.usage()
        r_ref, a_ref, u_ref = pyoracle.run_limits(lay, t, lj, ref.placements)
        bad = np.nonzero(reason != r_ref)[0]
        assert bad.size == 0, f"{tag}: job {bad[0]}: gpu {lm.LIMIT_REASON_STR[int(reason[bad[0]])]!r} oracle {lm.LIMIT_REASON_STR[int(r_ref[bad[0]])]!r}"
        assert adm == a_ref
        for f in usage.__dataclass_fields__:
            assert np.array_equal(getattr(usage, f), getattr(u_ref, f)), f"{tag}: usage table {f} differs"
        tm = eng.limit_timing()
        assert tm["admitted"] == adm and tm["candidates"] == int((r_ref != 255).sum())
        if expect_fallback is not None:
            assert bool(tm["ordered_fallback"]) == expect_fallback, tm
        # re-running from the same tables gives the same answer (the working copy is reset)
        eng.run_limits_resident()
        reason2, adm2 = eng.download_limits()
        assert np.array_equal(reason2, reason) and adm2 == adm
        return reason, usage
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_gpu_kat(engine_default, name, mode):
    specs, keys, ua, t, exp, extra = SCENARIOS[name]()
    cluster, lay = _cluster()
    jobs = kat.jobs(specs)
    lj = _limjobs(keys, ua, jobs.time_limit_sec)
    reason, usage = _gpu_vs_oracle(engine_default, cluster, jobs, NOW, lay, t, lj, name, expect_fallback=mode == "ordered")
    assert list(reason) == exp
    _check_extra(usage, extra)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,tight", [(1, True), (2, True), (3, True), (4, False), (5, True), (6, False)])
def test_gpu_random(engine_default, seed, tight, mode):
    cluster, jobs, now, lay, t, lj = random_limit_case(seed, J=900, N=128, tight=tight)
    _gpu_vs_oracle(engine_default, cluster, jobs, now, lay, t, lj, f"seed {seed}", expect_fallback=mode == "ordered")


def dependency_chain_case(n=200):
    """job i is admitted iff job i-1 was rejected: jobs 2m-1, 2m share an (account, qos) usage record capped at 1 core,
    jobs 2m, 2m+1 share an (account, partition) record capped at 1 job.  The bracketing rounds decide one job per
    round here, so the engine must notice and hand over to the ordered kernel (same answer: 1,0,1,0,...)."""
    cluster, lay = _cluster()
    jobs = kat.jobs([dict(cpu=1)] * n)
    Q = Pn = n // 2 + 1
    ua = [(0, 0)]
    qos = [lm.qos_limits(max_tres_per_account=lm.tres(cpu=1, mem=1 << 60)) for _ in range(Q)]
    pl = np.array([lm.part_limit(max_jobs=1, max_tres=lm.tres(mem=1 << 60))], lm.PART_LIMIT_DT)
    t = _tables(qos, [NONE], 1, ua, Pn=Pn, part_limits=pl, acct_part_limit=[0] * Pn)
    keys = [(0, (i + 1) // 2, i // 2) for i in range(n)]      # i = 0: (q0, p0); 1: (q1, p0); 2: (q1, p1); 3: (q2, p1) ...
    return cluster, jobs, lay, t, _limjobs(keys, ua, jobs.time_limit_sec)


def test_oracle_dependency_chain():
    cluster, jobs, lay, t, lj = dependency_chain_case(40)
    sel = pyoracle.select(cluster, jobs, NOW)
    reason, adm, _ = pyoracle.run_limits(lay, t, lj, sel.placements)
    # 0: ok | 1: a0 x p0 has a job -> AccPartitionJobsLimit | 2: (a0,q1) empty (1 was rejected), p1 empty -> ok | 3: (a0,q2) ok, p1 full ...
    assert list(reason[:6]) == [0, 11, 0, 11, 0, 11] and adm == 20


@pytest.mark.gpu
def test_gpu_dependency_chain_falls_back_to_ordered_kernel(engine_default, monkeypatch):
    monkeypatch.delenv("CNS_LIMITS_MODE", raising=False)
    cluster, jobs, lay, t, lj = dependency_chain_case(200)
    reason, _ = _gpu_vs_oracle(engine_default, cluster, jobs, NOW, lay, t, lj, "chain", expect_fallback=True)
    assert list(reason[:4]) == [0, 11, 0, 11]
    cluster, jobs, lay, t, lj = dependency_chain_case(40)        # short chain: the rounds finish it
    _gpu_vs_oracle(engine_default, cluster, jobs, NOW, lay, t, lj, "short chain", expect_fallback=False)


@pytest.mark.gpu
def test_gpu_state_and_argument_checks(engine_default):
    from cranesched_amd.engine import EngineError
    cluster, lay = _cluster()
    jobs = kat.jobs([dict(cpu=1)] * 3)
    ua = [(0, 0)]
    t = _tables([lm.qos_limits()], [NONE], 1, ua)
    lj = _limjobs([(0, 0, 0)] * 3, ua, jobs.time_limit_sec)
    eng = engine_default(device=0)
    try:
        with pytest.raises(EngineError):      # the GRES layout arrives with the nodes
            eng.set_run_limits(t)
        eng.set_nodes(cluster)
        eng.set_run_limits(t)
        with pytest.raises(EngineError):      # no NodeSelect results yet
            eng.upload_limit_jobs(lj)
        eng.node_select(NOW, jobs)
        bad = _limjobs([(0, 0, 0)] * 3, ua, jobs.time_limit_sec)
        bad.qos = np.array([0, 5, 0], np.uint32)
        with pytest.raises(EngineError):
            eng.apply_run_limits(bad)
        cyc = _tables([lm.qos_limits()], [1, 0], 1, [(0, 0)])   # a cycle in the account tree
        with pytest.raises(EngineError):
            eng.set_run_limits(cyc)
        eng.set_run_limits(t)
        reason, adm = eng.apply_run_limits(lj)
        assert list(reason) == [0, 0, 0] and adm == 3
    finally:
        eng.close()


@pytest.mark.parametrize("seed,tight", [(1, True), (2, True), (3, True), (4, False), (5, True), (6, False)])
def test_oracle_vs_independent_python_restatement(seed, tight):
    """The C++ oracle against a second restatement that shares no code with it (tests/limits_pyref.py: plain dicts
    keyed like the reference's maps) — the strongest pin available while the reference itself cannot run."""
    from tests import limits_pyref
    cluster, jobs, now, lay, t, lj = random_limit_case(seed, J=500, N=96, tight=tight)
    sel = pyoracle.select(cluster, jobs, now)
    r_cpp, a_cpp, _ = pyoracle.run_limits(lay, t, lj, sel.placements)
    r_py, a_py = limits_pyref.run(lay, t, lj, sel.placements)
    bad = np.nonzero(r_cpp != r_py)[0]
    assert bad.size == 0, f"job {bad[0]}: C++ {lm.LIMIT_REASON_STR[int(r_cpp[bad[0]])]!r}, python {lm.LIMIT_REASON_STR[int(r_py[bad[0]])]!r}"
    assert a_cpp == a_py
