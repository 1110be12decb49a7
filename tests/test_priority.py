"""MultiFactorPriority: the oracle against hand-derived values, the C ABI surface, and — on
an MI355X — the engine against the oracle, bit for bit (priorities compared as fp64 bit patterns)."""
import numpy as np
import pytest

from cranesched_amd.priority import PrioPending, PrioRunning, PriorityConfig, synth_priority_case
from oracle import pyoracle

NOW = 1_700_000_000
GIB = 1 << 30


def _kat():
    """Three pending jobs, two running jobs, two accounts — every factor derived by hand from
    JobScheduler.cpp:7633-7817 (numbers in the asserts below).
      pending  age  qos part nodes cpus mem   account
        0      100   0   1    1     1   1 GiB   0
        1      300  10   1    2     4   4 GiB   1
        2     9999  10   5    1     2   2 GiB   0      (age capped at MaxAge = 500)
      running  run_time nodes cpus mem   account
        0       1000     1     2   2 GiB   0
        1       2000     4     8   8 GiB   1
    bounds: age [100,500]  qos [0,10]  part [1,5]  nodes [1,4]  cpus [1,8]  mem [1,8] GiB
    service_val(r0) = (2-1)/7 + 0/3 + 1/7 ; service_val(r1) = 1 + 1 + 1 = 3
    acc0 = (2/7)*1000, acc1 = 3*2000 = 6000 -> sv [285.71.., 6000]
    """
    cfg = PriorityConfig(max_age_sec=500, weight_age=1000, weight_fair_share=2000, weight_job_size=300,
                         weight_partition=40, weight_qos=5, favor_small=True)
    pd = PrioPending(submit_sec=[NOW - 100, NOW - 300, NOW - 9999], qos_priority=[0, 10, 10],
                     partition_priority=[1, 1, 5], node_num=[1, 2, 1], total_cpu_raw=[256, 4 * 256, 2 * 256],
                     total_mem=[GIB, 4 * GIB, 2 * GIB], account=[0, 1, 0])
    rn = PrioRunning(start_sec=[NOW - 1000, NOW - 2000], qos_priority=[0, 10], partition_priority=[1, 5],
                     node_num=[1, 4], alloc_cpu_raw=[2 * 256, 8 * 256], alloc_mem=[2 * GIB, 8 * GIB], account=[0, 1])
    return cfg, pd, rn


def _expected_kat():
    acc0 = ((2.0 - 1.0) / (8.0 - 1.0) + 1.0 * 0 / 3 + 1.0 * float(GIB) / float(7 * GIB)) * 1000.0
    acc1 = 3.0 * 2000.0
    out = []
    for age, qos, part, nn, cpus, mem, acc in ((100, 0, 1, 1, 1.0, 1, acc0), (300, 10, 1, 2, 4.0, 4, acc1),
                                               (500, 10, 5, 1, 2.0, 2, acc0)):
        age_f = 1.0 * float(age - 100) / float(400)
        qos_f = 1.0 * (qos - 0) / 10
        part_f = 1.0 * (part - 1) / 4
        size = 0.0
        size += 1.0 * (cpus - 1.0) / 7.0
        size += 1.0 * (nn - 1) / 3
        size += 1.0 * float((mem - 1) * GIB) / float(7 * GIB)
        size = 1.0 - size / 3
        fair = 1.0 - (acc - acc0) / (acc1 - acc0)
        out.append(1000 * age_f + 40 * part_f + 300 * size + 2000 * fair + 5 * qos_f)
    return np.array(out)


def test_oracle_kat():
    cfg, pd, rn = _kat()
    order, prio = pyoracle.priority_order(NOW, cfg, 2, pd, rn)
    exp = _expected_kat()
    assert prio.view(np.uint64).tolist() == exp.view(np.uint64).tolist()
    # job 2: oldest, fair share 1 -> first; job 0: fair share 1, youngest; job 1: fair share 0
    assert order.tolist() == [2, 0, 1]


def test_oracle_degenerate_bounds_and_ties():
    # all attributes equal: every max == min -> every factor 0 except job size (favor_small: 1 - 0/3 = 1);
    # equal priorities keep input order (canonical tie-break); cached priorities are kept verbatim (:7616)
    cfg = PriorityConfig(max_age_sec=1000, weight_age=7, weight_fair_share=11, weight_job_size=13,
                         weight_partition=17, weight_qos=19, favor_small=True)
    J = 6
    pd = PrioPending(submit_sec=[NOW - 50] * J, qos_priority=[3] * J, partition_priority=[2] * J, node_num=[1] * J,
                     total_cpu_raw=[512] * J, total_mem=[GIB] * J, account=[0] * J,
                     cached_priority=[0, 0, 99.5, 0, 13.0, 0])
    order, prio = pyoracle.priority_order(NOW, cfg, 1, pd, None)
    assert prio.tolist() == [13.0, 13.0, 99.5, 13.0, 13.0, 13.0]
    assert order.tolist() == [2, 0, 1, 3, 4, 5]
    cfg.favor_small = False
    _, prio2 = pyoracle.priority_order(NOW, cfg, 1, pd, None)
    assert prio2[0] == 0.0 and prio2[2] == 99.5


def test_oracle_properties_random():
    pd, rn, now = synth_priority_case(5000, 800, 37, seed=5, cached_frac=0.1)
    cfg = PriorityConfig()
    order, prio = pyoracle.priority_order(now, cfg, 37, pd, rn)
    assert sorted(order.tolist()) == list(range(5000))
    p = prio[order]
    assert (p[:-1] >= p[1:]).all()
    same = p[:-1] == p[1:]
    assert (order[:-1][same] < order[1:][same]).all()          # ties: ascending input index
    kept = pd.cached_priority != 0
    assert (prio[kept] == pd.cached_priority[kept]).all()


def test_priority_abi_symbols():
    from cranesched_amd import engine
    L = engine.lib()
    for s in engine.PRIORITY_ABI_SYMBOLS:
        assert hasattr(L, s), s


# ---- GPU parity ------------------------------------------------------------------------------------------

def _gpu_vs_oracle(engine_default, cfg, A, pd, rn, now, limit=None):
    eng = engine_default(device=0)
    try:
        order, prio, nord = eng.priority_order(now, cfg, A, pd, rn, limit=limit)
        ro, rp = pyoracle.priority_order(now, cfg, A, pd, rn)
        assert prio.view(np.uint64).tolist() == rp.view(np.uint64).tolist(), "priorities differ (fp64 bit patterns)"
        assert order.tolist() == ro.tolist(), "order differs"
        assert nord == min(pd.num_jobs, pd.num_jobs if limit is None else limit)
        return eng.priority_timing()
    finally:
        eng.close()


@pytest.mark.gpu
def test_gpu_priority_kat(engine_default):
    cfg, pd, rn = _kat()
    _gpu_vs_oracle(engine_default, cfg, 2, pd, rn, NOW)


@pytest.mark.gpu
@pytest.mark.parametrize("J,R,A,seed,cached", [(1, 0, 1, 1, 0.0), (257, 3, 2, 2, 0.0), (4097, 1000, 64, 3, 0.2),
                                               (100_000, 20_000, 300, 4, 0.05), (50_000, 0, 10, 5, 0.0)])
def test_gpu_priority_random(engine_default, J, R, A, seed, cached):
    pd, rn, now = synth_priority_case(J, R, A, seed, cached_frac=cached)
    for cfg in (PriorityConfig(), PriorityConfig(favor_small=False, weight_job_size=777, weight_age=3)):
        _gpu_vs_oracle(engine_default, cfg, A, pd, rn if R else None, now, limit=J // 2 + 1)


@pytest.mark.gpu
def test_gpu_priority_all_equal(engine_default):
    J = 10_000   # one priority value for everybody: the stable radix sort must return the identity
    pd = PrioPending(submit_sec=[NOW - 5] * J, qos_priority=[1] * J, partition_priority=[1] * J, node_num=[1] * J,
                     total_cpu_raw=[256] * J, total_mem=[GIB] * J, account=[0] * J)
    eng = engine_default(device=0)
    try:
        order, prio, _ = eng.priority_order(NOW, PriorityConfig(), 1, pd, None)
        assert order.tolist() == list(range(J))
    finally:
        eng.close()


@pytest.mark.gpu
def test_gpu_priority_full_size(engine_default):
    # 1 M pending jobs (the benchmark queue length): sortedness + permutation + tie order, and a sample vs the oracle
    J, R, A = 1_000_000, 100_000, 256
    pd, rn, now = synth_priority_case(J, R, A, seed=9)
    t = _gpu_vs_oracle(engine_default, PriorityConfig(), A, pd, rn, now)
    assert t["kernels_ms"] > 0


# ---------------------------------------------------------------------------------------------------------------------
# The oracle against the independent Python restatement (tests/prio_pyref.py, written from the reference alone)
# ---------------------------------------------------------------------------------------------------------------------
def _pyref_run(now, cfg, pd, rn):
    from tests import prio_pyref as pr
    pend = [dict(submit=int(pd.submit_sec[i]), qos=int(pd.qos_priority[i]), part=int(pd.partition_priority[i]),
                 nodes=int(pd.node_num[i]), cpu_raw=int(pd.total_cpu_raw[i]), mem=int(pd.total_mem[i]), account=int(pd.account[i]),
                 cached=0.0 if pd.cached_priority is None else float(pd.cached_priority[i])) for i in range(pd.num_jobs)]
    run = [] if rn is None else [
        dict(start=int(rn.start_sec[i]), qos=int(rn.qos_priority[i]), part=int(rn.partition_priority[i]), nodes=int(rn.node_num[i]),
             cpu_raw=int(rn.alloc_cpu_raw[i]), mem=int(rn.alloc_mem[i]), account=int(rn.account[i])) for i in range(rn.num_jobs)]
    c = dict(max_age=cfg.max_age_sec, w_age=cfg.weight_age, w_fair=cfg.weight_fair_share, w_size=cfg.weight_job_size,
             w_part=cfg.weight_partition, w_qos=cfg.weight_qos, favor_small=cfg.favor_small)
    return pr.ordered(now, c, pend, run)


def test_python_restatement_on_the_hand_derived_case():
    cfg, pd, rn = _kat()
    order, prio = _pyref_run(NOW, cfg, pd, rn)
    assert np.array(prio).view(np.uint64).tolist() == _expected_kat().view(np.uint64).tolist()
    assert order == [2, 0, 1]


@pytest.mark.parametrize("seed", range(40))
def test_python_restatement_agrees_with_the_oracle(seed):
    rng = np.random.default_rng(seed)
    J, R, A = int(rng.integers(1, 400)), int(rng.integers(0, 120)), int(rng.integers(1, 12))
    pd, rn, now = synth_priority_case(J, R, A, seed=seed, cached_frac=0.15 if seed % 3 == 0 else 0.0)
    if seed % 5 == 0:       # degenerate bounds: one value per attribute
        pd.node_num[:] = 2; pd.total_mem[:] = 4 * GIB
        if R:
            rn.node_num[:] = 2; rn.alloc_mem[:] = 4 * GIB
    if seed % 7 == 0:       # a job "submitted in the future": the unsigned age wraps and is capped at MaxAge (:7664-7665)
        pd.submit_sec[0] = now + 50
    cfg = PriorityConfig(max_age_sec=int(rng.integers(100, 100000)), weight_age=int(rng.integers(0, 2000)),
                         weight_fair_share=int(rng.integers(0, 2000)), weight_job_size=int(rng.integers(0, 2000)),
                         weight_partition=int(rng.integers(0, 2000)), weight_qos=int(rng.integers(0, 2000)),
                         favor_small=bool(seed % 2))
    order, prio = pyoracle.priority_order(now, cfg, A, pd, rn if R else None)
    o2, p2 = _pyref_run(now, cfg, pd, rn if R else None)
    assert np.array(p2).view(np.uint64).tolist() == prio.view(np.uint64).tolist()
    assert o2 == order.tolist()
