"""k_wide with 1, 2, 3 and 4 home workgroups per partition (cranesched_amd/csrc/wide_kernel.inc, "MORE THAN ONE HOME WORKGROUP PER
PARTITION"; CNS_WIDE_AUX=<extra homes>) against the oracle's whole-queue digests, and the protocol under perturbation.

The default launch runs ONE extra home (every other GPU test: two homes where the tile allows); here the same queues go through none
and through the build's maximum, incl. the cycles that stress what crosses the workgroups: C4v (thousands of flushes in reservations
whose jobs name the same nodes again and again: every dependency crosses the homes), C4r (a loaded cluster: 469 flushes), c5deep
(72 % backfills).  Model of the protocol: tests/test_multi_home_model.py."""
import os

import numpy as np
import pytest

from tests import fullrun
from tests.golden.make_fullrun import CASES, load_case5, load_resv

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


_CASE = {}


def _case(tag):
    """(reference digest, cluster, jobs, now, running, reservations) — built once per tag (a 1 M-job queue takes seconds to generate)."""
    if tag not in _CASE:
        name, J, N, P = CASES[tag]
        cluster, jobs, now, running, pre = load_case5(name, J, N, P)
        _CASE[tag] = (dict(np.load(os.path.join(GOLDEN, f"fullrun_{tag}.npz"))), cluster, jobs, now, running, load_resv(name, cluster))
    return _CASE[tag]


def _run(tag, monkeypatch, env, same_as=None):
    """One cycle under `env`.  same_as=None: the whole digest against the oracle's; else: every result array and every fp64 cost against an
    earlier run's (which was held to the oracle) — the time maps follow from those and cost a Python call per node to read."""
    monkeypatch.setenv("CNS_SELECT_KERNEL", "wide")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    from cranesched_amd.engine import GpuNodeSelector
    ref, cluster, jobs, now, running, resv = _case(tag)
    eng = GpuNodeSelector(device=0)
    try:
        eng.set_nodes(cluster)
        if resv is not None:
            eng.set_reservations(resv)
        if running is not None:
            eng.set_running(running)
        got = eng.node_select(now, jobs)
        costs = eng.costs().view(np.uint64).copy()
        if same_as is None:
            msg = fullrun.compare(fullrun.digest(got, costs, eng.timeline, cluster.num_nodes), ref)
        else:
            msg = got.diff(same_as[0])
            if msg is None and not np.array_equal(costs, same_as[1]):
                msg = "placements identical but fp64 cost bit patterns differ"
        return msg, eng.last_kernel(), eng.wide_stats(), eng.timing()["select_ms"], cluster, (got, costs)
    finally:
        eng.close()


@pytest.mark.parametrize("tag,aux", [("c2", "0"), ("c2", "3"), ("c4", "0"), ("c4", "3"), ("c5deep", "2"), ("c4r", "3"), ("c4v", "0"), ("tile10", "2")])
def test_digest_with_n_homes(gpu, monkeypatch, tag, aux):
    msg, kernel, ws, ms, cluster, _ = _run(tag, monkeypatch, {"CNS_WIDE_AUX": aux})
    assert kernel.startswith("k_wide"), kernel
    assert msg is None, f"{tag} with CNS_WIDE_AUX={aux}: {msg}"
    assert "retry after" not in kernel
    if tag in ("c2", "c4", "c5", "c5deep", "c4r", "tile10"):   # (64 scanner waves: up to 3 extra homes; every partition reports its count)
        busy = cluster.num_partitions
        assert ws["home_workgroups"] == busy * (1 + int(aux)), (ws, kernel)
    print(f"{tag} CNS_WIDE_AUX={aux}: identical to the oracle; {kernel} {ms:.1f} ms; homes {ws['home_workgroups']} flushes {ws['flushes']}")


@pytest.mark.parametrize("tag", ["c4", "c4v"])
def test_protocol_under_perturbation(gpu, monkeypatch, tag):
    """The lock-free protocol of 18 - 20 workgroups per partition must not depend on the instruction schedule or on who is faster
    (a type-punned load reordered by the compiler once made a regime non-deterministic).  The same digest five times
    in one process (the first against the oracle's digest, the others against the first) under settings that change the timing and nothing else: more / fewer homes, the supervisor's batch posting and the
    testers' fused commit off, one or many host threads, windows forced on / off, a second engine hammering the same GPU."""
    from cranesched_amd import synth
    from cranesched_amd.engine import GpuNodeSelector
    import threading
    stop = threading.Event()

    def hammer():   # another engine of this process keeps the GPU's other CUs busy (C2: one partition, 18 - 20 workgroups)
        c, j, now = synth.make_config("C2")
        e = GpuNodeSelector(device=0)
        try:
            e.set_nodes(c); e.upload_jobs(j)
            while not stop.is_set():
                e.run_resident(now)
        finally:
            e.close()

    first = None
    settings = [{"CNS_WIDE_AUX": "1"}, {"CNS_WIDE_AUX": "3", "CNS_HOST_THREADS": "1"}, {"CNS_WIDE_AUX": "2", "CNS_WIDE_BATCH_POST": "0", "CNS_WIDE_TESTER_OPT": "0"},
                {"CNS_WIDE_AUX": "0", "CNS_HOST_THREADS": "64", "CNS_WIDE_WINDOW": "16"}, {"CNS_WIDE_AUX": "3", "CNS_WIDE_WINDOW": "0", "_hammer": "1"}]
    for i, env in enumerate(settings):
        th = None
        if env.pop("_hammer", None):
            th = threading.Thread(target=hammer, daemon=True)
            th.start()
        try:
            msg, kernel, ws, ms, _, res = _run(tag, monkeypatch, env, same_as=first)
            first = first or res   # (run 0 against the oracle's digest, the others against run 0)
        finally:
            if th is not None:
                stop.set()
                th.join(timeout=120)
        for k in env:
            monkeypatch.delenv(k, raising=False)
        assert msg is None, f"{tag} run {i} under {env}: {msg}"
        assert kernel.startswith("k_wide") and "retry after" not in kernel, kernel
        print(f"{tag} run {i} {env}: identical; {kernel} {ms:.1f} ms, flushes {ws['flushes']}")
