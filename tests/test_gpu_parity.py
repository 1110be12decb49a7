"""GPU parity proper: the HIP engine through the C ABI vs the CPU oracle, bit for bit."""
import numpy as np
import pytest

from cranesched_amd import synth
from tests import helpers

pytestmark = pytest.mark.gpu


def _run(engine_cls, cluster, jobs, now, running=None, tag="", **cfg):
    from oracle import pyoracle
    eng = engine_cls(device=0, **cfg)
    try:
        eng.set_nodes(cluster)
        if running is not None:
            eng.set_running(running)
        got = eng.node_select(now, jobs)
        ref = pyoracle.select(cluster, jobs, now, running=running, **cfg)
        helpers.assert_same(eng, got, ref, cluster, tag=tag)
        return got, eng.timing()
    finally:
        eng.close()


def test_c1_full(engine_cls):
    c, j, now = synth.make_config("C1")
    got, _ = _run(engine_cls, c, j, now, tag="C1")
    assert (got.reason[:j.num_jobs] == 0).all()


@pytest.mark.parametrize("name,J,N,P", [("C2", 20000, 1024, 1), ("C3", 12000, 1536, 1), ("C4", 20000, 2048, 8),
                                        ("C5", 30000, 1024, 8)])
def test_scaled_configs(engine_cls, name, J, N, P):
    c, j, now = synth.make_config(name, J=J, N=N, P=P)
    got, t = _run(engine_cls, c, j, now, tag=f"{name} scaled")
    r = got.reason[:J]
    assert (r == 1).sum() > 0, "scenario must exercise backfill"


@pytest.mark.parametrize("P", [8, 9, 20, 24, 25])
def test_many_partitions(engine_cls, P):
    """k_wide lays 17 (up to 8 partitions), 9 (up to 24), 5 (up to 48) or 3 (up to 80) workgroups per partition out over the XCDs in
    groups of 8 partitions: the edges of that layout, against the oracle (more of them: tests/test_gpu_wide_narrow.py)."""
    c, j, now = synth.make_config("C4", J=24000, N=128 * P, P=P)
    got, t = _run(engine_cls, c, j, now, tag=f"C4 with {P} partitions")
    assert (got.reason[:j.num_jobs] == 1).sum() > 0, "scenario must exercise backfill"


@pytest.mark.parametrize("seed", range(8))
def test_random_heterogeneous(engine_cls, seed):
    c, j, now, run = helpers.random_case(seed)
    _run(engine_cls, c, j, now, running=run, tag=f"random{seed}")


@pytest.mark.parametrize("seed,N,J", [(40, 96, 900), (41, 160, 2500), (42, 48, 2500), (43, 400, 4000)])
def test_nodes_with_192_and_256_cores(engine_cls, seed, N, J):
    """Core ids 128..255 (ABI 3: core_w2 / core_w3 through cns_node_soa, the running allocations, the time maps in HBM and the
    placement records): every feature of the heterogeneous cases on clusters whose big nodes have 192 or 256 cores; allocations
    cross the id-127 boundary, deep queues backfill into maps whose entries carry all four mask words."""
    c, j, now, run = helpers.random_case(seed, N=N, J=J, P=1 + seed % 3, running=N // 2)
    c = helpers.widen_cores(c, seed)
    got, _ = _run(engine_cls, c, j, now, running=run, tag=f"wide cores {seed}")
    assert (got.core_w2 != 0).any() and (got.core_w3 != 0).any(), "case must allocate core ids above 127 and above 191"
    assert (got.reason[:j.num_jobs] == 1).sum() > 20, "case must backfill"


@pytest.mark.parametrize("seed", [100, 101])
def test_random_tight_limits(engine_cls, seed):
    # kAlgoMaxJobNumPerNode and kAlgoMaxTimeWindow reached: nodes drop out (:6194), backfill gives up (h:815)
    c, j, now, run = helpers.random_case(seed, N=24, J=900, P=1, running=10)
    _run(engine_cls, c, j, now, running=run, tag=f"tight{seed}", max_job_num_per_node=12,
         max_time_window_sec=6 * 3600)


def test_batch_limit(engine_cls):
    c, j, now = synth.make_config("C1")
    got, _ = _run(engine_cls, c, j, now, tag="batch", scheduled_batch_size=400)
    assert (got.reason[400:1000] == 1).all()


@pytest.mark.parametrize("N", [380, 1100, 4100, 8192, 10000, 13000])
def test_tile_widths(engine_cls, N):
    # every register-tile width (nodes per scanner lane 1,3,10,19,28,37 at 7 scanner waves)
    c, j, now = synth.make_config("C3", J=3000, N=(N // 4) * 4, P=1)
    _run(engine_cls, c, j, now, tag=f"tile N={N}")


@pytest.mark.parametrize("name,J,N", [("C2", 5000, 16), ("C3", 4000, 24), ("C2", 14000, 8)])
def test_deep_backfill_long_time_maps(engine_cls, name, J, N):
    # few nodes, many jobs: time maps grow far beyond one 64-entry chunk (up to the 1000-entry limit),
    # which exercises the chunked window-min / commit / earliest-start routines and the :6194 cut-off
    c, j, now = synth.make_config(name, J=J, N=N, P=1)
    got, _ = _run(engine_cls, c, j, now, tag=f"deep {name}")
    r = got.reason[:J]
    assert (r == 1).sum() > J // 2


def test_deep_backfill_heterogeneous(engine_cls):
    c, j, now, run = helpers.random_case(11, N=10, J=2500, P=1, running=6)
    _run(engine_cls, c, j, now, running=run, tag="deep hetero")


@pytest.mark.parametrize("seed", [7, 8])
def test_wide_multi_node_jobs(engine_cls, seed):
    # node_num from 2 to 40 on a small cluster: the parallel helper protocol (k <= 8, start-now and common
    # earliest start), the sequential one-candidate-per-round protocol (9..32) and the general path (> 32),
    # interleaved with single-node jobs so that the pre-scan pipeline is broken and refilled all the time
    c, j, now, run = helpers.random_case(seed, N=160, J=500, P=2, running=30, general=False, lists=False,
                                         exclusive=False)
    rng = np.random.default_rng(1000 + seed)
    wide = rng.random(j.num_jobs) < 0.35
    k = np.where(wide, rng.choice([2, 3, 5, 8, 9, 12, 16, 31, 32, 33, 40], j.num_jobs), j.node_num).astype(np.uint32)
    j.node_num[:] = k
    j.ntasks[:] = k
    j.ntasks_per_node_min[:] = 1
    j.ntasks_per_node_max[:] = 1
    got, _ = _run(engine_cls, c, j, now, running=run, tag=f"wide{seed}")
    r = got.reason[:j.num_jobs]
    assert ((r == 0) & wide).sum() > 10 and ((r == 1) & wide).sum() > 10, "wide jobs must start now and backfill"


def test_wide_protocol_fault_is_retried_on_the_single_workgroup_kernels(built, monkeypatch):
    """k_wide's workgroups wait for each other; every wait is bounded and ends in a device fault.  Such a fault says
    nothing about the input: the cycle is re-run on k_pipe / k_select (one workgroup per partition, no co-residency
    needed) and the caller gets the exact result.  The hook makes the leader scanner of partition 0 drop its granules of
    one exchange, so that every other wave's wait really runs out (fault 28 after ~2 s)."""
    from cranesched_amd.engine import GpuNodeSelector
    from oracle import pyoracle
    monkeypatch.setenv("CNS_SELECT_KERNEL", "wide")
    monkeypatch.setenv("CNS_WIDE_NO_RETRY", "0")   # (conftest turns the retry off for every other test)
    c, j, now = synth.make_config("C4", J=6000, N=1024, P=8)
    ref = pyoracle.select(c, j, now)
    eng = GpuNodeSelector(device=0)
    try:
        eng.set_nodes(c)
        got = eng.node_select(now, j)
        assert eng.last_kernel().startswith("k_wide") and "retry" not in eng.last_kernel()
        helpers.assert_same(eng, got, ref, c, tag="before the stall")
        monkeypatch.setenv("CNS_WIDE_INJECT_STALL", "137")
        got = eng.node_select(now, j)
        k = eng.last_kernel()
        assert not k.startswith("k_wide") and "retry after" in k and "code 2" in k, k
        helpers.assert_same(eng, got, ref, c, tag="retried after the injected stall")
        monkeypatch.delenv("CNS_WIDE_INJECT_STALL")
        got = eng.node_select(now, j)
        assert eng.last_kernel().startswith("k_wide") and "retry" not in eng.last_kernel()   # the next cycle is back on k_wide
        helpers.assert_same(eng, got, ref, c, tag="after the stall")
    finally:
        eng.close()


def test_page_locked_caller_buffers(built, monkeypatch):
    """cns_host_alloc: the caller's job table and result arrays in page-locked memory (reused across cycles) — same results as from
    pageable numpy arrays, the arrays really are the handed-out buffers, cns_host_free rejects foreign pointers."""
    import ctypes as C
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cranesched_amd.engine import GpuNodeSelector, EngineError
    c, j, now = synth.make_config("C4", J=20000, N=2048, P=8)
    eng = GpuNodeSelector(device=0)
    try:
        eng.set_nodes(c)
        ref = eng.node_select(now, j)
        pj, pout = eng.pinned_jobs(j), eng.pinned_placements(j)
        bases = {p: a.nbytes for p, a in eng._pinned.items()}
        inside = lambda a: any(b <= a.ctypes.data and a.ctypes.data + a.nbytes <= b + n for b, n in bases.items())
        assert inside(pj.partition) and inside(pj.gres_spec) and inside(pout.start_sec) and inside(pout.node_idx)
        for _ in range(2):   # reused across cycles
            got = eng.node_select(now, pj, out=pout)
            assert got is pout and ref.diff(pout) is None
        with pytest.raises(EngineError):
            eng._check(eng._L.cns_host_free(eng._h, C.c_void_p(ref.start_sec.ctypes.data)))
        # the natural per-cycle use must not grow the set of page-locked buffers: refill in place ...
        n_bufs = len(eng._pinned)
        c2, j2, _ = synth.make_config("C4", J=20000, N=2048, P=8)
        j2.time_limit_sec = j2.time_limit_sec + 600
        pj2 = eng.pinned_jobs(j2, into=pj)
        assert len(eng._pinned) == n_bufs and pj2.partition.ctypes.data == pj.partition.ctypes.data
        ref2 = eng.node_select(now, j2)
        assert ref2.diff(eng.node_select(now, pj2, out=pout)) is None
        # ... a queue of another size swaps the buffers it must, and a set that is given back is gone
        c3, j3, _ = synth.make_config("C4", J=12000, N=2048, P=8)
        pj3 = eng.pinned_jobs(j3, into=pj2)
        assert len(eng._pinned) == n_bufs
        eng.free_pinned_jobs(pj3)
        assert len(eng._pinned) == n_bufs - sum(getattr(pj3, f) is not None for f in pj3.__dataclass_fields__)
        with pytest.raises(EngineError):
            eng.free_pinned(pj3.partition)
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pin,expect", [(0, "k_wide"), (2, "k_pipe"), (1, "k_select")])
def test_kernel_pin_of_the_config(gpu, monkeypatch, pin, expect):
    """cns_config::kernel_pin: a controller that shares its GPU keeps the engine off k_wide (whose workgroups must all be resident at once);
    the placements are the same under every pin"""
    monkeypatch.delenv("CNS_SELECT_KERNEL", raising=False)
    from cranesched_amd import synth
    from cranesched_amd.engine import GpuNodeSelector
    from oracle import pyoracle
    c, j, now = synth.make_config("C4", J=20000, N=2048, P=8)
    ref = pyoracle.select(c, j, now)
    eng = GpuNodeSelector(device=0, kernel_pin=pin)
    try:
        eng.set_nodes(c)
        got = eng.node_select(now, j)
        assert eng.last_kernel().startswith(expect), eng.last_kernel()
        assert got.diff(ref.placements) is None
    finally:
        eng.close()
