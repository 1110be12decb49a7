"""k_wide's builds for clusters of many partitions — 16 scanner waves (1 + 4 workgroups per partition, 25..48 partitions of
one launch) and 8 scanner waves (1 + 2 workgroups, 49..80 partitions) — against the CPU oracle, bit for bit: a cross-section of
the parity suite under CNS_SELECT_KERNEL=wide16 / wide8 (the cap makes small clusters run on them too), the partition counts at
the edges of the four builds, and C4p64 (64 partitions of 1 024 nodes, 1 M jobs) against the oracle's whole-run digest."""
import numpy as np
import pytest

from cranesched_amd import synth
from tests import helpers, kat
from tests.test_gpu_parity import _run

pytestmark = pytest.mark.gpu


def _served_by(eng_name, waves):
    return eng_name.startswith("k_wide") and eng_name.split(" + ")[0].endswith(f" x{waves}")


@pytest.mark.parametrize("scn", kat.scenarios(), ids=lambda s: s[0])
def test_kat(engine_cls_narrow, scn):
    from tests.test_gpu_kat import test_kat_on_gpu
    test_kat_on_gpu(engine_cls_narrow, scn)


@pytest.mark.parametrize("name,J,N,P", [("C2", 20000, 1024, 1), ("C3", 12000, 1536, 1), ("C4", 20000, 2048, 8), ("C5", 30000, 1024, 8)])
def test_scaled_configs(engine_cls_narrow, name, J, N, P):
    c, j, now = synth.make_config(name, J=J, N=N, P=P)
    got, _ = _run(engine_cls_narrow, c, j, now, tag=f"{name} scaled")
    assert (got.reason[:J] == 1).sum() > 0, "scenario must exercise backfill"


@pytest.mark.parametrize("seed", range(8))
def test_random_heterogeneous(engine_cls_narrow, seed):
    c, j, now, run = helpers.random_case(seed)
    _run(engine_cls_narrow, c, j, now, running=run, tag=f"random{seed}")


@pytest.mark.parametrize("N", [380, 1100, 2048, 4096, 4100, 8192])
def test_tile_widths(engine_cls_narrow, N):
    # 1 / 2 / 4 / 8 rows per scanner lane of both builds (512 / 1 024 lanes per row); 4 100 nodes are beyond the 8-wave build
    c, j, now = synth.make_config("C3", J=3000, N=(N // 4) * 4, P=1)
    _run(engine_cls_narrow, c, j, now, tag=f"tile N={N}")


def test_deep_backfill_heterogeneous(engine_cls_narrow):
    c, j, now, run = helpers.random_case(11, N=10, J=2500, P=1, running=6)
    _run(engine_cls_narrow, c, j, now, running=run, tag="deep hetero")


@pytest.mark.parametrize("seed", [7, 8])
def test_wide_multi_node_jobs(engine_cls_narrow, seed):
    from tests.test_gpu_parity import test_wide_multi_node_jobs as t
    t(engine_cls_narrow, seed)


@pytest.mark.parametrize("seed", range(3))
def test_reservations_random(engine_cls_narrow, seed):
    from tests.test_reservations import test_gpu_reservation_random as t
    t(engine_cls_narrow, seed)


@pytest.mark.parametrize("P,waves", [(24, 32), (25, 16), (40, 16), (48, 16), (49, 8), (64, 8), (80, 8), (81, 0)])
def test_partition_counts_pick_the_widest_build_that_fits(built, monkeypatch, P, waves):
    """Default kernel choice: 64 scanner waves per partition up to 8 partitions, 32 up to 24, 16 up to 48, 8 up to 80 (every
    workgroup of the launch resident at once, a partition's workgroups on one XCD); beyond that k_pipe.  Against the oracle."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import pyoracle
    from cranesched_amd.engine import GpuNodeSelector
    monkeypatch.delenv("CNS_SELECT_KERNEL", raising=False)
    c, j, now = synth.make_config("C4", J=40000, N=64 * P, P=P)
    eng = GpuNodeSelector(device=0)
    try:
        eng.set_nodes(c)
        got = eng.node_select(now, j)
        ref = pyoracle.select(c, j, now)
        helpers.assert_same(eng, got, ref, c, tag=f"{P} partitions")
        name = eng.last_kernel()
        assert (_served_by(name, waves) if waves else name.startswith("k_pipe")), name
        assert (got.reason[:j.num_jobs] == 1).sum() > 0, "scenario must exercise backfill"
    finally:
        eng.close()


@pytest.mark.parametrize("tag", ["tile1", "tile3", "c2", "c5deep", "c4p64"])
def test_full_run_matches_oracle_digest(engine_cls_narrow, tag):
    """Whole queues against the oracle's digests (tests/golden/fullrun_*.npz): deep backfill on one partition (tile*, C2: 1 to 8
    rows per lane), 8 partitions of 2 048 nodes with 72 % of the jobs backfilled (c5deep), 64 partitions of 1 024 nodes (C4p64)."""
    from tests.test_gpu_fullrun import test_full_run_matches_oracle_digest as t
    t(engine_cls_narrow, tag)
