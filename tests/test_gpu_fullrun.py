"""BASELINE.json's configurations at FULL size, and contended single-partition cases for every register-tile width,
against the CPU oracle's whole-queue results (tests/golden/fullrun_*.npz, made by tests/golden/make_fullrun.py with
one oracle process per partition).  Compared: every job's start / reason / placement records (CRC per 10 000 jobs),
the fp64 cost bit pattern of every (partition, node) slot and the final time maps of a node sample (tests/fullrun.py).
Unlike a prefix of the queue these runs reach the loaded-cluster regime: 24-74 % of the jobs are backfilled."""
import os

import numpy as np
import pytest

from cranesched_amd import synth
from tests import fullrun
from tests.golden.make_fullrun import CASES, load_case5, load_resv

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("tag", list(CASES))
def test_full_run_matches_oracle_digest(engine_cls, tag):
    path = os.path.join(GOLDEN, f"fullrun_{tag}.npz")
    if tag == "c3" and not os.path.exists(path):   # one 1 M-job chain on 16 k nodes: hours of oracle time on one core
        pytest.skip(f"{path} not generated (python tests/golden/make_fullrun.py c3)")
    assert os.path.exists(path), f"{path} missing: run tests/golden/make_fullrun.py {tag}"
    if tag == "c4rp" and os.environ.get("CNS_SELECT_KERNEL") in ("pipe", "wide32"):
        pytest.skip("the preempting partition runs on k_select's general path under every setting (25 s each): legacy and wide cover "
                    "the two shapes of the cycle (one launch / split), the other partitions are C4r's")
    if tag == "c4all64k" and os.environ.get("CNS_SELECT_KERNEL") != "wide":
        pytest.skip("the one group of 131 072 slots runs on k_wide's home workgroup alone (k_mem) under every setting: once is enough")
    ref = dict(np.load(path))
    name, J, N, P = CASES[tag]
    cluster, jobs, now, running, pre = load_case5(name, J, N, P)
    eng = engine_cls(device=0)
    try:
        eng.set_nodes(cluster)
        resv = load_resv(name, cluster)
        if resv is not None:
            eng.set_reservations(resv)
        if running is not None:
            eng.set_running(running)
        if pre is None:
            got = eng.node_select(now, jobs)
        else:
            got, po = eng.node_select_preempt(now, jobs, pre)
        d = fullrun.digest(got, eng.costs().view(np.uint64), eng.timeline, cluster.num_nodes)
        if pre is not None:
            pairs = [(j, (ref | (1 << 31)) if is_pd else ref) for j, lst in enumerate(po.lists()) for is_pd, ref in lst]
            d["preempt_crc"] = fullrun.preempt_crc(np.asarray(pairs, np.int64).reshape(-1, 2), np.asarray(po.cancelled_ids(), np.int64))
        msg = fullrun.compare(d, ref)
        assert msg is None, f"{tag}: engine differs from the oracle's full run: {msg}"
        t = eng.timing()
        if name in ("C4all", "C4rp") and os.environ.get("CNS_SELECT_KERNEL") in ("wide", "wide32"):   # the split really happened
            assert " + k_select" in eng.last_kernel() and eng.last_kernel().startswith("k_wide"), eng.last_kernel()
        print(f"{tag}: {jobs.num_jobs} jobs x {cluster.num_nodes} nodes identical to the oracle "
              f"(start-now {d['counts'][0]}, backfilled {d['counts'][1]}, failed {d['counts'][2]}); "
              f"{eng.last_kernel()} {t['select_ms']:.1f} ms = {1e3 * jobs.num_jobs / t['select_ms']:.0f} decisions/s"
              + (f"; windows {ws['windows']} decided {ws['jobs_decided_in_windows']} dry {ws['windows_that_decided_nothing']} flushes {ws['flushes']}"
                 if eng.last_kernel().startswith("k_wide") and (ws := eng.wide_stats()) else ""))
    finally:
        eng.close()
