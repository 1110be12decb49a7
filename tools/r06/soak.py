"""Soak of k_wide's multi-home protocol: the same full-size cycles again and again in one process, every result array and fp64 cost against the
first run's (which is held to the oracle's digest).  python tools/r06/soak.py <rounds> <tag> [<tag> ...]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ["CNS_WIDE_NO_RETRY"] = "1"
os.environ["CNS_SELECT_KERNEL"] = "wide"
import numpy as np
from tests.golden.make_fullrun import CASES, load_case5, load_resv
from tests import fullrun
from cranesched_amd.engine import GpuNodeSelector
rounds = int(sys.argv[1])
bad = 0
for tag in sys.argv[2:]:
    name, J, N, P = CASES[tag]
    cluster, jobs, now, running, pre = load_case5(name, J, N, P)
    ref = dict(np.load(f"tests/golden/fullrun_{tag}.npz"))
    resv = load_resv(name, cluster)
    first = None
    t0 = time.time()
    for r in range(rounds):
        os.environ["CNS_WIDE_AUX"] = str((1, 3, 2, 0)[r % 4])
        eng = GpuNodeSelector(device=0)
        eng.set_nodes(cluster)
        if resv is not None: eng.set_reservations(resv)
        if running is not None: eng.set_running(running)
        got = eng.node_select(now, jobs)
        costs = eng.costs().view(np.uint64).copy()
        if first is None:
            msg = fullrun.compare(fullrun.digest(got, costs, eng.timeline, cluster.num_nodes), ref)
            first = (got, costs)
        else:
            msg = got.diff(first[0]) or (None if np.array_equal(costs, first[1]) else "costs differ")
        if msg is not None:
            bad += 1
            print(tag, "round", r, "aux", os.environ["CNS_WIDE_AUX"], "DIFFERS:", msg, flush=True)
        eng.close()
    print(f"{tag}: {rounds} rounds in {time.time() - t0:.0f} s, {'all identical' if not bad else 'FAILURES'}", flush=True)
sys.exit(1 if bad else 0)
