import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
os.environ["CNS_WIDE_NO_RETRY"] = "1"
os.environ["CNS_SELECT_KERNEL"] = "wide"
from tests.golden.make_fullrun import CASES, load_case5, load_resv
from tests import fullrun
from cranesched_amd.engine import GpuNodeSelector
tag = sys.argv[1] if len(sys.argv) > 1 else "c4v"
name, J, N, P = CASES[tag]
cluster, jobs, now, running, pre = load_case5(name, J, N, P)
ref = dict(np.load(f"tests/golden/fullrun_{tag}.npz"))
res = {}
for aux in sys.argv[2:] or ["0", "1"]:
    os.environ["CNS_WIDE_AUX"] = aux
    eng = GpuNodeSelector(device=0)
    eng.set_nodes(cluster)
    rv = load_resv(name, cluster)
    if rv is not None: eng.set_reservations(rv)
    if running is not None: eng.set_running(running)
    got = eng.node_select(now, jobs)
    d = fullrun.digest(got, eng.costs().view(np.uint64), eng.timeline, cluster.num_nodes)
    print("aux", aux, eng.last_kernel(), "%.1f ms" % eng.timing()["select_ms"], fullrun.compare(d, ref), eng.wide_stats())
    res[aux] = (got.start_sec.copy(), got.reason.copy())
    eng.close()
keys = list(res)
if len(keys) >= 2:
    a, b = res[keys[0]], res[keys[1]]
    ne = np.nonzero((a[0] != b[0]) | (a[1] != b[1]))[0]
    print("differing jobs", len(ne))
    for j in ne[:40]:
        print(j, "part", jobs.partition[j], "resv", jobs.reservation[j] if jobs.reservation is not None else None, "k", jobs.node_num[j], "ntasks", jobs.ntasks[j],
              keys[0], a[0][j] - now if a[0][j] else 0, a[1][j], keys[1], b[0][j] - now if b[0][j] else 0, b[1][j])
