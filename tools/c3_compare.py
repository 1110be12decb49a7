#!/usr/bin/env python3
"""BASELINE.json config 3 (1 M jobs x 16 384 nodes, ONE partition) through every selection kernel that covers it:
timings, and the full-run digests of the kernels against each other (and against tests/golden/fullrun_c3.npz when the
oracle's run exists).  usage: python tools/c3_compare.py [kernels...]   (default: wide legacy)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cranesched_amd import synth
from cranesched_amd.engine import GpuNodeSelector
from tests import fullrun

kernels = sys.argv[1:] or ["wide", "legacy"]
cluster, jobs, now = synth.make_config("C3")
ref_path = os.path.join(ROOT, "tests", "golden", "fullrun_c3.npz")
ref = dict(np.load(ref_path)) if os.path.exists(ref_path) else None
first = None
for kname in kernels:
    os.environ["CNS_SELECT_KERNEL"] = kname
    eng = GpuNodeSelector(device=0)
    eng.set_nodes(cluster)
    got = eng.node_select(now, jobs)
    d = fullrun.digest(got, eng.costs().view(np.uint64), eng.timeline, cluster.num_nodes)
    t = eng.timing()
    print(f"C3 {jobs.num_jobs} jobs x {cluster.num_nodes} nodes, {eng.last_kernel()}: {t['select_ms']:.1f} ms = "
          f"{1e3 * jobs.num_jobs / t['select_ms']:.0f} decisions/s; start-now {d['counts'][0]}, backfilled {d['counts'][1]}, "
          f"failed {int(d['counts'][2:].sum())}; sha256 {bytes(d['sha256'][:8]).hex()}")
    if ref is not None:
        print("   vs the oracle's full run:", fullrun.compare(d, ref) or "identical")
    if first is None:
        first = (kname, d)
    else:
        print(f"   vs {first[0]}:", fullrun.compare(d, first[1]) or "identical")
    eng.close()
