#!/usr/bin/env python3
"""k_select's counters by engine partition for a mixed cycle (needs a -DCNS_PROF build; rows are per engine partition):
   CNS_SELECT_KERNEL=legacy CNS_ENGINE_LIB=build_var/v_X.so python tools/prof_group.py [C4all]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cranesched_amd.engine import GpuNodeSelector
from tests.golden.make_fullrun import load_case5, load_resv

name = sys.argv[1] if len(sys.argv) > 1 else "C4all"
cluster, jobs, now, running, pre = load_case5(name, None)
e = GpuNodeSelector()
e.set_nodes(cluster)
resv = load_resv(name, cluster)
if resv is not None:
    e.set_reservations(resv)
if running is not None:
    e.set_running(running)
got = e.node_select(now, jobs) if pre is None else e.node_select_preempt(now, jobs, pre)[0]
print(f"{name} J={jobs.num_jobs}: {e.last_kernel()} {e.timing()['select_ms']:.1f} ms")
pr = e.prof().astype(np.float64)
ms = lambda c: c / 2370.0 / 1e3
for p in range(pr.shape[0]):
    r = pr[p]
    if r[:8].sum() == 0:
        continue
    print(f"  row {p}: inline A {r[11]:.0f} jobs, inline B {r[12]:.0f}, general {r[13]:.0f} in {ms(r[5]):.0f} ms, multi-node {r[15]:.0f} in {ms(r[6]):.0f} ms (start-now {r[30]:.0f}, backfill {r[31]:.0f}), rejected {r[14]:.0f}")
    print(f"         worker ms: wait for the scanners {ms(r[0]):.0f} | A block load {ms(r[1]):.0f} | window-min + test {ms(r[2]):.0f} | commit {ms(r[3]):.0f} | phase B {ms(r[4]):.0f} "
          f"(load {ms(r[8]):.0f}, alloc + next fit {ms(r[9]):.0f}, commit {ms(r[10]):.0f}) | decode + merge {ms(r[7]):.0f} | multi: record {ms(r[24]):.0f}, lists + merge {ms(r[25]):.0f}, verify + commit {ms(r[26]):.0f}, total lists {ms(r[28]):.0f}, common start + commit {ms(r[29]):.0f}")
    print(f"         scanner wave 1 ms: full scan + B1 {ms(r[17]):.0f} | pre-scan {ms(r[18]):.0f} | waiting for the verdict {ms(r[19]):.0f} | owner update {ms(r[20]):.0f} | merge wait {ms(r[21]):.0f}")
