#!/usr/bin/env python3
"""Per-phase cycle breakdown of k_select (needs the -DCNS_PROF build):
   CNS_ENGINE_LIB=cranesched_amd/libcrane_gpu_nodeselect_prof.so python tools/prof_run.py [config] [J] [N] [P]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cranesched_amd import synth
from cranesched_amd.engine import GpuNodeSelector

name = sys.argv[1] if len(sys.argv) > 1 else "C4"
J = int(sys.argv[2]) if len(sys.argv) > 2 else None
N = int(sys.argv[3]) if len(sys.argv) > 3 else None
P = int(sys.argv[4]) if len(sys.argv) > 4 else None
c, j, now = synth.make_config(name, J=J, N=N, P=P)
e = GpuNodeSelector()
e.set_nodes(c); e.upload_jobs(j); e.run_resident(now)
t = e.timing(); pr = e.prof().astype(np.float64)
jobs = j.num_jobs / c.num_partitions
print(f"{name} J={j.num_jobs} N={c.num_nodes} P={c.num_partitions}: k_select {t['select_ms']:.1f} ms = "
      f"{1e3*t['select_ms']/jobs:.2f} us/job/partition")
names = {0: "W wait scanners (B1, no pre-scan)", 7: "W next-job decode + merge",  1: "W node block load", 2: "W window-min+feasible", 3: "W commit(now)",
         4: "W backfill+commit", 8: "W   B: node block load", 9: "W   B: alloc vs total + next fit", 16: "W     B: alloc vs total", 22: "W   A: before the window-min (counter latency only)", 23: "W   A: window-min", 10: "W   B: commit", 5: "W slow-path job", 6: "W multi-node job", 11: "#fast start-now",
         12: "#fast backfill", 13: "#slow jobs", 14: "#rejected candidates", 15: "#multi-node jobs", 24: "W multi: job record -> LDS", 25: "W multi: start-now lists + merge", 26: "W multi: helpers verify + commit", 28: "W multi: res_total lists + merge", 29: "W multi: common start + commit", 30: "#multi start-now", 31: "#multi backfill", 17: "S scan/mask completion (+B1)", 18: "S next-job prep+pre-scan",
         19: "S wait verdict", 20: "S owner update", 21: "S wait worker merge (B1')"}
m = pr.mean(axis=0)
tot_w = m[[0, 1, 2, 3, 4, 5, 6, 7]].sum(); tot_s = m[[17, 18, 19, 20, 21]].sum()  # 8..10 are parts of 4; tot_s = m[[17, 18, 19, 20, 21]].sum()
for k, v in names.items():
    if k in (11, 12, 13, 14, 15, 30, 31):
        print(f"  {v:32s} {m[k]:12.0f}")
    else:
        print(f"  {v:32s} {m[k]/jobs:10.0f} cyc/job   ({100*m[k]/(tot_w if k < 17 or k in (22, 23) else tot_s):5.1f}%)")
print(f"  worker total {tot_w/jobs:.0f} cyc/job, scanner total {tot_s/jobs:.0f} cyc/job")
