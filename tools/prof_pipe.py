#!/usr/bin/env python3
"""Cycle breakdown of k_pipe (needs the -DCNS_PROF build):
   CNS_ENGINE_LIB=cranesched_amd/libcrane_gpu_nodeselect_prof.so python tools/prof_pipe.py [config] [J] [N] [P]"""
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
m = pr.mean(axis=0)
print(f"{name} J={j.num_jobs} N={c.num_nodes} P={c.num_partitions}: selection kernel {t['select_ms']:.1f} ms = "
      f"{1e3*t['select_ms']/jobs:.2f} us/job/partition; supervisor loop {m[19]/jobs:.0f} cycles/job")
per_job = {1: "supervisor: decision -> task", 4: "supervisor: serial-mode jobs", 6: "supervisor: waiting for a task slot",
           16: "scanner w0: whole job (autonomous)", 17: "scanner w0: waiting for a command", 22: "scanner w0:   row loop",
           23: "scanner w0:   wave argmins", 24: "scanner w0:   publish + exchange wait", 20: "scanner w0:   decision + own-row update"}
for k, v in per_job.items():
    print(f"  {v:40s} {m[k]/jobs:10.0f} cyc/job")
print(f"  {'#tasks':40s} {m[5]:10.0f}   #serial jobs {m[2]:.0f}   #flushes {m[3]:.0f}   #scans (w6) {m[18]:.0f}")
print(f"  scanner w0 scans: full {m[28]:.0f}, lean {m[25]:.0f}")
busy = m[8:13]
print(f"  tester busy cycles per task (all testers): {busy.sum()/max(m[5],1):.0f}; per tester share of the kernel: "
      + ", ".join(f"{b/max(m[19],1):.2f}" for b in busy))
if m[15]:
    print(f"  tester 0: {m[15]:.0f} tasks, dep wait {m[13]/m[15]:.0f} cyc/task, verdict wait {m[14]/m[15]:.0f} cyc/task")
