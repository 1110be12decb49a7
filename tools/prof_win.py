#!/usr/bin/env python3
"""The window path of k_wide by segment (needs a -DCNS_PROF_WIN build):
   CNS_ENGINE_LIB=build_var/v_pwin.so python tools/prof_win.py [config]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cranesched_amd import synth
from cranesched_amd.engine import GpuNodeSelector

name = sys.argv[1] if len(sys.argv) > 1 else "C5"
if name in synth.LOADED:
    c, j, now, running = synth.make_loaded(name)
else:
    c, j, now = synth.make_config(name); running = None
e = GpuNodeSelector()
e.set_nodes(c)
if running is not None:
    e.set_running(running)
e.upload_jobs(j); e.run_resident(now)
t = e.timing(); pr = e.prof().astype(np.float64)
m = pr.sum(axis=0)
W = 64.0
nw = max(m[0], 1)
seg = ["queue + this wave's rows", "lap guard + publish + poll", "pool -> registers + bound", "the jobs on the pool"]
print(f"{name} {e.last_kernel()} {t['select_ms']:.1f} ms; windows per wave {m[0]/W/c.num_partitions:.0f} per partition, jobs per window {m[10]/nw:.2f}, polls per window {m[9]/nw:.2f}")
tot = 0
for i, s in enumerate(seg):
    print(f"  {s:28s} {m[1+i]/nw:9.0f} cycles per window and wave")
    tot += m[1 + i] / nw
jw = max(m[10] / nw, 1e-9)
print(f"  {'sum':28s} {tot:9.0f}  = {tot/jw:.0f} per job; the jobs alone {m[4]/nw/jw:.0f} per job")
print("  wide_stats:", e.wide_stats())
