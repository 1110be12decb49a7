#!/usr/bin/env python3
"""MultiFactorPriority on the GPU: time + HBM roofline of cns_priority_order at the benchmark queue length.
   python tools/prio_bench.py [J] [R] [A]      (prints one JSON line; CPU oracle timed beside it)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cranesched_amd.engine import GpuNodeSelector
from cranesched_amd.priority import PriorityConfig, synth_priority_case

J = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
A = int(sys.argv[3]) if len(sys.argv) > 3 else 256
pd, rn, now = synth_priority_case(J, R, A, seed=9)
cfg = PriorityConfig()
eng = GpuNodeSelector()
ms, wall = [], []
for _ in range(6):
    t0 = time.perf_counter()
    order, prio, _ = eng.priority_order(now, cfg, A, pd, rn)
    wall.append(1e3 * (time.perf_counter() - t0))
    ms.append(eng.priority_timing()["kernels_ms"])
nb = eng.priority_timing()["algorithmic_bytes"]
eng.close()
from oracle import pyoracle
t0 = time.perf_counter()
ro, rp = pyoracle.priority_order(now, cfg, A, pd, rn)
cpu_s = time.perf_counter() - t0
k = float(np.median(ms[1:]))
print(json.dumps({"what": "MultiFactorPriority order (bounds + service values + priorities + stable radix sort)",
                  "pending": J, "running": R, "accounts": A, "kernels_ms": k, "call_wall_ms_incl_pcie": float(np.median(wall[1:])),
                  "jobs_per_s_kernels": J / (k * 1e-3),
                  "roofline": {"bound": "hbm", "achieved": nb / (k * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                               "frac": nb / (k * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes": nb},
                  "cpu_baseline": {"value": J / cpu_s, "unit": "jobs/s", "cores": 1, "kind": "port", "seconds": cpu_s},
                  "bit_exact_vs_oracle": bool((order == ro).all() and (prio.view(np.uint64) == rp.view(np.uint64)).all())}))
