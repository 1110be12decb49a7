#!/usr/bin/env python3
"""Step scheduler (include/crane_gpu/steps.h): device time of cns_schedule_steps vs the CPU oracle on the same input.
   python tools/steps_bench.py [jobs]      -> one JSON line"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cranesched_amd.engine import GpuNodeSelector
from oracle import pyoracle
from tests import kat
from tests.test_steps import random_step_case

J = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
lay, jobs, steps = random_step_case(11, J=J)
eng = GpuNodeSelector()
eng.set_nodes(kat.cluster([4], layout=lay))
eng.schedule_steps(jobs, steps)                      # warm-up (buffers)
ms = []
for _ in range(5):
    got, k = eng.schedule_steps(jobs, steps)
    ms.append(k)
t0 = time.perf_counter()
ref = pyoracle.schedule_steps(lay, jobs, steps)
cpu_ms = 1e3 * (time.perf_counter() - t0)
S = steps.num_steps
nodes_b = jobs.num_nodes * 40 * 2 + S * 112 + int(steps.node_num.sum()) * 48 + int(steps.ntasks.sum()) * 44
print(json.dumps({"jobs": J, "job_nodes": int(jobs.num_nodes), "steps": S, "scheduled": int(got.scheduled[:S].sum()),
                  "tasks": int(steps.ntasks.sum()), "kernel_ms": float(np.median(ms)), "cpu_oracle_ms": cpu_ms,
                  "identical": got.diff(ref) is None, "steps_per_s": S / (np.median(ms) * 1e-3),
                  "bytes_by_construction": nodes_b, "GBps": nodes_b / (np.median(ms) * 1e-3) / 1e9}))
eng.close()
