#!/usr/bin/env python3
"""Experiment helper: time k_select of one engine build on some configs and print a checksum of its output.
   CNS_ENGINE_LIB=build_var/v_X.so python tools/var_bench.py C4 C2"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cranesched_amd import abi, synth
if os.environ.get("CNS_VAR_ABI"):   # an older build of the engine (its cns_create checks the version; the structs only grew at the end)
    abi.CNS_ABI_VERSION = int(os.environ["CNS_VAR_ABI"])
from cranesched_amd.engine import GpuNodeSelector

tag = os.path.basename(os.environ.get("CNS_ENGINE_LIB", "default"))
for name in sys.argv[1:] or ["C4"]:
    c, j, now = synth.make_config(name)
    e = GpuNodeSelector()
    e.set_nodes(c); e.upload_jobs(j)
    ms = []
    for _ in range(2):
        e.run_resident(now); ms.append(e.timing()["select_ms"])
    got = e.download()
    h = hashlib.sha1()
    for a in (got.start_sec, got.reason, got.node_idx, got.ntasks, got.cpu_raw, got.mem, got.core_lo, got.core_hi, got.gres):
        h.update(np.ascontiguousarray(a).tobytes())
    print(f"{tag:14s} {name}: k_select {min(ms):8.1f} ms  {j.num_jobs/min(ms)/1e3:7.3f} M/s  sha1 {h.hexdigest()[:12]}  {e.last_kernel()}", flush=True)
    e.close()
