#!/usr/bin/env python3
"""Which jobs flush k_wide's pipeline, and on which node (needs the -DCNS_DEBUG_FLUSH_LOG build):
   CNS_ENGINE_LIB=cranesched_amd/libcrane_gpu_nodeselect_prof.so python tools/flush_log.py C4r"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cranesched_amd import synth
from cranesched_amd.engine import GpuNodeSelector
name = sys.argv[1] if len(sys.argv) > 1 else "C4r"
c, j, now, running = synth.make_loaded(name) if name in synth.LOADED else (*synth.make_config(name), None)
e = GpuNodeSelector()
e.set_nodes(c)
if running is not None:
    e.set_running(running)
e.upload_jobs(j); e.run_resident(now)
P = c.num_partitions
out = np.zeros(P * 48 + 2048, np.uint64)
e._check(e._L.cns_debug_get_prof(e._h, out.ctypes.data_as(C.c_void_p), C.c_uint32(len(out))))
# cns_debug_get_prof caps at P*40: read the raw region through a second call is not possible; the cap is lifted in this build
log = out[P * 48:]
log = log[log != 0]
orig = (log >> np.uint64(32)).astype(np.int64)
cause = ((log >> np.uint64(24)) & np.uint64(0xFF)).astype(np.int64)
code = (log & np.uint64(0xFFFFFF)).astype(np.int64)
got = e.download()
gpu = j.gres_total.sum(axis=1) > 0 if j.gres_total is not None else np.zeros(j.num_jobs, bool)
print(f"{name}: {len(log)} flush records of partition 0; causes {np.bincount(cause, minlength=5).tolist()}")
print("  flushed jobs: gpu/npu request", int(gpu[orig].sum()), "of", len(orig), "; cpus", np.bincount(j.task_cpu_raw[orig] // 256, minlength=9).tolist(),
      "; node_num", np.bincount(j.node_num[orig], minlength=9).tolist())
print("  distinct nodes", len(np.unique(code)), "; most hit:", np.unique(code, return_counts=True)[1].max(), "times; distinct jobs", len(np.unique(orig)))
print("  time limits (h) of flushed jobs: mean", j.time_limit_sec[orig].mean() / 3600, "all jobs", j.time_limit_sec.mean() / 3600)
print("  outcome of flushed jobs:", np.bincount(got.reason[orig], minlength=4).tolist())
print("  first 12:", list(zip(orig[:12].tolist(), cause[:12].tolist(), code[:12].tolist())))
# the nodes: how long are their maps at the end, what do they look like now
for cd in np.unique(code)[:4]:
    node = int(c.part_nodes[c.part_offsets[0] + (cd & 0xFFF) + ((cd >> 12) * 4096)]) if True else 0
    tl = e.timeline(node)
    print(f"  node {node} (code {cd}): final map {len(tl['t'])} entries; cpu at now {tl['cpu_raw'][0] // 256}, min cpu {tl['cpu_raw'][:-1].min() // 256}, gres now {bin(int(tl['gres'][0])).count('1')}, min gres {min(bin(int(g)).count('1') for g in tl['gres'][:-1])}")
