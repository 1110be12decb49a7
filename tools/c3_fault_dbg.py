#!/usr/bin/env python3
"""Diagnostics of a cost-chain fault (device fault 22) of k_wide: needs a -DCNS_DEBUG_FLUSH_LOG build and CNS_WIDE_NO_RETRY=1.
   CNS_WIDE_NO_RETRY=1 CNS_SELECT_KERNEL=wide CNS_ENGINE_LIB=build_var/v_dbg.so python tools/c3_fault_dbg.py C3 60000"""
import os, sys, ctypes as C, struct
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cranesched_amd import synth
from cranesched_amd.engine import GpuNodeSelector
name = sys.argv[1]; J = int(sys.argv[2])
c, j, now = synth.make_config(name, J=J)
e = GpuNodeSelector(); e.set_nodes(c); e.upload_jobs(j)
try:
    e.run_resident(now); print("run ok:", e.last_kernel())
except Exception as ex:
    print("run failed:", ex)
P = c.num_partitions
out = np.zeros(P * 48 + 2048, np.uint64)
e._check(e._L.cns_debug_get_prof(e._h, out.ctypes.data_as(C.c_void_p), C.c_uint32(len(out))))
log = out[P * 48:P * 48 + 2040]; log = log[log != 0]
orig = (log >> np.uint64(32)).astype(np.int64); cause = ((log >> np.uint64(24)) & np.uint64(0xFF)).astype(np.int64); code = (log & np.uint64(0xFFFFFF)).astype(np.int64)
print("flushes:", list(zip(orig.tolist(), cause.tolist(), code.tolist()))[-20:])
d = out[P * 48 + 2040:]
f = lambda x: struct.unpack('d', struct.pack('Q', int(x)))[0]
print("fault 22: cost0", f(d[0]), "P.cost", f(d[1]), "orig", int(d[2]) >> 32, "node", int(d[2]) & 0xFFFFFFFF, "start-now", int(d[3]) - now, "alloc.cpu", int(d[4]) / 256, "total.cpu", int(d[5]) / 256, "L", int(d[6]))
print("always-on counters", out[P * 32:P * 48].tolist())
