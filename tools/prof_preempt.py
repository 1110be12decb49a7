#!/usr/bin/env python3
"""TryPreempt_ on the device by phase and by size (needs a -DCNS_PROF_PRE build):
   CNS_ENGINE_LIB=build_var/v_ppre.so python tools/prof_preempt.py [C4rp] [J]
and, whatever the build, the digest check of the run against tests/golden/fullrun_c4rp.npz when the size is the committed one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cranesched_amd.engine import GpuNodeSelector
from tests import fullrun
from tests.golden.make_fullrun import load_case5

name = sys.argv[1] if len(sys.argv) > 1 else "C4rp"
J = int(sys.argv[2]) if len(sys.argv) > 2 else None
cluster, jobs, now, running, pre = load_case5(name, J)
e = GpuNodeSelector()
e.set_nodes(cluster)
e.set_running(running)
got, po = e.node_select_preempt(now, jobs, pre)
t = e.timing()
print(f"{name} J={jobs.num_jobs}: {e.last_kernel()} {t['select_ms']:.1f} ms")
if J is None:
    ref = dict(np.load(os.path.join(ROOT, "tests", "golden", f"fullrun_{name.lower()}.npz")))
    d = fullrun.digest(got, e.costs().view(np.uint64), e.timeline, cluster.num_nodes)
    pairs = [(j, (r | (1 << 31)) if is_pd else r) for j, lst in enumerate(po.lists()) for is_pd, r in lst]
    d["preempt_crc"] = fullrun.preempt_crc(np.asarray(pairs, np.int64).reshape(-1, 2), np.asarray(po.cancelled_ids(), np.int64))
    print("  digest:", fullrun.compare(d, ref) or "identical to the oracle's full run")
m = e.prof().astype(np.float64).sum(axis=0)
if m[0] and not os.environ.get("CNS_PROF_KSELECT"):   # (a -DCNS_PROF_PRE build; under -DCNS_PROF the same slots hold k_select's counters)
    mhz = 2370.0   # clock64() counts shader cycles (k_wide's supervisor loop over the kernel time: 2.37 GHz on these boxes)
    us = lambda c: c / mhz
    print(f"  TryPreempt_ calls {m[0]:.0f} (with candidates {m[11]:.0f}, trees satisfied {m[10]:.0f}, jobs preempted {m[16]:.0f}); "
          f"candidates per call {m[6]/m[0]:.1f}, added to the trees {m[15]/max(m[11],1):.1f}")
    print(f"  per call with candidates: gather {us(m[1])/m[0]:.0f} us (all calls) | order {us(m[2])/max(m[11],1):.0f} | time maps into the trees {us(m[3])/max(m[11],1):.0f} | "
          f"candidates in {us(m[4])/max(m[11],1):.0f} | candidates out again {us(m[5])/max(m[10],1):.0f} us")
    print(f"  per call with candidates: range operations {m[7]/max(m[11],1):.0f} ({m[12]/max(m[11],1):.0f} for the time maps), node visits {m[8]/max(m[11],1):.0f} "
          f"({m[13]/max(m[11],1):.0f}), tree nodes {m[9]/max(m[11],1):.0f} ({m[14]/max(m[11],1):.0f}), node fetches + claims that missed the LDS cache {m[17]/max(m[11],1):.0f}; "
          f"calls that ran out of compressed records and were redone node for node {m[18]:.0f}; total in TryPreempt_ {us(m[1]+m[2]+m[3]+m[4]+m[5])/1e3:.0f} ms of {t['select_ms']:.0f}")
if m[0] and not os.environ.get("CNS_PROF_KSELECT") and (m[19] or m[21]):
    print(f"  ordering: {m[21]:.0f} calls with more than 64 candidates took {us(m[22])/1e3:.0f} ms, {m[19]:.0f} of them past the rank sort's capacity took {us(m[20])/1e3:.0f} ms")
if m[0] and not os.environ.get("CNS_PROF_KSELECT") and m[23:29].any():   # record visits of the compressed trees by kind (count in the low 24 bits, cycles above)
    pr = e.prof().astype(np.uint64)
    names = ("push_up", "outside the range", "covered completely", "a leaf split into a chain", "range end inserted into a chain", "(every partial visit, whole)")
    for k, nm in enumerate(names):
        col = pr[:, 23 + k]
        n = int((col & np.uint64(0xFFFFFF)).sum()); cyc = int((col >> np.uint64(24)).sum())
        if n: print(f"  record visits, {nm}: {n} x {cyc / n:.0f} cycles = {cyc / 2370.0 / 1e3:.0f} ms")
    print(f"  range operations on the compressed trees: the calls of pre_crange_v as the caller sees them {us(m[30])/1e3:.0f} ms; apply() of the forward pass, whole {us(m[31])/1e3:.0f} ms")
if os.environ.get("CNS_PROF_KSELECT"):   # a -DCNS_PROF build (without CNS_PROF_PRE): k_select's own counters of the preempting partition (block 0)
    r = e.prof().astype(np.float64)
    r = r[int(np.argmax(r[:, 13]))]   # (one row of counters per partition: the one whose jobs took the general path)
    mhz = 2370.0
    ms = lambda c: c / mhz / 1e3
    print(f"  k_select worker of the preempting partition: inline path done in phase A {r[11]:.0f} jobs, phase B {r[12]:.0f}; general path {r[13]:.0f} jobs in {ms(r[5]):.0f} ms "
          f"({ms(r[5])*1e3/max(r[13],1):.0f} us each); multi-node protocols {r[15]:.0f} jobs in {ms(r[6]):.0f} ms; rejected candidates {r[14]:.0f}")
    print(f"  worker ms: wait for the scanners {ms(r[0]):.0f} | phase A block load {ms(r[1]):.0f} | window-min + test {ms(r[2]):.0f} | commit {ms(r[3]):.0f} | "
          f"phase B {ms(r[4]):.0f} | next-job decode + merge {ms(r[7]):.0f} | record -> LDS {ms(r[24]):.0f}")
    print(f"  scanner wave 1 ms: full scan + B1 {ms(r[17]):.0f} | pre-scan {ms(r[18]):.0f} | waiting for the verdict {ms(r[19]):.0f} | owner update {ms(r[20]):.0f} | merge wait {ms(r[21]):.0f}")
    if os.environ.get("CNS_PROF_DIP"):   # ... built with -DCNS_PROF_DIP too: the dips k_select's worker posted (slots of the multi-node protocols, unused on this partition)
        print(f"  rejected inline candidates: a dip found {r[28]:.0f}; front fine and no single entry fails by counts {r[27]:.0f}; the exact test fails on the front entry: by cpu {r[29]:.0f}, by memory {r[30]:.0f}, otherwise {r[31]:.0f}")
