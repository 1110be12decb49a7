#!/usr/bin/env python3
"""Cycle breakdown of k_wide (needs the -DCNS_PROF build):
   CNS_SELECT_KERNEL=wide CNS_ENGINE_LIB=cranesched_amd/libcrane_gpu_nodeselect_prof.so python tools/prof_wide.py [config] [J] [N] [P]"""
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
running = None
if name in synth.LOADED:
    c, j, now, running = synth.make_loaded(name, J=J, N=N, P=P)
else:
    c, j, now = synth.make_config(name, J=J, N=N, P=P)
e = GpuNodeSelector()
e.set_nodes(c)
if running is not None:
    e.set_running(running)
e.upload_jobs(j); e.run_resident(now)
t = e.timing(); pr = e.prof().astype(np.float64)
jobs = j.num_jobs / c.num_partitions
m = pr.mean(axis=0)
print(f"{name} J={j.num_jobs} N={c.num_nodes} P={c.num_partitions} {e.last_kernel()}: {t['select_ms']:.1f} ms = "
      f"{1e3*t['select_ms']/jobs:.2f} us/job/partition; supervisor loop {m[28]/jobs:.0f} cycles/job "
      f"(=> {m[28]/max(t['select_ms'],1e-9)/1e3:.0f} MHz counter)")
rows = {19: "leader scanner: whole job", 16: "leader:   (fetch + decode +) row loop", 17: "leader:   lap guard (waiting for the supervisor)",
        18: "leader:   argmins + publish + exchange wait", 10: "leader:     own argmins + payload", 11: "leader:     publish + own-row arithmetic + next record",
        12: "leader:     exchange wait (polls)", 14: "leader:   decision (argmin over the granules), node_num 1", 21: "leader: stopped (command wait + reload)",
        22: "supervisor: records -> task (incl. slot wait)", 23: "supervisor:   waiting for a free task slot", 26: "supervisor: stop handling"}
for k, v in rows.items():
    print(f"  {v:50s} {m[k]/jobs:10.0f} cyc/job")
if m[0]:
    print(f"  leader, node_num > 1: {m[0]:.0f} jobs; per such job: lists {m[1]/m[0]:.0f}, exchange wait {m[2]/m[0]:.0f}, vote {m[3]/m[0]:.0f} cycles; "
          f"{m[4]:.0f} second exchanges at {m[5]/max(m[4],1):.0f} cycles; {m[7]/max(m[0]+m[4],1):.2f} polls per list exchange")
ws = e.wide_stats()
print(f"  always-on counters (every build): {ws}; jobs per consuming look {j.num_jobs/max(ws['looks_consumed'],1):.2f}; "
      f"leader polls in front of a full ring per job {ws['leader_polls_ring_full']/j.num_jobs:.2f}")
print(f"  leader polls {m[13]:.0f} ({m[13]/max(m[20],1):.2f} per job)")
print(f"  leader jobs {m[20]:.0f}; supervisor: consumed {m[24]:.0f}, empty polls {m[27]:.0f}, stops {m[25]:.0f}, flushes {m[29]:.0f}")
if m[15]:
    print(f"  tester 0: {m[15]:.0f} retirements with progress at {m[6]/m[15]:.0f} cycles (incl. commits); {m[31]:.0f} idle naps")
if m[9]:
    print(f"  tester 0: {m[9]:.0f} tasks tested, {m[8]/m[9]:.0f} cyc/test (incl. dependency waits); 7 testers => {m[8]/m[9]/7:.0f} cyc/job of test capacity used")
if os.environ.get("CNS_PROF_TESTER"):   # a -DCNS_PROF_TESTER build: tester 0's test and commit by phase (in the leader's slots)
    n0 = max(m[0], 1)
    print(f"  tester 0, node_num 1 start now: {m[0]:.0f} tests; job record {m[1]/n0:.0f} | dependency wait {m[2]/n0:.0f} | node block {m[3]/n0:.0f} | "
          f"window minimum + exact test {m[4]/n0:.0f} cycles; verdict into the slot {m[5]/max(m[9],1):.0f} (all kinds)")
    print(f"  tester 0, node_num 1 backfill: {m[10]:.0f} tests at {m[11]/max(m[10],1):.0f} cycles; node_num > 1: {m[12]:.0f} tests at {m[13]/max(m[12],1):.0f} cycles; "
          f"dropped before the test {m[14]:.0f}")
    nc = max(m[16], 1)
    print(f"  tester 0, commits: {m[16]:.0f}; claim + task fields {m[17]/nc:.0f} | node blocks {m[18]/nc:.0f} | map + cost + arrays + record {m[19]/nc:.0f} | "
          f"start/reason + drain + state {m[20]/nc:.0f} cycles per commit")
if os.environ.get("CNS_PROF_FAST"):   # a -DCNS_PROF_FAST build: the scanners' steady-state loop, summed over the waves of a partition
    for w, what in ((0, "a wave that did not win the job before"), (1, "the wave that won the job before")):
        n = max(m[0 + w], 1)
        print(f"  steady-state loop, {what}: {m[0+w]:.0f} wave-jobs; record + rows + argmin + publish {m[2+w]/n:.0f} | publish -> exchange complete {m[4+w]/n:.0f} | "
              f"decision + row write {m[6+w]/n:.0f} cycles")
    n = max(m[0] + m[1], 1)
    print(f"  inside the first segment: record -> scalars + control words {m[10]/n:.0f} | rows {m[11]/n:.0f} | argmin cascades {m[12]/n:.0f} | "
          f"payloads + lap guard + publish {(m[2]+m[3]-m[10]-m[11]-m[12])/n:.0f} cycles")
    print(f"  wave-jobs that recomputed the res_total argmin {m[8]/n:.3f}; first poll found the exchange complete {m[9]/n:.3f}")
