#!/usr/bin/env python3
"""One partition of a synthetic configuration, full size, through the REFERENCE's own code (oracle/_ref) and through
the restated oracle; everything compared exactly (placements, fp64 cost bit patterns, every node's final time map).

    python tools/ref_fullsize.py C4 0            # partition 0 of C4: 125 k jobs x 8 192 nodes (the reference build needs MORE than 7 hours for it: stopped unfinished in round 4; use --budget)
    python tools/ref_fullsize.py tile10          # a tests/golden/make_fullrun.py case: also checks the committed digest
    python tools/ref_fullsize.py C4 0 --budget 600   # the longest PREFIX of that partition's queue the reference build finishes in
                                                 # about 600 s (its cost per decision grows steeply once the partition fills), the
                                                 # oracle on the same prefix beside it -> profiles/r04_ref_prefix_by_config.txt

Test infrastructure only (drives oracle/, never the product).  Appends one line to profiles/r04_ref_vs_oracle_fullsize.txt (round 3: r03_...).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cranesched_amd import synth  # noqa: E402
from oracle import pyoracle  # noqa: E402
from tests import fullrun  # noqa: E402
from tests.golden.make_fullrun import CASES  # noqa: E402


def prefix_run(tag, p, budget):
    """Grows a prefix of partition p's queue until the reference build has worked for about `budget` seconds."""
    c, j, now = synth.make_config(tag)
    sub, idx = synth.select_partitions(c, j, [p])
    n, last = min(sub.num_jobs, 4000), None
    while True:
        pre = synth.make_config(tag, J=int(idx[n - 1]) + 1)[1]
        psub, pidx = synth.select_partitions(c, pre, [p])
        b = pyoracle.select(c, psub, now, backend="ref")
        last = (psub, b)
        nxt = min(sub.num_jobs, int(n * 1.35))
        if nxt == n or b.seconds >= 0.6 * budget or b.seconds * (nxt / n) ** 3 > 1.3 * budget:
            break
        n = nxt
    psub, b = last
    a = pyoracle.select(c, psub, now)
    d = b.placements.diff(a.placements)
    same_cost = np.array_equal(a.costs().view(np.uint64), b.costs().view(np.uint64))
    tl_bad = 0
    for nn in range(c.num_nodes):
        x, y = a.timeline(nn), b.timeline(nn)
        if len(y["t"]) == 0:
            continue
        tl_bad += any(not np.array_equal(x[f], y[f]) for f in x)
    counts = np.bincount(b.placements.reason[:psub.num_jobs], minlength=8).tolist()
    line = (f"{tag} partition {p}, first {psub.num_jobs} of its {sub.num_jobs} jobs on {int(c.part_offsets[p + 1] - c.part_offsets[p])} nodes: "
            f"reference build {b.seconds:.1f} s = {psub.num_jobs / b.seconds:.0f} decisions/s, oracle {a.seconds:.2f} s = {psub.num_jobs / a.seconds:.0f} decisions/s; "
            f"placements {'IDENTICAL' if d is None else 'DIFFER at ' + str(d)}; fp64 costs {'identical' if same_cost else 'DIFFER'}; "
            f"time maps differing: {tl_bad}; reasons {counts}")
    print(line, flush=True)
    with open(os.path.join(ROOT, "profiles", "r04_ref_prefix_by_config.txt"), "a") as f:
        f.write(line + "\n")


def main():
    tag = sys.argv[1]
    if "--budget" in sys.argv:
        return prefix_run(tag, int(sys.argv[2]), float(sys.argv[sys.argv.index("--budget") + 1]))
    golden = None
    if tag.lower() in CASES:
        name, J, N, P = CASES[tag.lower()]
        c, j, now = synth.make_config(name, J=J, N=N, P=P)
        golden = np.load(os.path.join(ROOT, "tests", "golden", f"fullrun_{tag.lower()}.npz"))
        what = f"{tag.lower()} ({name} mix, {j.num_jobs} jobs x {c.num_nodes} nodes)"
        sub = j
    else:
        p = int(sys.argv[2])
        c, j, now = synth.make_config(tag)
        sub, idx = synth.select_partitions(c, j, [p])
        what = f"{tag} partition {p} ({sub.num_jobs} jobs x {int(c.part_offsets[p + 1] - c.part_offsets[p])} nodes)"
    t0 = time.time()
    a = pyoracle.select(c, sub, now)
    t1 = time.time()
    b = pyoracle.select(c, sub, now, backend="ref")
    t2 = time.time()
    d = b.placements.diff(a.placements)
    same_cost = np.array_equal(a.costs().view(np.uint64), b.costs().view(np.uint64))
    tl_bad = 0
    for n in range(c.num_nodes):
        x, y = a.timeline(n), b.timeline(n)
        if len(y["t"]) == 0:
            continue
        tl_bad += any(not np.array_equal(x[f], y[f]) for f in x)
    counts = np.bincount(b.placements.reason[:sub.num_jobs], minlength=8).tolist()
    line = (f"{what}: reference build {b.seconds:.1f} s, oracle {a.seconds:.2f} s; placements "
            f"{'IDENTICAL' if d is None else 'DIFFER at ' + str(d)}; fp64 costs {'identical' if same_cost else 'DIFFER'}; "
            f"time maps differing: {tl_bad}; reasons {counts}")
    if golden is not None:
        dg = fullrun.digest(b.placements, b.costs().view(np.uint64), b.timeline, c.num_nodes)
        line += f"; committed digest (tests/golden/fullrun_{tag.lower()}.npz): {fullrun.compare(dg, golden) or 'reproduced by the reference build'}"
    print(line, flush=True)
    with open(os.path.join(ROOT, "profiles", "r04_ref_vs_oracle_fullsize.txt"), "a") as f:
        f.write(line + "\n")


if __name__ == "__main__":
    main()
