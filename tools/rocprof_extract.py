#!/usr/bin/env python3
"""Summaries of rocprofv3 (rocpd sqlite) output for profiles/:
   rocprof_extract.py stats <dir>            -> CSV of the top_kernels view
   rocprof_extract.py pmc <dir> <COUNTER>    -> per-dispatch counter values of k_select (summed over dimensions)"""
import glob, sqlite3, sys

def db_of(d):
    f = sorted(glob.glob(d + "/**/*.db", recursive=True))
    if not f: sys.exit("no .db under " + d)
    return sqlite3.connect(f[-1])

mode, d = sys.argv[1], sys.argv[2]
c = db_of(d)
if mode == "stats":
    cols = [r[1] for r in c.execute("pragma table_info(top_kernels)")]
    print(",".join(cols))
    for r in c.execute("select * from top_kernels"): print(",".join(str(x) for x in r))
else:
    counter = sys.argv[3]
    views = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    v = "counters_collection" if "counters_collection" in views else [x for x in views if "counter" in x.lower()][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({v})")]
    rows = list(c.execute(f"select * from {v}"))
    ix = {n: i for i, n in enumerate(cols)}
    agg = {}
    for r in rows:
        name = r[ix.get("counter_name", ix.get("name", 0))]
        if name != counter: continue
        key = (r[ix["dispatch_id"]], r[ix.get("kernel_name", ix.get("name", 0))]) if "dispatch_id" in ix else (0, "")
        agg[key] = agg.get(key, 0.0) + float(r[ix.get("value", ix.get("counter_value"))])
    for (did, kn), val in sorted(agg.items()):
        print(f"{counter},{kn},{did},{val}")
