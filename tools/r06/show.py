import json, sys, glob, os
d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "aux*_*.json"))):
    try:
        j = json.load(open(f)); w = j["config"]["wide_protocol_counters"] or {}
        print(os.path.basename(f)[:-5], "%.1f ms" % j["ms_per_step"], j["config"]["selection_kernel"], "homes", w.get("home_workgroups"), "ringfull", w.get("leader_polls_ring_full"),
              "empty", w.get("looks_empty"), "flushes", w.get("flushes"), "winjobs", w.get("jobs_decided_in_windows"))
    except Exception as e:
        print(os.path.basename(f), "FAIL", open(f[:-5] + ".err").read()[-400:])
