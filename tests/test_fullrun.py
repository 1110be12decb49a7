"""CPU side of the full-run digests: the per-partition merge of tests/golden/make_fullrun.py equals ONE oracle run
over all partitions (so the committed fullrun_*.npz digests are digests of the oracle's whole-queue result), and
the committed digests are self-consistent."""
import os

import numpy as np
import pytest

from cranesched_amd import synth
from tests import fullrun
from tests.golden import make_fullrun

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name,J,N,P", [("C4", 24000, 2048, 8), ("C5", 12000, 512, 4), ("C4r", 24000, 2048, 8), ("C4v", 24000, 2048, 8)])
def test_partition_merge_equals_single_run(built, name, J, N, P):
    from oracle import pyoracle
    cluster, jobs, full, costs, timelines, _ = make_fullrun.merged_run(name, J=J, N=N, P=P, procs=2)
    running = make_fullrun.load_case(name, J, N, P)[3]   # (C4r: the loaded cluster's running jobs, split by partition in the merge)
    ref = pyoracle.select(cluster, jobs, synth.NOW, running=running, reservations=make_fullrun.load_resv(name, cluster))
    if name == "C4v":   # the case must reach every reservation verdict
        rs = ref.placements.reason[:J]
        assert (rs == 3).sum() > 0 and (rs == 6).sum() > 0, np.bincount(rs, minlength=8)
    assert full.diff(ref.placements) is None
    a = fullrun.digest(full, costs, lambda n: timelines[n], cluster.num_nodes)
    b = fullrun.digest(ref.placements, ref.costs().view(np.uint64), ref.timeline, cluster.num_nodes)
    assert fullrun.compare(a, b) is None
    # (a later start is "Priority", or "Resource" when the allocation exceeds the cycle-start res_avail: loaded clusters)
    assert (ref.placements.start_sec[:J] > synth.NOW).sum() > J // 20, "case must exercise backfill"


@pytest.mark.parametrize("tag", list(make_fullrun.CASES))
def test_committed_digests_well_formed(tag):
    path = os.path.join(GOLDEN, f"fullrun_{tag}.npz")
    if tag == "c3" and not os.path.exists(path):   # one 1 M-job chain on 16 k nodes: hours of oracle time on one core
        pytest.skip(f"{path} not generated (python tests/golden/make_fullrun.py c3)")
    assert os.path.exists(path), f"{path} missing: run tests/golden/make_fullrun.py {tag}"
    d = np.load(path)
    name, J, N, P = make_fullrun.CASES[tag]
    base = {"C4all": "C4", "C4rp": "C4", "C4v": "C4", "C4all64k": "C4"}.get(name, synth.LOADED.get(name, (name,))[0])
    J = J or synth.CONFIGS[base]["J"]
    assert int(d["jobs"][0]) == J and int(d["nodes"][0]) == (N or synth.CONFIGS[base]["N"])
    assert len(d["chunk_crc"]) == (J + fullrun.CHUNK - 1) // fullrun.CHUNK
    assert int(d["counts"].sum()) == J
    if tag == "c4all64k":   # a prefix of C4's queue on the FULL cluster (the case is about the 131 072-slot group): 10.6 % backfilled
        assert d["counts"][1] >= J // 12
    elif tag != "c5":  # the frozen C5 queue does not fill its 64 k nodes (see make_fullrun.CASES["c5deep"])
        # later starts: "Priority", or "Resource" with a start time on a loaded cluster (allocation > cycle-start res_avail)
        assert d["counts"][1] + (d["counts"][2] if (name in synth.LOADED or name == "C4rp") else 0) >= J // 5, "the queue must reach the backfill regime (>= 20 % backfilled)"
