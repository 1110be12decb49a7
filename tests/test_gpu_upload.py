"""cns_upload_jobs on the device box: the host pass of round 5 (cranesched_amd/csrc/jobs_host.inc — chunks of the queue on several host
threads, page-locked staging, the caller's arrays already on their way while it runs) must hand the kernels the same tables whatever
the chunking, and a queue it rejects must leave the handle usable.  The pass itself against the one-thread walk: tests/test_jobs_host.py
(no GPU).  Reference: BasicPriority (JobScheduler.h:185-200), the pre-checks of the ordered loop (JobScheduler.cpp:6744-6761)."""
import dataclasses

import numpy as np
import pytest

from cranesched_amd import synth
from tests import helpers

pytestmark = pytest.mark.gpu


def test_chunked_upload_same_cycle_for_every_thread_count(engine_default, monkeypatch):
    """70 000 jobs on 8 partitions (large enough for four chunks): 1, 3 and the default number of host threads, from pageable arrays and
    from page-locked ones — one result, and it is the oracle's."""
    from oracle import pyoracle
    c, j, now = synth.make_config("C4", J=70000, N=4096, P=8)
    eng = engine_default(device=0)
    try:
        eng.set_nodes(c)
        monkeypatch.setenv("CNS_HOST_THREADS", "1")
        one = eng.node_select(now, j)
        ref = pyoracle.select(c, j, now)
        helpers.assert_same(eng, one, ref, c, tag="C4 70k, one host thread")
        for t in ("3", None):
            if t is None:
                monkeypatch.delenv("CNS_HOST_THREADS")
            else:
                monkeypatch.setenv("CNS_HOST_THREADS", t)
            assert one.diff(eng.node_select(now, j)) is None, f"CNS_HOST_THREADS={t}"
            pj, pout = eng.pinned_jobs(j), eng.pinned_placements(j)
            assert one.diff(eng.node_select(now, pj, out=pout)) is None, f"CNS_HOST_THREADS={t}, page-locked arrays"
            eng.free_pinned_jobs(pj)
        # the handle's own setting (cns_set_host_threads) goes before the environment's
        from cranesched_amd.engine import EngineError
        monkeypatch.setenv("CNS_HOST_THREADS", "1")
        eng.set_host_threads(2)
        assert one.diff(eng.node_select(now, j)) is None, "cns_set_host_threads(2)"
        with pytest.raises(EngineError) as e:
            eng.set_host_threads(65)
        assert e.value.status == -1
        eng.set_host_threads(0)
        assert one.diff(eng.node_select(now, j)) is None, "cns_set_host_threads(0)"
    finally:
        eng.close()


def test_rejected_queue_reports_the_first_bad_job_and_the_handle_goes_on(engine_default):
    from cranesched_amd.engine import EngineError
    c, j, now = synth.make_config("C4", J=70000, N=4096, P=8)
    eng = engine_default(device=0)
    try:
        eng.set_nodes(c)
        good = eng.node_select(now, j)
        k = j.node_num.copy()
        k[65000] = 0
        k[40123] = 0          # (in another chunk than 65 000: the FIRST one is reported, as the one-thread walk did)
        with pytest.raises(EngineError) as e:
            eng.node_select(now, dataclasses.replace(j, node_num=k))
        assert e.value.status == -1 and "job 40123:" in str(e.value)
        if len(c.gres.class_name) < 8:
            gs = j.gres_spec.copy()
            gs[69999, 7] = 1
            with pytest.raises(EngineError) as e:
                eng.node_select(now, dataclasses.replace(j, gres_spec=gs))
            assert e.value.status == -1 and "undefined GRES class" in str(e.value)
            # the invalid job wins over the undefined class, wherever the two sit
            with pytest.raises(EngineError) as e:
                eng.node_select(now, dataclasses.replace(j, node_num=k, gres_spec=gs))
            assert "job 40123:" in str(e.value)
        assert good.diff(eng.node_select(now, j)) is None
    finally:
        eng.close()


def test_batch_limit_and_unknown_partitions_across_chunks(engine_default):
    """ScheduledBatchSize in the middle of a chunk, jobs of unknown partitions and pre-set reasons sprinkled over the queue: against the oracle."""
    from oracle import pyoracle
    c, j, now = synth.make_config("C4", J=70000, N=4096, P=8)
    part = j.partition.copy()
    part[::97] = 8 + (np.arange(len(part[::97])) % 3)     # "Partition Not Found"
    skip = np.zeros(j.num_jobs, np.uint8)
    skip[5::211] = 1                                      # the caller pre-set a reason (e.g. "License")
    j2 = dataclasses.replace(j, partition=part, skip=skip)
    eng = engine_default(device=0, scheduled_batch_size=50001)
    try:
        eng.set_nodes(c)
        got = eng.node_select(now, j2)
        ref = pyoracle.select(c, j2, now, scheduled_batch_size=50001)
        helpers.assert_same(eng, got, ref, c, tag="C4 70k, batch 50 001")
        r = got.reason[:j.num_jobs]
        named = skip[:50001:97] == 0     # (a pre-set reason comes before the partition is looked at)
        assert (r[50001:] == 1).all() and (r[:50001:97][named] == 4).all() and (r[:50001][skip[:50001] == 1] == 5).all()
    finally:
        eng.close()
