"""Several devices inside the product (include/crane_gpu/node_select.h "several devices", csrc/group_host.inc): cns_group_* — one process,
N engines, the groups of partitions dealt over them, shards on their own host threads, the packed results all-gathered on the devices
and merged in queue order — and the one-process-per-device form (cns_comm_* + cns_allgather_results) that bench.py drives.  On a one-GPU
box: a group over ONE device runs the real ncclAllGather (world 1); a group over a REPEATED ordinal ([0, 0], [0, 0, 0, 0]) runs N engines
side by side on that GPU and gathers with device-to-device copies (RCCL refuses two ranks on one device) — the merged result must be the
single-engine result, bit for bit.  Reference: one SchedulerAlgo per controller, JobScheduler.cpp:158-159,1441; independent
LocalSchedulers per partition :6723-6732."""
import numpy as np
import pytest

from cranesched_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture
def classes(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cranesched_amd.engine import GpuNodeSelector, GpuNodeSelectorGroup
    return GpuNodeSelector, GpuNodeSelectorGroup


def _single(GpuNodeSelector, cluster, jobs, now, running=None, resv=None):
    e = GpuNodeSelector(device=0)
    try:
        e.set_nodes(cluster)
        if resv is not None:
            e.set_reservations(resv)
        if running is not None:
            e.set_running(running)
        return e.node_select(now, jobs), e.costs().view(np.uint64).copy()
    finally:
        e.close()


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0, 0]])
@pytest.mark.parametrize("name,J,N,P", [("C4", 40000, 4096, 8), ("C5", 30000, 2048, 8), ("C4p64", 30000, 4096, 64), ("C3", 4000, 512, 1)])
def test_group_select_is_the_single_engine_result(classes, devices, name, J, N, P):
    GpuNodeSelector, Group = classes
    cluster, jobs, now = synth.make_config(name, J=J, N=N, P=P)
    ref, _ = _single(GpuNodeSelector, cluster, jobs, now)
    g = Group(devices)
    try:
        g.set_nodes(cluster)
        got = g.node_select(now, jobs)
        assert got.diff(ref) is None
        info = g.info()
        assert info["num_devices"] == len(devices)
        assert info["gather_mode"] == ("rccl" if len(set(devices)) == len(devices) else "device-copies")
        for p in range(cluster.num_partitions):   # disjoint partitions: partition p on device p % N
            assert g.device_of_partition(p) == p % len(devices)
        again = g.node_select(now, jobs)          # a second cycle on the same group
        assert again.diff(ref) is None
    finally:
        g.close()


def test_group_with_running_jobs(classes):
    GpuNodeSelector, Group = classes
    cluster, jobs, now, running = synth.make_loaded("C4r", J=30000, N=4096, P=8)
    ref, _ = _single(GpuNodeSelector, cluster, jobs, now, running=running)
    g = Group([0, 0, 0])
    try:
        g.set_nodes(cluster)
        g.set_running(running)
        assert g.node_select(now, jobs).diff(ref) is None
    finally:
        g.close()


def test_group_keeps_partitions_that_share_nodes_together(classes):
    GpuNodeSelector, Group = classes
    cluster, jobs, now, _, _ = synth.make_mixed("C4all", J=20000, N=2048)   # an ALL partition over partition 0's nodes
    ref, _ = _single(GpuNodeSelector, cluster, jobs, now)
    g = Group([0, 0])
    try:
        g.set_nodes(cluster)
        P = cluster.num_partitions
        assert g.device_of_partition(0) == g.device_of_partition(P - 1)
        assert g.node_select(now, jobs).diff(ref) is None
    finally:
        g.close()


def test_more_devices_than_groups(classes):
    GpuNodeSelector, Group = classes
    cluster, jobs, now = synth.make_config("C2", J=5000, N=512, P=1)
    ref, _ = _single(GpuNodeSelector, cluster, jobs, now)
    g = Group([0, 0, 0])
    try:
        g.set_nodes(cluster)
        assert g.node_select(now, jobs).diff(ref) is None
    finally:
        g.close()


def test_comm_allgather_world_one(classes):
    """the one-process-per-device form on its only rank: id, communicator, ONE ncclAllGather of the packed results, download, unpack"""
    GpuNodeSelector, _ = classes
    from cranesched_amd import sharding
    cluster, jobs, now = synth.make_config("C4", J=20000, N=2048, P=8)
    e = GpuNodeSelector(device=0)
    try:
        e.set_nodes(cluster)
        e.upload_jobs(jobs); e.run_resident(now)
        ref = e.download()
        lay = e.results_layout()
        assert lay["total_bytes"] == sharding.results_layout(jobs.num_jobs, jobs.total_places())["total"]
        e.comm_init_rank(1, 0, GpuNodeSelector.comm_unique_id())
        slot = (lay["total_bytes"] + 15) & ~15
        e.allgather_results(slot)
        buf = e.download_gathered(slot)
        got = sharding.unpack_results(buf, jobs)
        assert got.diff(ref) is None
        ms, nbytes = e.gather_timing()
        assert nbytes == slot and ms >= 0
    finally:
        e.close()


def test_group_with_reservations(classes):
    """every device gets every reservation (their share comes out of the real nodes' time maps); the jobs submitted TO a reservation run on one device"""
    GpuNodeSelector, Group = classes
    cluster, jobs, now, _, _ = synth.make_mixed("C4v", J=30000, N=4096)
    resv = synth.mixed_reservations("C4v", cluster, now)
    ref, _ = _single(GpuNodeSelector, cluster, jobs, now, resv=resv)
    g = Group([0, 0, 0])
    try:
        g.set_nodes(cluster)
        g.set_reservations(resv)
        assert g.node_select(now, jobs).diff(ref) is None
    finally:
        g.close()
