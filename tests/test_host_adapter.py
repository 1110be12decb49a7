"""C++ host adapter (cranesched_amd/host): INodeSelectionAlgo::NodeSelect over the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "cranesched_amd", "host", "test_host_adapter")


def test_adapter_without_gpu_is_loud(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([EXE, "--no-gpu"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr


@pytest.mark.gpu
def test_adapter_known_answers_on_gpu(built):
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["lazy", "deferred"])
def test_whole_cycle_through_the_adapter_on_gpu(built, mode):
    """One NodeSelect per cycle through the C++ adapter on the GPU (pack into page-locked arrays kept across cycles, cns_select,
    write-back), with the default and the deferred write-back (allocated_res of the launched jobs on demand: MaterializeAllocation);
    the full-size run is `test_host_adapter --e2e-bench 65536 8 1000000 [deferred]` (profiles/r03_adapter_e2e.txt)."""
    r = subprocess.run([EXE, "--e2e-bench", "4096", "8", "40000"] + (["deferred"] if mode == "deferred" else []),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_incremental_running_pack_equals_full_pack(built):
    """the per-job cache of packed running allocations gives byte-identical cns_running_soa arrays, also
    after jobs ended / started (host-only, no device needed); the printed timings are the measurement."""
    r = subprocess.run([EXE, "--pack-bench", "2048", "20000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_pending_pack_and_write_back_host_only(built):
    """cns_job_soa packing + write-back of synthetic placements into PdJobInScheduler objects (no device needed); the deferred
    write-back followed by MaterializeAllocation builds the same objects as the full one."""
    r = subprocess.run([EXE, "--cycle-bench", "1024", "20000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_unsupported_configurations_are_refused_at_the_snapshot(built):
    """core ids >= 128 and PreemptType != NONE must surface as CNS_ERR_UNSUPPORTED when the snapshot is set
    (host-only check), not as silently wrong placements or a per-cycle GpuEngineError."""
    r = subprocess.run([EXE, "--config-checks"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_event_fed_mirror_equals_running_vector_walk(built):
    """fed by MallocResourceFromNode / FreeResourceFromNode, the adapter hands the engine the same running
    tables as the per-cycle walk over the running vector (after churn, an end-time change and a new snapshot)."""
    r = subprocess.run([EXE, "--mirror-check", "2048", "20000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["0", "0,0", "0,0,0,0"])
def test_one_algorithm_object_over_several_devices_on_gpu(built, devices):
    """GpuNodeSelectionAlgo(std::vector<int> devices): one engine per device behind ONE algorithm object (the reference builds one,
    JobScheduler.cpp:158-159), the groups of partitions dealt over them, shards on their own host threads, the packed results
    all-gathered on the devices (a repeated ordinal runs the engines side by side on one GPU and gathers with device-to-device
    copies; "0": ncclAllGather is not involved at all) and merged into the PdJobInSchedulers in queue order — against one engine, job by job."""
    r = subprocess.run([EXE, "--group-check", "4096", "8", "40000", devices], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


@pytest.mark.gpu
def test_whole_cycle_through_the_adapter_on_two_engines(built):
    r = subprocess.run([EXE, "--e2e-bench", "4096", "8", "40000", "lazy", "2", "--devices", "0,0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


@pytest.mark.gpu
def test_a_node_beyond_the_limits_refuses_only_its_partition_through_the_adapter(built):
    """a 512-core node in one of four partitions: its jobs leave NodeSelect as "GpuEngineRefused" (RefusedJobs / RefusedPartitions name them
    for the caller's CPU scheduler), the other partitions' jobs are placed exactly as without that node"""
    r = subprocess.run([EXE, "--refusal-check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
