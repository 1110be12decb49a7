"""The multi-GPU code path of bench.py on ONE GPU (CNS_BENCH_FORCE_DIST=1): RCCL process group, device-pointer aliasing of the
engine's packed placement buffer, all_gather_into_tensor, unpack + merge on rank 0, diff against a single-engine run.  The driver's
1-GPU box thereby exercises the collective path every round (the 1 / 2 / 4 / 8-GPU curve itself needs an 8-GPU node)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("config,jobs,nodes", [("C4", 40000, 4096), ("C4r", 40000, 4096)])
def test_rccl_allgather_path_on_one_gpu(built, config, jobs, nodes):
    env = dict(os.environ, CNS_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--config", config,
                        "--jobs", str(jobs), "--nodes", str(nodes), "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["allgather_merged_identical_to_single_gpu"] is True
    assert line["config"]["selection_kernel"].startswith("k_wide")


def test_launch_is_sized_by_the_partitions_that_have_pending_jobs(built):
    """The reference builds a LocalScheduler only for partitions some pending job names (JobScheduler.cpp:6516-6530,6571-6573).
    A 100-partition snapshot with 5 busy partitions must run the widest k_wide build (x64: up to 8 partitions), not k_pipe
    (what 100 partitions would get), with results identical to the oracle's."""
    import numpy as np
    from cranesched_amd import synth
    from cranesched_amd.engine import GpuNodeSelector
    from oracle import pyoracle
    from tests import helpers
    c, j, now = synth.make_config("C4", J=60000, N=12800, P=100)
    busy = np.array([3, 17, 42, 64, 99], np.uint32)
    j.partition = busy[j.partition % 5]
    ref = pyoracle.select(c, j, now)
    eng = GpuNodeSelector(device=0)
    try:
        eng.set_nodes(c)
        got = eng.node_select(now, j)
        k = eng.last_kernel()
        assert k.startswith("k_wide") and " x64" in k and "5 busy of 100 partitions" in k, k
        helpers.assert_same(eng, got, ref, c, tag="5 busy of 100")
        # ... and the next cycle, with every partition busy, is back on the launch over all of them
        c2, j2, now2 = synth.make_config("C4", J=60000, N=12800, P=100)
        got2 = eng.node_select(now2, j2)
        assert "busy of" not in eng.last_kernel() and eng.last_kernel().startswith("k_pipe"), eng.last_kernel()
        helpers.assert_same(eng, got2, pyoracle.select(c2, j2, now2), c2, tag="100 busy of 100")
    finally:
        eng.close()


@pytest.mark.parametrize("world,expect", [(1, "k_pipe"), (4, " x8"), (8, " x16")])
def test_c4p256_shards_get_the_k_wide_build_the_plan_predicts(built, world, expect):
    """C4p256 (256 partitions of 256 nodes): what each rank of an N-GPU run executes — its own snapshot (sharding.shard_cluster),
    its own jobs — run here rank after rank on ONE GPU, merged like bench.py's all-gather, compared with the single-engine run.
    One GPU: 256 partitions -> k_pipe; 4 GPUs: 64 per rank -> k_wide x8; 8 GPUs: 32 per rank -> k_wide x16."""
    import numpy as np
    from cranesched_amd import sharding, synth
    from cranesched_amd.engine import GpuNodeSelector
    from oracle import pyoracle
    c, j, now = synth.make_config("C4p256", J=60000)
    ref = pyoracle.select(c, j, now)
    shards, kernels = [], set()
    for rk in range(world):
        sub, mine, idx = sharding.shard_cluster(c, j, rk, world) if world > 1 else (c, j, np.arange(j.num_jobs))
        eng = GpuNodeSelector(device=0)
        try:
            eng.set_nodes(sub)
            shards.append((eng.node_select(now, mine), idx))
            kernels.add(eng.last_kernel())
        finally:
            eng.close()
    assert all(expect in k for k in kernels), kernels
    merged = sharding.merge(j, shards) if world > 1 else shards[0][0]
    assert merged.diff(ref.placements) is None


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_under_torch_distributed_run_on_distinct_gpus(built, world):
    """The driver's launch line on a multi-GPU node (one rank per GPU, RCCL over xGMI): skips on the one-GPU boxes this project has had.
    Checks what the line must carry: the merged all-gather identical to one engine's run, the engine's own ncclAllGather (not the torch
    fall-back), the flat C4 line and — same ranks — the C4p64 line on which GPUs add chains."""
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, this box has {torch.cuda.device_count()}")
    for config in ("C4", "C4p64"):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                            "--master-port", str(29700 + os.getpid() % 200), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
                            "--config", config, "--no-cpu-baseline"], capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == world and line["allgather_merged_identical_to_single_gpu"] is True
        assert line["allgather"].startswith("engine"), line["allgather"]
