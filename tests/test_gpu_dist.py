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
