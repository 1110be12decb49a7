"""The host pass of cns_upload_jobs (cranesched_amd/csrc/jobs_host.inc: validation, routing by partition / reservation, pre-set reasons,
placement offsets and the queue grouped by partition — two passes over chunks of the queue on several host threads since round 5) compiled
with g++ and compared with the one-thread walk it replaced (tests/cpp/jobs_host_test.cpp) on random queues, for every thread count 1..9,
incl. the error returns and their precedence.  No GPU involved: this is the part of the C ABI's `cns_select` bracket that runs on the host
(reference: BasicPriority, JobScheduler.h:185-200; the pre-checks of the ordered loop, JobScheduler.cpp:6744-6761; the split by partition,
:6516-6530)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def jobs_host(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("jobs_host") / "jobs_host_test")
    subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "cpp", "jobs_host_test.cpp")], check=True)
    return exe


def test_chunked_host_pass_equals_the_one_thread_walk(jobs_host):
    r = subprocess.run([jobs_host, "600"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.startswith("ok: 600 cases"), r.stdout + r.stderr


def test_bench_mode_at_the_headline_queue_length(jobs_host):
    # 1 M jobs through both passes at 1 .. 16 threads and through the one-thread walk (the timings are printed, not asserted)
    r = subprocess.run([jobs_host, "1", "bench"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "16 threads" in r.stdout and "one-thread walk" in r.stdout, r.stdout + r.stderr
