"""The DEVICE source of TryPreempt_'s two tree forms (cranesched_amd/csrc/preempt_dev.inc: the node-for-node tree behind its
LDS write-back cache, and the compressed one) compiled for the host and run against each other AND against the reference's
recursion (written out in tests/host_seg/seg_host.cpp) on random operation sequences.  With tests/test_seg_compact.py (the Python description against the reference's
recursion) this ties the code the GPU runs to the reference's tree without a GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def seg_host(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("seg_host") / "seg_host")
    subprocess.run(["g++", "-O1", "-std=c++17", "-fno-strict-aliasing", "-Wno-unused-function", "-o", exe, "seg_host.cpp"],
                   cwd=os.path.join(ROOT, "tests", "host_seg"), check=True)
    return exe


def test_device_source_of_both_tree_forms_agrees_on_the_host(seg_host):
    r = subprocess.run([seg_host, "4000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok: 4000 cases"), r.stdout + r.stderr
