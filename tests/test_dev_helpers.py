"""Host-compiled unit test of the device helpers (feasible / feasible_counts / priority_queue
emulation) against the oracle's MaskAlgebra and the real std::priority_queue."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dev_helpers(tmp_path):
    exe = tmp_path / "test_dev_helpers"
    subprocess.run(["g++", "-O1", "-std=c++20", os.path.join(ROOT, "tests", "cpp", "test_dev_helpers.cpp"), "-o", str(exe)],
                   check=True, cwd=ROOT)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok")
