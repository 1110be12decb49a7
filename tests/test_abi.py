"""The C ABI library builds for gfx950, loads without a GPU, exports every symbol the header
declares, and refuses to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(name="node_select.h"):
    src = open(os.path.join(ROOT, "include", "crane_gpu", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cns_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported(built):
    from cranesched_amd import engine
    lib = engine.lib()
    names = header_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in node_select.h but not exported"
    assert set(names) == set(engine.ABI_SYMBOLS), "engine.ABI_SYMBOLS out of sync with the header"
    from cranesched_amd import abi
    assert lib.cns_abi_version() == abi.CNS_ABI_VERSION == 4
    # every header under include/crane_gpu/: priority.h (MultiFactorPriority)
    pnames = header_functions("priority.h")
    assert set(pnames) == set(engine.PRIORITY_ABI_SYMBOLS)
    for n in pnames:
        assert hasattr(lib, n), f"{n} declared in priority.h but not exported"
    # ... and run_limits.h (run-limit admission of the commit loop)
    lnames = header_functions("run_limits.h")
    assert set(lnames) == set(engine.LIMITS_ABI_SYMBOLS)
    for n in lnames:
        assert hasattr(lib, n), f"{n} declared in run_limits.h but not exported"
    # ... and steps.h (step scheduler)
    snames = header_functions("steps.h")
    assert set(snames) == set(engine.STEPS_ABI_SYMBOLS)
    for n in snames:
        assert hasattr(lib, n), f"{n} declared in steps.h but not exported"
    # ... and preempt.h (preemption inside the cycle, SURVEY 8f-4: served on the device)
    qnames = header_functions("preempt.h")
    assert set(qnames) == set(engine.PREEMPT_ABI_SYMBOLS)
    for n in qnames:
        assert hasattr(lib, n), f"{n} declared in preempt.h but not exported"
    assert sorted(os.listdir(os.path.join(ROOT, "include", "crane_gpu"))) == ["node_select.h", "preempt.h", "priority.h", "run_limits.h", "steps.h"]


def test_no_gpu_means_loud_failure(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from cranesched_amd import abi, engine
    with pytest.raises(engine.EngineError) as ei:
        engine.GpuNodeSelector(device=0)
    assert ei.value.status == -2  # CNS_ERR_NO_DEVICE: never a silent CPU path


def test_bad_abi_version_rejected(built):
    from cranesched_amd import abi, engine
    h = C.c_void_p()
    cfg = abi.CnsConfig(99, 0, 0, 0, 0, 0)
    assert engine.lib().cns_create(C.byref(cfg), C.byref(h)) == -1
    assert b"ABI" in engine.lib().cns_last_error(None)


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under cranesched_amd/ may import, link or exec it."""
    for d, _, files in os.walk(os.path.join(ROOT, "cranesched_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", "Makefile")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f"{f} imports the oracle"
                assert "liboracle" not in txt and "pyoracle" not in txt and "oracle/" not in txt, f"{f} uses the oracle"


def test_bench_names_the_shape_of_every_wide_build():
    """bench.py's `config.sharding` text is derived from the kernel name the engine reports (k_wide<NPL> x<scanner waves>)."""
    import bench
    assert bench._wide_shape("k_wide<2> x64") == "single GPU, 1 + 16 workgroups per partition (k_wide, 64 scanner waves)"
    assert bench._wide_shape("k_wide<4> x32").startswith("single GPU, 1 + 8 workgroups")
    assert bench._wide_shape("k_wide<2> x16").startswith("single GPU, 1 + 4 workgroups")
    assert bench._wide_shape("k_wide<2> x8").startswith("single GPU, 1 + 2 workgroups")
    assert bench._wide_shape("k_wide<2> x64 + k_select<37> on 1 of 8 partitions").startswith("single GPU, 1 + 16 workgroups")
    assert bench._wide_shape("k_wide") == "single GPU, k_wide"
