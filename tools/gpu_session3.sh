#!/bin/bash
# round-2 session 3: lean-scan variants timed, then full-run digests + parity + adapter on the product build
export TMPDIR=/tmp
mkdir -p gpurun_out/s3
bash tools/gpu_var.sh s3/var C4,C5 head noub prod nap8 > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_fullrun.py tests/test_gpu_parity.py tests/test_host_adapter.py tests/test_overlap.py -q -m gpu -x > gpurun_out/s3/tests.log 2>&1
echo "rc=$?" >> gpurun_out/s3/tests.log
cat gpurun_out/s3/var.txt; grep -v amdgpu.ids gpurun_out/s3/tests.log | tail -15
