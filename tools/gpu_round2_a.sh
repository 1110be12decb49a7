#!/bin/bash
# GPU session A: new pipelined kernel — KATs + parity, then full-run digests with both kernels
mkdir -p gpurun_out/a
export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_gpu_kat.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/a/parity.log 2>&1
echo "parity rc=$?" >> gpurun_out/a/parity.log
timeout 300 python -m pytest tests/test_gpu_fullrun.py -q -m gpu -s -k "c2 or c4 or c5 or tile" > gpurun_out/a/fullrun_pipe.log 2>&1
echo "fullrun pipe rc=$?" >> gpurun_out/a/fullrun_pipe.log
CNS_SELECT_KERNEL=legacy timeout 300 python -m pytest tests/test_gpu_fullrun.py -q -m gpu -s -k "c2 or c4 or c5 or tile" > gpurun_out/a/fullrun_legacy.log 2>&1
echo "fullrun legacy rc=$?" >> gpurun_out/a/fullrun_legacy.log
tail -5 gpurun_out/a/parity.log gpurun_out/a/fullrun_pipe.log gpurun_out/a/fullrun_legacy.log
