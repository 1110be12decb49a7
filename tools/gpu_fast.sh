#!/bin/bash
# usage: tools/gpu_fast.sh <tag> <variant lib> — A/B of a k_wide variant against the default library on one box (full-size runs, retry off),
# then the full-size digests and a parity cross-section under the variant
export TMPDIR=/tmp
tag=$1; lib=$2
tools/gpu_ab_widen.sh "C4 C5 C2 C4p64" cranesched_amd/libcrane_gpu_nodeselect.so $lib > gpurun_out/${tag}_ab.txt 2>&1
cat gpurun_out/${tag}_ab.txt | grep -v always-on
CNS_ENGINE_LIB=$lib timeout 900 python -m pytest tests/test_gpu_fullrun.py tests/test_gpu_wide_narrow.py tests/test_gpu_parity.py -q -m gpu -x -k "wide" > gpurun_out/${tag}_tests.log 2>&1
echo "rc=$?" >> gpurun_out/${tag}_tests.log
grep -v amdgpu.ids gpurun_out/${tag}_tests.log | tail -n 8
