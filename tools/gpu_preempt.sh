#!/bin/bash
out=gpurun_out/${1:-pre1}; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_preempt.py -x -q -m gpu > $out/preempt.log 2>&1
echo "rc=$?" >> $out/preempt.log
grep -v amdgpu.ids $out/preempt.log | tail -n 40
