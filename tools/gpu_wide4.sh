#!/bin/bash
# k_wide quick check: parity subset + full-run digests with timings
out=gpurun_out/${1:-w10}; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kat.py tests/test_gpu_parity.py tests/test_reservations.py -x -q -m gpu -k wide > $out/parity.log 2>&1
echo "parity rc=$?" >> $out/parity.log
timeout 300 python -m pytest tests/test_gpu_fullrun.py -q -m gpu -s -k "wide and (c2 or c4 or c5 or tile19 or tile37)" > $out/fullrun.log 2>&1
echo "fullrun rc=$?" >> $out/fullrun.log
grep -v amdgpu.ids $out/parity.log | tail -n 3; grep -v amdgpu.ids $out/fullrun.log | tail -n 9
