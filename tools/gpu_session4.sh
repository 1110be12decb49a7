#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/s4
for v in chk chknoub noub prod; do
  CNS_ENGINE_LIB=build_var/v_$v.so timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py tests/test_reservations.py -q -m gpu -k "pipe" > gpurun_out/s4/$v.log 2>&1
  echo "== $v"; grep -v amdgpu.ids gpurun_out/s4/$v.log | grep -E "fault|Error|passed|failed|FAILED" | head -12
done
