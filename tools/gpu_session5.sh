#!/bin/bash
export TMPDIR=/tmp CNS_SELECT_KERNEL=pipe
mkdir -p gpurun_out/s5
bash tools/gpu_var.sh s5/var C4,C5 head noub prod s512t2 s512t3 > /dev/null 2>&1
for v in chk; do
  CNS_ENGINE_LIB=build_var/v_$v.so timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py tests/test_reservations.py -q -m gpu -k "pipe" > gpurun_out/s5/$v.log 2>&1
  echo "== $v" >> gpurun_out/s5/var.txt; grep -v amdgpu.ids gpurun_out/s5/$v.log | grep -E "fault|Error|passed|failed|FAILED" | head -12 >> gpurun_out/s5/var.txt
done
: > gpurun_out/s5/prof.txt
for v in profprod; do for cfg in C4 C5; do
  CNS_ENGINE_LIB=build_var/v_$v.so timeout 120 python tools/prof_pipe.py $cfg 2>&1 | grep -v amdgpu.ids >> gpurun_out/s5/prof.txt
done; done
cat gpurun_out/s5/var.txt gpurun_out/s5/prof.txt
