#!/bin/bash
mkdir -p gpurun_out/c
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kat.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/c/parity.log 2>&1
echo "parity rc=$?" >> gpurun_out/c/parity.log
timeout 300 python -m pytest tests/test_gpu_fullrun.py -q -m gpu -s -k "c2 or c4 or c5 or tile1 or tile3" > gpurun_out/c/fullrun_pipe.log 2>&1
echo "fullrun rc=$?" >> gpurun_out/c/fullrun_pipe.log
for cfg in C5 C4 C2; do
CNS_ENGINE_LIB=cranesched_amd/libcrane_gpu_nodeselect_prof.so timeout 120 python tools/prof_pipe.py $cfg > gpurun_out/c/prof_$cfg.txt 2>&1
done
tail -n 4 gpurun_out/c/parity.log; grep -v amdgpu.ids gpurun_out/c/fullrun_pipe.log | tail -n 14; cat gpurun_out/c/prof_*.txt | grep -v amdgpu.ids
