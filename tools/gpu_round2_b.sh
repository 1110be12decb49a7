#!/bin/bash
mkdir -p gpurun_out/b
export TMPDIR=/tmp
for cfg in C5 C4; do
CNS_ENGINE_LIB=cranesched_amd/libcrane_gpu_nodeselect_prof.so timeout 120 python tools/prof_pipe.py $cfg > gpurun_out/b/prof_$cfg.txt 2>&1
done
CNS_ENGINE_LIB=cranesched_amd/libcrane_gpu_nodeselect_prof.so timeout 120 python tools/prof_pipe.py C2 > gpurun_out/b/prof_C2.txt 2>&1
cat gpurun_out/b/prof_*.txt
