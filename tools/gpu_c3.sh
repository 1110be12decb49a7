#!/bin/bash
out=gpurun_out/c3; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python tools/c3_compare.py wide legacy 2>&1 | grep -v amdgpu.ids > $out/c3.txt
CNS_SELECT_KERNEL=wide CNS_ENGINE_LIB=cranesched_amd/libcrane_gpu_nodeselect_prof.so timeout 200 python tools/prof_wide.py C3 2>&1 | grep -v amdgpu.ids >> $out/c3.txt
cat $out/c3.txt
