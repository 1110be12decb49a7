#!/bin/bash
# cycle breakdown of k_wide (PROF build) on C4 / C5 / one C3-like partition
out=gpurun_out/w2; mkdir -p $out
export TMPDIR=/tmp
for cfg in "C4" "C5" "C3 130000 8192 1"; do
CNS_SELECT_KERNEL=wide CNS_ENGINE_LIB=cranesched_amd/libcrane_gpu_nodeselect_prof.so timeout 200 python tools/prof_wide.py $cfg 2>&1 | grep -v amdgpu.ids >> $out/prof.txt
done
cat $out/prof.txt
