#!/bin/bash
# usage: tools/gpu_prof.sh <outdir> [configs...]   (cycle breakdown of k_pipe with the -DCNS_PROF build)
out=gpurun_out/$1; shift
mkdir -p $out
export TMPDIR=/tmp CNS_SELECT_KERNEL=pipe
for cfg in "$@"; do
CNS_ENGINE_LIB=cranesched_amd/libcrane_gpu_nodeselect_prof.so timeout 120 python tools/prof_pipe.py $cfg 2>&1 | grep -v amdgpu.ids > $out/prof_$cfg.txt
done
cat $out/prof_*.txt
