#!/bin/bash
# usage: tools/gpu_var.sh <outname> <configs,comma> <variant tags...>   — times engine build variants (build_var/v_<tag>.so)
out=gpurun_out/$1.txt; cfgs=${2//,/ }; shift 2
export TMPDIR=/tmp
: > $out
for v in "$@"; do
  CNS_ENGINE_LIB=build_var/v_$v.so timeout 200 python tools/var_bench.py $cfgs 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
