#!/bin/bash
# usage: tools/gpu_profscan.sh <out> <lib> <configs...> — leader-scanner cycle breakdown (-DCNS_PROF -DCNS_PROF_SCAN_ONLY build), retry off
export TMPDIR=/tmp
out=gpurun_out/$1; lib=$2; shift; shift
: > $out
for cfg in "$@"; do
  CNS_WIDE_NO_RETRY=1 CNS_ENGINE_LIB=$lib timeout 200 python tools/prof_wide.py $cfg 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
