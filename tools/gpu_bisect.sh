#!/bin/bash
# tools/gpu_bisect.sh <config> <J> <lib>...: one k_wide run per library, no retry on a fault (what fails is seen)
export TMPDIR=/tmp
cfg=$1; J=$2; shift 2
for lib in "$@"; do
  echo "== $cfg $J $lib"
  CNS_WIDE_NO_RETRY=1 CNS_SELECT_KERNEL=wide CNS_ENGINE_LIB=$lib timeout 100 python tools/prof_wide.py $cfg $J 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300
done
