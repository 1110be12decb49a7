#!/bin/bash
# A/B on ONE box: tools/gpu_ab_widen.sh "<configs>" <lib>... — k_wide runs (no retry on a fault) with the always-on protocol counters
export TMPDIR=/tmp
cfgs=$1; shift
for cfg in $cfgs; do
for lib in "$@"; do
  echo "== $cfg wide $lib"
  CNS_WIDE_NO_RETRY=1 CNS_SELECT_KERNEL=wide CNS_ENGINE_LIB=$lib timeout 200 python tools/prof_wide.py $cfg 2>&1 | grep -v amdgpu.ids | grep "us/job\|always-on\|rror\|fast run" | cut -c1-330
done; done
